"""Start one process per GPU for a script of this repository (bench.py --gpus N, the trainer), the role of the
reference's mpiexec / Lightning launcher (``rho_diffusion/xpu.py:335-413``: rank / world from the environment, one
process per device, MASTER_ADDR / MASTER_PORT rendezvous).

The parent only forks children and relays their output: it never initialises HIP (no ``torch.cuda`` call, no
``exec``), so it is safe on pools where a process that touched the GPU must not be replaced.  Children receive
RANK / LOCAL_RANK / WORLD_SIZE / LOCAL_WORLD_SIZE / MASTER_ADDR / MASTER_PORT exactly as ``torch.distributed.run``
would set them, so a script cannot tell the two launchers apart.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
import time
from typing import Dict, List, Optional, Sequence


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def rank_env(rank: int, world: int, port: int, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, world))))
    return env


def rank_command(script: str, argv: Sequence[str]) -> List[str]:
    return [sys.executable, "-u", script, *argv]


def spawn_ranks(script: str, argv: Sequence[str], nproc: int, env: Optional[Dict[str, str]] = None,
                timeout: Optional[float] = None, stdout=None, stderr=None, json_only: bool = False) -> int:
    """Run ``script argv`` as ``nproc`` ranks; rank 0's stdout goes to ours, every rank's stderr to ours (prefixed for
    rank > 0), other ranks' stdout is dropped into stderr.  Returns 0 when every rank exited 0, otherwise the first
    non-zero exit code seen; when one rank fails the others are terminated (their own PIDs, never by pattern).
    ``json_only``: of rank 0's stdout only lines that start with ``{`` reach our stdout (a backend's own chatter, e.g. gloo's
    connection notes, goes to stderr), so a driver that parses ONE JSON line sees exactly that line."""
    if nproc < 1:
        raise ValueError("nproc must be >= 1")
    stdout = stdout or sys.stdout
    stderr = stderr or sys.stderr
    port = free_port()
    procs: List[subprocess.Popen] = []
    pumps: List[threading.Thread] = []

    def pump(src, dst, prefix):
        for line in iter(src.readline, ""):
            to = stderr if (json_only and dst is stdout and not line.lstrip().startswith("{")) else dst
            to.write(prefix + line)
            to.flush()
        src.close()

    for r in range(nproc):
        p = subprocess.Popen(rank_command(script, argv), env=rank_env(r, nproc, port, env), stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True, bufsize=1)
        procs.append(p)
        pre = "" if r == 0 else f"[rank {r}] "
        pumps.append(threading.Thread(target=pump, args=(p.stdout, stdout if r == 0 else stderr, pre), daemon=True))
        pumps.append(threading.Thread(target=pump, args=(p.stderr, stderr, pre), daemon=True))
    for t in pumps:
        t.start()

    rc = 0
    t0 = time.monotonic()
    live = set(range(nproc))
    kill_at = None                                      # once ranks were asked to stop: SIGKILL (by PID) what is left after 10 s
    while live:
        if kill_at is None and rc != 0:
            kill_at = time.monotonic() + 10.0
        if kill_at is not None and time.monotonic() > kill_at:
            for q in live:
                procs[q].kill()
            kill_at = float("inf")
        for r in sorted(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 128 - code      # killed by signal -> 128 + signo, as a shell reports it
                stderr.write(f"[launch] rank {r} exited with {code}; stopping the other ranks\n")
                for q in live:
                    procs[q].terminate()
        if live and timeout is not None and time.monotonic() - t0 > timeout:
            stderr.write(f"[launch] timeout after {timeout:.0f} s; stopping {len(live)} ranks\n")
            for q in live:
                procs[q].terminate()
            rc = rc or 124
            timeout = None
        if live:
            time.sleep(0.05)
    deadline = time.monotonic() + 10.0
    for p in procs:                                     # a rank that ignores SIGTERM is killed by PID
        try:
            p.wait(timeout=max(0.1, deadline - time.monotonic()))
        except subprocess.TimeoutExpired:
            p.kill()
    for t in pumps:
        t.join(timeout=5.0)
    return rc
