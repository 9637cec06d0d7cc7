"""Register / scratch use of every kernel of one source file (development aid): compiles it with hipcc's
-Rpass-analysis=kernel-resource-usage and prints name, VGPRs, AGPRs, spills, scratch, LDS, occupancy.
usage: python tools/kernel_regs.py rho_diffusion_amd/csrc/wgrad.hip [substring filter] [extra hipcc flags...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I",
       os.path.join(ROOT, "rho_diffusion_amd/csrc"), '-DRHO_BUILD_ID="probe"', "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
for line in out.split("\n"):
    m = re.search(r"remark: .*?Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sspill", r"SGPRs Spill: (\d+)"), ("vspill", r"VGPRs Spill: (\d+)"),
                     ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur:
            cur[key] = int(m.group(1))
            if key == "lds":
                if flt in cur["name"]:
                    dm = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
                    print(f'{dm[:110]:110s} v={cur.get("vgpr")} a={cur.get("agpr")} sspill={cur.get("sspill")} vspill={cur.get("vspill")} '
                          f'scratch={cur.get("scratch")} occ={cur.get("occ")}')
                cur = {}
