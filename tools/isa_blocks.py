"""Instruction mix per basic block of one kernel in a hipcc -S listing (development aid).
usage: python tools/isa_blocks.py conv.s <mangled kernel name>"""
import re, sys, collections
path, name = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
blocks, cur = [], None
for i in range(start, end):
    l = lines[i]
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m or cur is None:
        cur = {"label": m.group(1) if m else "entry", "line": i + 1, "c": collections.Counter(), "br": []}
        blocks.append(cur)
        if m: continue
    t = l.strip().split()
    if not t or t[0].startswith((";", ".")): continue
    op = t[0]
    c = cur["c"]
    if op.startswith("v_mfma"): c["mfma"] += 1
    elif op.startswith("v_"):
        c["valu"] += 1
        if re.match(r"v_(exp|rcp|log|rsq|sqrt|sin|cos)", op): c["trans"] += 1
    elif op.startswith("ds_"): c["ds_r" if "read" in op else "ds_w"] += 1
    elif op.startswith(("global_", "buffer_", "flat_")): c["vmem_l" if "load" in op else "vmem_s"] += 1
    elif op.startswith("s_waitcnt"): c["wait"] += 1
    elif op.startswith("s_barrier"): c["barrier"] += 1
    elif op.startswith(("s_cbranch", "s_branch")): cur["br"].append(t[1] if len(t) > 1 else "?"); c["salu"] += 1
    elif op.startswith("s_"): c["salu"] += 1
for b in blocks:
    n = sum(b["c"].values())
    if n >= int(sys.argv[3]) if len(sys.argv) > 3 else 20:
        print(f'{b["label"]:>12} @{b["line"]:<7} {dict(b["c"])} -> {b["br"]}')
tot = collections.Counter()
for b in blocks: tot.update(b["c"])
print("total", dict(tot))
