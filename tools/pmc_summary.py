"""Summarise the rocprofv3 --pmc passes written by tests/gpu_pmc.sh (gpurun_out/pmc/{sq1,sq2,fetch,write}) into the two JSON
files kept under profiles/:  <tag>_pmc_traffic_conv3.json (HBM bytes per launch of the 3x3x3 k_conv variants: FETCH_SIZE in
KiB doubled per MI355X_MICROARCH.md's gfx950 correction + WRITE_SIZE in KiB) and <tag>_pmc_sq_counters_summary.json.
usage: python tools/pmc_summary.py <tag> [steps_profiled]"""
import collections, csv, glob, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0   # plan-building eager step + 1 warm-up + 2 timed replays
PMC_DIR = sys.argv[3] if len(sys.argv) > 3 else "pmc"       # "pmc2": the training-step passes of tests/gpu_pmc2.sh (SQ counters only)


def collect(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    files = glob.glob(os.path.join(ROOT, "gpurun_out", PMC_DIR, sub, "**/*counter_collection.csv"), recursive=True)
    for f in sorted(files, key=os.path.getmtime)[-1:]:     # gpurun merges into the same directory: newest run only
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r.get("Dispatch_Id"))
            if key not in seen:
                seen.add(key); calls[k] += 1
    return agg, calls


meta = {}
mpath = os.path.join(ROOT, "gpurun_out", PMC_DIR, "meta.json")
if os.path.exists(mpath):
    meta = json.load(open(mpath))


def durations():
    """kernel name -> summed duration in seconds over the un-instrumented trace pass (tests/gpu_pmc.sh `trace`)."""
    dur = collections.defaultdict(float)
    files = glob.glob(os.path.join(ROOT, "gpurun_out", PMC_DIR, "trace", "**/*kernel_trace.csv"), recursive=True)
    for f in sorted(files, key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
    return dur


def clocks():
    """kernel name -> effective shader clock in GHz over the GRBM pass: sum(GRBM_GUI_ACTIVE) / 8 XCDs / sum(kernel duration of the
    same pass) (MI355X_MICROARCH.md, DVFS give-back: within 3 % of the in-kernel clock on dispatches of >= 0.3 ms)."""
    act, dur = collections.defaultdict(float), collections.defaultdict(float)
    base = os.path.join(ROOT, "gpurun_out", PMC_DIR, "grbm")
    cfiles = sorted(glob.glob(os.path.join(base, "**/*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]
    tfiles = sorted(glob.glob(os.path.join(base, "**/*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1:]
    for f in cfiles:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                act[r["Kernel_Name"]] += float(r["Counter_Value"])
    for f in tfiles:
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
    return {k: act[k] / 8.0 / dur[k] / 1e9 for k in act if dur.get(k)}


sq = collections.defaultdict(dict)
for sub in ("sq1", "sq2", "fetch", "write"):
    agg, calls = collect(sub)
    for k, v in agg.items():
        sq[k].update(v)
        sq[k]["_calls_" + sub] = calls[k]
out = {}
for k, v in sq.items():
    if v.get("SQ_INSTS_MFMA"):
        v["VALU_per_MFMA"] = v.get("SQ_INSTS_VALU", 0) / v["SQ_INSTS_MFMA"]
    if v.get("SQ_LDS_IDX_ACTIVE"):
        v["LDS_conflict_frac"] = v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"]
    if v.get("SQ_WAVE_CYCLES"):
        v["WAIT_INST_ANY_frac"] = v.get("SQ_WAIT_INST_ANY", 0) / v["SQ_WAVE_CYCLES"]
        v["WAIT_ANY_frac"] = v.get("SQ_WAIT_ANY", 0) / v["SQ_WAVE_CYCLES"]
    out[k[:110]] = v
CLK = clocks()
for k, v in sq.items():
    if k in CLK:
        v["effective_clock_GHz"] = CLK[k]
# MFMA utilisation against the gfx950 peak issue rate: wave-level MFMA instructions x SIMD cycles each (32 for 32x32x16 bf16,
# 16 for 16x16x32 bf16, 64 for 32x32x2 f32: MI355X_MICROARCH.md cycle constants) / (1024 SIMDs x kernel time x 2.4 GHz)
DUR = durations()
for k, v in sq.items():
    if v.get("SQ_INSTS_MFMA") and DUR.get(k):
        m_ = re.search(r"k_conv<([^>]*)>", k)                 # template arguments: T, KD, KH, KW, BM, MAXP, NW, M16[, FSK]
        a_ = [x.strip() for x in m_.group(1).split(",")] if m_ else []
        is_m16 = len(a_) >= 8 and a_[7] == "true"
        cyc = 64.0 if "<float" in k else (16.0 if (is_m16 and "k_conv" in k) else 32.0)
        v["duration_s_trace_pass"] = DUR[k]
        v["mfma_cycles_per_inst_assumed"] = cyc
        v["mfma_util"] = v["SQ_INSTS_MFMA"] * cyc / (1024.0 * DUR[k] * 2.4e9)
suffix = "" if PMC_DIR == "pmc" else "_train"
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc{suffix}_sq_counters_summary.json"), "w"), indent=1)
if PMC_DIR != "pmc":
    sys.exit(0)

# the launches bench.py files under kind "conv3": the 27-tap kernels and the 12-tap sub-pixel phases of the Upsample convs
# (... and the 1- / 2-tap parity launches of the stride-2 Downsample convs)
# (3-D workloads: kernel depth 3; 2-D workloads - c1 / c2 - run the same kinds on the 1 x k x k instantiations)
dims = int((meta.get("workload") or {}).get("dims", 3))
TAPS3 = ("3, 3, 3", "3, 2, 2", "3, 1, 1", "3, 1, 2", "3, 2, 1") if dims == 3 else ("1, 3, 3", "1, 2, 2", "1, 1, 2", "1, 2, 1", "1, 1, 3")
conv3 = [k for k in sq if "k_conv<" in k and any(("short, " + t) in k or ("float, " + t) in k for t in TAPS3)]
fetch = sum(sq[k].get("FETCH_SIZE", 0.0) for k in conv3) * 1024.0 * 2.0
write = sum(sq[k].get("WRITE_SIZE", 0.0) for k in conv3) * 1024.0
launches = sum(sq[k].get("_calls_fetch", 0) for k in conv3)
mf = sum(sq[k].get("SQ_INSTS_MFMA", 0.0) * sq[k].get("mfma_cycles_per_inst_assumed", 32.0) for k in conv3)
dsum = sum(DUR.get(k, 0.0) for k in conv3)
tj = {"kernel": ("k_conv<bf16,3,{3|2|1},{3|2|1},*>" if dims == 3 else "k_conv<f32|bf16,1,{3|2|1},{3|2|1},*>") + " (all variants of the launches bench.py files under kind conv3)", "build_id": meta.get("build_id"), "workload": meta.get("workload"),
      "mfma_util": (mf / (1024.0 * dsum * 2.4e9)) if dsum > 0 else None,
      "effective_clock_GHz": ({k[:60]: round(CLK[k], 3) for k in conv3 if k in CLK} or None),
      "mfma_util_note": "sum over the conv3 variants of SQ_INSTS_MFMA x SIMD cycles per instruction / (1024 SIMDs x summed kernel time "
                        "of the un-instrumented trace pass x 2.4 GHz)",
      "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 bench.py --mode sample --steps 2 "
                 f"--warmup 1 ({steps:g} executions of the step profiled; tests/gpu_pmc.sh, tools/pmc_summary.py)",
      "launches_profiled": launches, "launches_per_step": launches / steps,
      "fetch_bytes_per_step": fetch / steps, "write_bytes_per_step": write / steps, "hbm_bytes_per_step": (fetch + write) / steps,
      "hbm_bytes_per_launch": (fetch + write) / max(launches, 1),
      "note": "FETCH_SIZE (KiB) x2 gfx950 correction per MI355X_MICROARCH.md; WRITE_SIZE (KiB) exact"}
json.dump(tj, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic_conv3.json"), "w"), indent=1)
print(json.dumps(tj, indent=1))
