"""Summarise the rocprofv3 --pmc passes written by tests/gpu_pmc.sh (gpurun_out/pmc/{sq1,sq2,fetch,write}) into the two JSON
files kept under profiles/:  <tag>_pmc_traffic_conv3.json (HBM bytes per launch of the 3x3x3 k_conv variants: FETCH_SIZE in
KiB doubled per MI355X_MICROARCH.md's gfx950 correction + WRITE_SIZE in KiB) and <tag>_pmc_sq_counters_summary.json.
usage: python tools/pmc_summary.py <tag> [steps_profiled]"""
import collections, csv, glob, json, os, sys

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
    out[k[:70]] = v
suffix = "" if PMC_DIR == "pmc" else "_train"
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc{suffix}_sq_counters_summary.json"), "w"), indent=1)
if PMC_DIR != "pmc":
    sys.exit(0)

conv3 = [k for k in sq if "k_conv<" in k and "3, 3, 3" in k]
fetch = sum(sq[k].get("FETCH_SIZE", 0.0) for k in conv3) * 1024.0 * 2.0
write = sum(sq[k].get("WRITE_SIZE", 0.0) for k in conv3) * 1024.0
launches = sum(sq[k].get("_calls_fetch", 0) for k in conv3)
tj = {"kernel": "k_conv<bf16,3,3,3,*> (all variants)",
      "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- python3 bench.py --mode sample --steps 2 "
                 f"--warmup 1 ({steps:g} executions of the step profiled; tests/gpu_pmc.sh, tools/pmc_summary.py)",
      "launches_profiled": launches, "launches_per_step": launches / steps,
      "fetch_bytes_per_step": fetch / steps, "write_bytes_per_step": write / steps, "hbm_bytes_per_step": (fetch + write) / steps,
      "hbm_bytes_per_launch": (fetch + write) / max(launches, 1),
      "note": "FETCH_SIZE (KiB) x2 gfx950 correction per MI355X_MICROARCH.md; WRITE_SIZE (KiB) exact"}
json.dump(tj, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic_conv3.json"), "w"), indent=1)
print(json.dumps(tj, indent=1))
