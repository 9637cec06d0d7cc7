"""Copy the artefacts of tests/gpu_profile_all.sh <tag> (gpurun_out/) into profiles/ under the round's naming and print the headline numbers.
usage: python tools/collect_profiles.py <tag>"""
import glob, json, os, re, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
for cfg in ("c3", "c5", "c2", "c1"):
    t = tag if cfg == "c3" else f"{tag}_{cfg}"
    for kind, sub in (("both", f"prof_{t}"), ("sample", f"prof_{t}_sample")):
        files = sorted(glob.glob(os.path.join(G, sub, "**", "*_kernel_stats.csv"), recursive=True), key=os.path.getmtime)
        if files:
            shutil.copy(files[-1], os.path.join(P, f"{tag}_{kind}_{cfg}_kernel_stats.csv"))
    for src, dst in ((f"ops_{t}.txt", f"{tag}_sample_{cfg}_per_launch.txt"), (f"ops_{t}.txt.bwd", f"{tag}_train_bwd_{cfg}_per_launch.txt")):
        if os.path.exists(os.path.join(G, src)):
            shutil.copy(os.path.join(G, src), os.path.join(P, dst))
    blog = os.path.join(G, f"bench_{t}.log")
    if os.path.exists(blog):
        text = open(blog).read()
        m = re.search(r"^\{.*\}$", text, re.M)
        name = f"{tag}_bench_default_c3.log" if cfg == "c3" else f"{tag}_bench_{cfg}.log"
        open(os.path.join(P, name), "w").write((m.group(0) if m else text[-4000:]) + "\n")
        if m:
            j = json.loads(m.group(0))
            r, tr = j["roofline"], j["training"]
            print(f"{cfg}: {j['value']:.2f} steps/s ({j['ms_per_step']:.1f} ms) ddim {j['ddim_sampling']['value']:.2f} | train {tr['value']:.1f} samples/s "
                  f"({tr['ms_per_step']:.1f} ms, peak {tr['peak_mem_gb']} GB; ckpt {tr['use_checkpoint']['samples_per_sec']:.1f} @ {tr['use_checkpoint']['peak_mem_gb']} GB) | "
                  f"{r['kind']} {r['achieved']:.0f} TF/s frac {r['frac']:.3f} exec {r['executed_frac']:.3f} build {r['build_id']}")
            print("   by_kind_ms", r["by_kind_ms"])
            print("   train", tr["roofline"]["frac"], tr["roofline"]["non_mfma_ms"], {k: (v["ms"], v["TFLOPs"]) for k, v in tr["roofline"]["per_kind"].items()})
