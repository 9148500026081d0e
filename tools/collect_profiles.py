#!/usr/bin/env python3
"""Distil one gpurun_out/<tag>/ round profile (tools/gpu_round_profile.sh) into the tracked profiles/ directory.

    python tools/collect_profiles.py r01c r01
writes profiles/<out>_*.{csv,json}: rocprofv3 --kernel-trace --stats kernel table, the bench JSON lines, per-launch means of
the SQ / GRBM PMC passes, HBM traffic from the FETCH_SIZE / WRITE_SIZE passes (profiles/<out>_traffic.json is read by bench.py)."""
import collections, csv, json, os, shutil, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, out = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", tag); dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def agg(path):
    a = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if "k_env_step" in r["Kernel_Name"]:
            a[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            a[r["Dispatch_Id"]]["dur_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return list(a.values())[2:]            # drop the two warm-up launches


# 1. rocprofv3 --kernel-trace --stats summary (the kernel table only; torch's helper kernels trimmed to the top rows)
rows = list(csv.reader(open(os.path.join(src, "stats", "stats_kernel_stats.csv"))))
with open(os.path.join(dst, out + "_rocprofv3_kernel_stats.csv"), "w", newline="") as f:
    csv.writer(f).writerows(rows[:6])
# 2. bench lines
for name in ("bench.json", "bench_spinkick.json", "bench_dog.json", "bench_pack1.json"):
    shutil.copy(os.path.join(src, name), os.path.join(dst, out + "_" + name))
shutil.copy(os.path.join(src, "phases.json"), os.path.join(dst, out + "_phase_cycles.json"))
# 3. PMC passes
n = json.load(open(os.path.join(src, "bench.json")))["config"]["envs_per_gpu"]
pmc = {"envs": n, "note": "per-launch means of the step kernel of the default bench (k_env_step_duo<float, false>: 2048 waves, two characters each); SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles (MI355X_MICROARCH.md)"}
for d in ("pmc_sq", "pmc_sq2"):
    a = agg(os.path.join(src, d, "pmc_counter_collection.csv"))
    for k in a[0]:
        pmc[k if k != "dur_ns" else d + "_kernel_ns"] = float(np.mean([v[k] for v in a]))
per_wave = {k: v / n for k, v in pmc.items() if k.startswith("SQ_")}
pmc["per_env_step"] = per_wave
clk = pmc["GRBM_GUI_ACTIVE"] / 8.0                   # counter is summed over the 8 XCDs
pmc["derived"] = {"kernel_cycles": clk, "shader_clock_ghz": clk / pmc["pmc_sq2_kernel_ns"],
                  "valu_busy_fraction_of_simd_time": per_wave["SQ_ACTIVE_INST_VALU"] * 4 * 4 / clk,
                  "valu_instructions_per_env_step": per_wave["SQ_INSTS_VALU"]}
json.dump(pmc, open(os.path.join(dst, out + "_pmc_sq.json"), "w"), indent=1)
# 4. HBM traffic (separate passes; FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 FETCH_SIZE counts 64 B per 128-B request -> x2)
f = np.mean([v["FETCH_SIZE"] for v in agg(os.path.join(src, "pmc_fetch", "pmc_counter_collection.csv"))])
w = np.mean([v["WRITE_SIZE"] for v in agg(os.path.join(src, "pmc_write", "pmc_counter_collection.csv"))])
b = json.load(open(os.path.join(src, "bench.json")))
traffic = {"scene": "humanoid3d_walk", "envs": n, "kernel": b["roofline"]["kernel"], "fetch_size_kib_raw": float(f), "write_size_kib_raw": float(w),
           "fetch_bytes_per_launch": float(f) * 1024 * 2, "write_bytes_per_launch": float(w) * 1024,
           "hbm_bytes_per_launch": float(f) * 1024 * 2 + float(w) * 1024,
           "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"],
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE uncalibrated, taken as KiB"}
json.dump(traffic, open(os.path.join(dst, out + "_traffic.json"), "w"), indent=1)
print(json.dumps({"pmc": pmc["derived"], "per_env_step": per_wave, "traffic": traffic}, indent=1))

# patch the headline bench line: bench.py read the traffic file of the PREVIOUS collection while this round's PMC passes were still to come
try:
    _b = json.load(open(os.path.join(dst, out + "_bench.json")))
    _t = json.load(open(os.path.join(dst, out + "_traffic.json")))
    if _b.get("roofline", {}).get("kernel") == _t.get("kernel"):
        _b["roofline"]["traffic"] = _t["hbm_bytes_per_launch"]
        open(os.path.join(dst, out + "_bench.json"), "w").write(json.dumps(_b) + "\n")
except Exception as ex:
    print("bench traffic patch skipped:", ex)
