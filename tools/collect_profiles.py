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


def kernel_source_sha1():
    """hash of the device sources the counters were taken on: bench.py compares it with the sources it runs and says when a committed counter file is stale"""
    import hashlib
    h = hashlib.sha1()
    for f in ("dm_device.h", "dm_device_duo.h", "dm_types.h", "dm_math.h"):
        h.update(open(os.path.join(ROOT, "deepmimic_amd", "csrc", f), "rb").read())
    with open(os.path.join(ROOT, "deepmimic_amd", "csrc", "Makefile")) as fh:          # the code-generation flags of the kernel families, not the host-side rules
        h.update("".join(l for l in fh if l.startswith(("HIPFLAGS", "NOLICM_IDS", "SCHED_IDS", "licmflag", "ARCH"))).encode())
    return h.hexdigest()


def agg(path):
    a = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if "k_env_step" in r["Kernel_Name"] and int(r.get("Grid_Size", 1 << 30)) >= 64 * 1024:       # (bench.py's lone-wave probe launches 64-env batches of the same kernel)
            a[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            a[r["Dispatch_Id"]]["dur_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return list(a.values())[2:]            # drop the two warm-up launches


# 1. rocprofv3 --kernel-trace --stats summary (the kernel table only; torch's helper kernels trimmed to the top rows)
rows = list(csv.reader(open(os.path.join(src, "stats", "stats_kernel_stats.csv"))))
with open(os.path.join(dst, out + "_rocprofv3_kernel_stats.csv"), "w", newline="") as f:
    csv.writer(f).writerows(rows[:6])
if os.path.exists(os.path.join(src, "stats_policy", "stats_kernel_stats.csv")):
    rows = list(csv.reader(open(os.path.join(src, "stats_policy", "stats_kernel_stats.csv"))))
    with open(os.path.join(dst, out + "_rocprofv3_kernel_stats_closed_loop.csv"), "w", newline="") as f:
        csv.writer(f).writerows(rows[:12])
# 2. bench lines and reports
for name in ("bench.json", "bench_driver_style.json", "bench_groups1.json", "bench_physics2.json", "bench_8192.json", "bench_record_exchange_rccl_groups_1rank.json", "bench_spinkick.json", "bench_dog.json", "bench_dog_dense.json", "bench_pack1.json", "bench_facade.json", "policy_bench.json", "policy_bench_16384.json",
             "bench_amp_heading_zombie.json", "bench_amp_dribble_zombie.json", "bench_scenes.json",
             "parity_report.json", "tail_probe.txt", "tail_probe_closed_loop.txt", "bench_record_exchange_1rank.json", "bench_record_exchange_cabi_1rank.json"):
    if os.path.exists(os.path.join(src, name)) and os.path.getsize(os.path.join(src, name)) > 0:
        if name.startswith("bench_record_exchange"):      # RCCL prints its version banner to stdout behind the line: keep the JSON line only
            js = [l for l in open(os.path.join(src, name)).read().splitlines() if l.startswith("{")]
            open(os.path.join(dst, out + "_" + name), "w").write(js[-1] + "\n")
        else:
            shutil.copy(os.path.join(src, name), os.path.join(dst, out + "_" + name))
SCENES = [("humanoid3d_walk", ""), ("humanoid3d_spinkick", "_humanoid3d_spinkick"), ("dog3d_pace", "_dog3d_pace"),
          ("amp_heading_zombie", "_amp_heading_zombie"), ("amp_dribble_zombie", "_amp_dribble_zombie")]
summary = {}
for scene, sfx in SCENES:
    if os.path.exists(os.path.join(src, "phases%s.json" % sfx)):
        shutil.copy(os.path.join(src, "phases%s.json" % sfx), os.path.join(dst, out + "_phase_cycles%s.json" % sfx))
    bname = {"": "bench.json", "_humanoid3d_spinkick": "bench_spinkick.json", "_dog3d_pace": "bench_dog.json"}.get(sfx, "bench%s.json" % sfx)
    if not os.path.exists(os.path.join(src, bname)):
        continue
    b = json.load(open(os.path.join(src, bname)))
    n = b["config"]["envs_per_gpu"]
    # 3. PMC passes
    pmc = {"scene": scene, "envs": n, "kernel": b["roofline"]["kernel"], "kernel_source_sha1": kernel_source_sha1(),
           "note": "per-launch means of the step kernel; SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles (MI355X_MICROARCH.md)"}
    try:
        for d in ("pmc_sq", "pmc_sq2"):
            a = agg(os.path.join(src, d + sfx, "pmc_counter_collection.csv"))
            for k in a[0]:
                pmc[k if k != "dur_ns" else d + "_kernel_ns"] = float(np.mean([v[k] for v in a]))
        per_env = {k: v / n for k, v in pmc.items() if k.startswith("SQ_")}
        pmc["per_env_step"] = per_env
        clk = pmc["GRBM_GUI_ACTIVE"] / 8.0                   # counter is summed over the 8 XCDs
        pmc["derived"] = {"kernel_cycles": clk, "shader_clock_ghz": clk / pmc["pmc_sq2_kernel_ns"],
                          "valu_busy_fraction_of_simd_time": per_env["SQ_ACTIVE_INST_VALU"] * n * 4 / (clk * 256 * 4),
                          "valu_instructions_per_env_step": per_env["SQ_INSTS_VALU"]}
        json.dump(pmc, open(os.path.join(dst, out + "_pmc_sq%s.json" % sfx), "w"), indent=1)
        summary[scene] = pmc["derived"]
    except Exception as ex:
        print("PMC of", scene, "skipped:", ex)
    # 4. HBM traffic (separate passes; FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 FETCH_SIZE counts 64 B per 128-B request -> x2)
    try:
        f = np.mean([v["FETCH_SIZE"] for v in agg(os.path.join(src, "pmc_fetch" + sfx, "pmc_counter_collection.csv"))])
        w = np.mean([v["WRITE_SIZE"] for v in agg(os.path.join(src, "pmc_write" + sfx, "pmc_counter_collection.csv"))])
        traffic = {"scene": scene, "envs": n, "kernel": b["roofline"]["kernel"], "kernel_source_sha1": kernel_source_sha1(), "fetch_size_kib_raw": float(f), "write_size_kib_raw": float(w),
                   "fetch_bytes_per_launch": float(f) * 1024 * 2, "write_bytes_per_launch": float(w) * 1024,
                   "hbm_bytes_per_launch": float(f) * 1024 * 2 + float(w) * 1024,
                   "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_env_step"] * n,      # of the launch the COUNTERS saw (PMC passes run --groups 1: all n envs in one launch), not of the bench line's group launch
                   "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B); WRITE_SIZE uncalibrated, taken as KiB"}
        json.dump(traffic, open(os.path.join(dst, out + "_traffic%s.json" % sfx), "w"), indent=1)
        summary[scene + "/traffic"] = traffic["hbm_bytes_per_launch"]
        # patch the bench line of this scene: bench.py read the traffic file of the PREVIOUS collection while this round's PMC passes were still to come
        dstb = os.path.join(dst, out + "_" + bname)
        _b = json.load(open(dstb))
        _b["roofline"]["traffic"] = traffic["hbm_bytes_per_launch"] * _b["config"].get("envs_per_launch", n) / n      # (groups: a launch carries its share of the envs)
        open(dstb, "w").write(json.dumps(_b) + "\n")
    except Exception as ex:
        print("traffic of", scene, "skipped:", ex)
# 5. counted fp32 arithmetic of the headline kernel (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32, SQ_INSTS_VALU_MFMA_MOPS_*: wave-level instruction
# counts; a wave instruction is 64 lane operations, an FMA two flops, one MFMA "MOP" 512 flops)
try:
    a = agg(os.path.join(src, "pmc_flops", "pmc_counter_collection.csv"))
    b = json.load(open(os.path.join(src, "bench.json")))
    n = b["config"]["envs_per_gpu"]
    m = {k: float(np.mean([v[k] for v in a])) / n for k in a[0] if k != "dur_ns"}
    valu = 64.0 * (m["SQ_INSTS_VALU_ADD_F32"] + m["SQ_INSTS_VALU_MUL_F32"] + 2 * m["SQ_INSTS_VALU_FMA_F32"] + m["SQ_INSTS_VALU_TRANS_F32"])
    mfma = 512.0 * (m["SQ_INSTS_VALU_MFMA_MOPS_F32"] + m["SQ_INSTS_VALU_MFMA_MOPS_F64"])
    rate = b["value"]
    json.dump({"scene": "humanoid3d_walk", "envs": n, "kernel": b["roofline"]["kernel"], "kernel_source_sha1": kernel_source_sha1(), "per_env_step_wave_instructions": m,
               "issued_fp32_valu_lane_flops_per_env_step": valu, "matrix_core_flops_per_env_step": mfma,
               "issued_flops_per_env_step": valu + mfma, "env_steps_per_s": rate,
               "issued_tflops": (valu + mfma) * rate / 1e12, "valu_tflops": valu * rate / 1e12,
               "fraction_of_fp32_vector_peak_157_3": (valu + mfma) * rate / 157.3e12,
               "note": "ISSUED lane operations (all 64 lanes of every fp32 VALU instruction, masked / idle lanes included; v_pk_fma_f32 counted once "
                       "by the counter, i.e. a lower bound on packed work): an upper bound on useful flops per env-step (SURVEY 8d estimated 7 M useful)"},
              open(os.path.join(dst, out + "_flops.json"), "w"), indent=1)
    summary["flops"] = valu + mfma
except Exception as ex:
    print("flop counters skipped:", ex)
if os.path.exists(os.path.join(src, "stats_g1", "stats_kernel_stats.csv")):
    rows = list(csv.reader(open(os.path.join(src, "stats_g1", "stats_kernel_stats.csv"))))
    with open(os.path.join(dst, out + "_rocprofv3_kernel_stats_groups1.csv"), "w", newline="") as f:
        csv.writer(f).writerows(rows[:6])
if os.path.exists(os.path.join(src, "stats_dog", "stats_kernel_stats.csv")):
    rows = list(csv.reader(open(os.path.join(src, "stats_dog", "stats_kernel_stats.csv"))))
    with open(os.path.join(dst, out + "_rocprofv3_kernel_stats_dog3d_pace.csv"), "w", newline="") as f:
        csv.writer(f).writerows(rows[:6])
print(json.dumps(summary, indent=1))
