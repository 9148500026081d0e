#!/usr/bin/env python3
"""Distil gpurun_out/r06p (tools/gpu_round6_profile.sh) into profiles/r06_*.   usage: python tools/collect_round6.py [gpurun_out/r06p] [r06]
Reuses tools/collect_profiles.py (bench lines, pmc_sq, traffic) and tools/collect_classes.py (instruction classes, wait split) on the round-6 directory layout and adds what
VERDICT r5 #5 asked for: launch statistics per (kernel, GRID SIZE) from the whole kernel traces, the TCC hit rate beside the HBM traffic, the FETCH_SIZE / WRITE_SIZE calibration
on the step kernels' access pattern, matrix-core flops from the MFMA counters, the closed-loop diagnostics."""
import csv, json, os, shutil, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rel = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r06p"; out = sys.argv[2] if len(sys.argv) > 2 else "r06"
src, dst = os.path.join(ROOT, rel), os.path.join(ROOT, "profiles")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
# the --stats tables of the traces where collect_profiles.py expects them (kept for continuity with r01..r05; the by-grid tables below are the ones to read)
for a, b in (("trace_walk", "stats"), ("trace_dog_groups1", "stats_dog"), ("trace_closed", "stats_policy")):
    if os.path.exists(os.path.join(src, a, "t_kernel_stats.csv")):
        os.makedirs(os.path.join(src, b), exist_ok=True)
        shutil.copy(os.path.join(src, a, "t_kernel_stats.csv"), os.path.join(src, b, "stats_kernel_stats.csv"))
subprocess.call([sys.executable, os.path.join(ROOT, "tools", "collect_profiles.py"), os.path.basename(src), out])
subprocess.call([sys.executable, os.path.join(ROOT, "tools", "collect_classes.py"), rel, out])
for name in ("bench_dog_groups1.json",):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, out + "_" + name))
for t, name in (("trace_walk", "kernel_stats_by_grid"), ("trace_dog_groups1", "kernel_stats_by_grid_dog3d_pace_groups1"), ("trace_dog_groups2", "kernel_stats_by_grid_dog3d_pace_groups2"),
                ("trace_closed", "kernel_stats_by_grid_closed_loop_spinkick")):
    p = os.path.join(src, t + ".by_grid.csv")
    if os.path.exists(p):
        rows = list(csv.reader(open(p)))
        with open(os.path.join(dst, "%s_%s.csv" % (out, name)), "w", newline="") as f:
            csv.writer(f).writerows(rows[:9])
for sc in ("humanoid3d_walk", "humanoid3d_spinkick", "dog3d_pace"):
    p = os.path.join(src, "closed_loop_%s.json" % sc)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, "%s_closed_loop_%s.json" % (out, sc.replace("humanoid3d_", "").replace("dog3d_pace", "dog"))))


def per_launch(path, counters, min_grid):
    acc = {}
    for r in csv.DictReader(open(path)):
        if "k_env_step" in r["Kernel_Name"] and int(r["Grid_Size"]) >= min_grid:
            acc.setdefault(r["Dispatch_Id"], {}).setdefault(r["Counter_Name"], 0.0)
            acc[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    rows = list(acc.values())[2:]
    return {c: float(np.mean([v[c] for v in rows])) for c in counters}


# calibration of the traffic counters on the kernels' access pattern (tools/fetch_calib.hip)
try:
    G, cal = 2048, {}
    for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum_TCC_MISS_sum"):
        for l in open(os.path.join(src, "calib_%s.jsonl" % c)):
            r = json.loads(l); cal.setdefault(r["kernel"].replace("void ", ""), {}).update({k: v for k, v in r.items() if k not in ("kernel", "grid", "dispatches")})
    for k, v in cal.items():
        W = 43 if "<43" in k else 48
        rd, wr = 3 * G * W * 4, 3 * G * W * 4 + G * 227 * 4
        v.update(bytes_read=rd, bytes_written=wr, FETCH_SIZE_KiB_x1024_over_bytes_read=v["FETCH_SIZE"] * 1024 / rd, WRITE_SIZE_KiB_x1024_over_bytes_written=v["WRITE_SIZE"] * 1024 / wr,
                 tcc_hit_rate=v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]))
    json.dump({"what": "rocprofv3 FETCH_SIZE / WRITE_SIZE (KiB) against KNOWN byte counts on the step kernels' access pattern: one wavefront per workgroup owns one row of W floats in three "
                       "arrays (lane k <-> element k, 4 B per lane), copies them and writes a 227-float record; <W, false>: unit = blockIdx; <W, true>: the XCD-aware unit of dm_wg_unit()",
               "reading": "XCD-aware, rows of 172 B: FETCH_SIZE x 1024 = 0.51 x the bytes read (the guide's factor 1/2: 128-B requests tallied at 64 B) and WRITE_SIZE x 1024 = 1.00 x the bytes "
                          "written -- so FETCH_SIZE x 2 + WRITE_SIZE is calibrated for this pattern.  unit = blockIdx: 0.87 x = 1.70 x over-fetch (a row's first and last line are shared "
                          "with the neighbouring rows, whose workgroups sit on other XCDs with L2s of their own) and 1.08 x on the writes: the '1.41 x' of rounds 2-5",
               "kernels": cal}, open(os.path.join(dst, out + "_traffic_calibration.json"), "w"), indent=1)
except Exception as ex:
    print("calibration skipped:", repr(ex))
# TCC hit rate next to the traffic files
for sc, sfx in (("humanoid3d_walk", ""), ("humanoid3d_spinkick", "_humanoid3d_spinkick"), ("dog3d_pace", "_dog3d_pace")):
    try:
        tp = os.path.join(dst, "%s_traffic%s.json" % (out, sfx))
        t = json.load(open(tp))
        d = per_launch(os.path.join(src, "pmc_tcc" + sfx, "pmc_counter_collection.csv"), ("TCC_HIT_sum", "TCC_MISS_sum"), 64 * 2048)
        t["tcc_hit_per_launch"], t["tcc_miss_per_launch"] = d["TCC_HIT_sum"], d["TCC_MISS_sum"]
        t["tcc_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
        t["ratio_to_algorithmic"] = t["hbm_bytes_per_launch"] / t["algorithmic_bytes_per_launch"]
        t["note"] = ("FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE as KiB: both calibrated on this access pattern, " + out + "_traffic_calibration.json. "
                     "Round 6: workgroup -> env mapping XCD-aware (dm_wg_unit), no prologue spill stores")
        json.dump(t, open(tp, "w"), indent=1)
    except Exception as ex:
        print("tcc of", sc, "skipped:", repr(ex))
# flops with the matrix-core term from the MFMA counters
for sc, sfx, bname in (("humanoid3d_walk", "", "bench.json"), ("dog3d_pace", "_dog3d_pace", "bench_dog.json")):
    try:
        n = 4096
        a = per_launch(os.path.join(src, "pmc_cls_a_it10" + sfx, "pmc_counter_collection.csv"), ("SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_TRANS_F32"), 64 * 2048)
        c = per_launch(os.path.join(src, "pmc_cls_c" + sfx, "pmc_counter_collection.csv"), ("SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_MFMA", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64"), 64 * 2048)
        m = {k: v / n for k, v in {**a, **c}.items()}
        b = json.load(open(os.path.join(src, bname)))
        valu = 64.0 * (m["SQ_INSTS_VALU_ADD_F32"] + m["SQ_INSTS_VALU_MUL_F32"] + 2 * m["SQ_INSTS_VALU_FMA_F32"] + m["SQ_INSTS_VALU_TRANS_F32"])
        f64 = 64.0 * (m["SQ_INSTS_VALU_ADD_F64"] + m["SQ_INSTS_VALU_MUL_F64"] + 2 * m["SQ_INSTS_VALU_FMA_F64"])
        mfma = 512.0 * (m["SQ_INSTS_VALU_MFMA_MOPS_F32"] + m["SQ_INSTS_VALU_MFMA_MOPS_F64"])
        rate = b["value"]
        json.dump({"scene": sc, "envs": n, "kernel": b["roofline"]["kernel"], "kernel_source_sha1": bench.kernel_source_sha1(), "per_env_step_wave_instructions": m,
                   "issued_fp32_valu_lane_flops_per_env_step": valu, "issued_fp64_valu_lane_flops_per_env_step": f64, "matrix_core_flops_per_env_step": mfma, "mfma_instructions_per_env_step": m["SQ_INSTS_MFMA"],
                   "issued_flops_per_env_step": valu + f64 + mfma, "env_steps_per_s": rate, "issued_tflops": (valu + f64 + mfma) * rate / 1e12,
                   "fraction_of_fp32_vector_peak_157_3": (valu + f64 + mfma) * rate / 157.3e12,
                   "note": "ISSUED lane operations (all 64 lanes of every fp32 VALU instruction, masked / idle lanes included).  Matrix core: the step kernels DO hold MFMA instructions -- "
                           "v_mfma_f32_32x32x2_f32 (the Gram matrix A = Y^T Y: 4096 flops an issue = 8 MOPS) and v_mfma_f64_16x16x4_f64 (subtree sums of the two-per-wave kernel: 2048 flops = 4 MOPS); "
                           "`mfma_instructions_per_env_step` is SQ_INSTS_MFMA, the flops are SQ_INSTS_VALU_MFMA_MOPS_{F32,F64} x 512 (r05's note said otherwise: it was wrong).  "
                           "An upper bound on useful flops per env-step (SURVEY 8d estimated 7 M useful)"},
                  open(os.path.join(dst, "%s_flops%s.json" % (out, sfx)), "w"), indent=1)
    except Exception as ex:
        print("flops of", sc, "skipped:", repr(ex))
print(sorted(f for f in os.listdir(dst) if f.startswith(out + "_")))
