#!/usr/bin/env python3
"""Distil the round-5 instruction-class and latency counter passes (tools/gpu_round5_profile.sh: parts `classes`, `latency`) into
profiles/r05_valu_classes*.json and profiles/r05_wait_split*.json.   usage: python tools/collect_classes.py [gpurun_out/r05p] [r05]"""
import collections, csv, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r05p"); out = sys.argv[2] if len(sys.argv) > 2 else "r05"
dst = os.path.join(ROOT, "profiles")
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_sha1)


def agg(path, envs_per_launch=4096):
    """per-launch means of the step kernel's counters (summed over the XCD / SE instances), warm-up launches dropped"""
    a = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        if "k_env_step" in r["Kernel_Name"] and int(r["Grid_Size"]) >= 64 * envs_per_launch // 2:       # (bench.py's lone-wave probe launches 64-env batches of the same kernel)
            a[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            a[r["Dispatch_Id"]]["dur_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    rows = list(a.values())[2:]
    return {k: float(np.mean([v[k] for v in rows])) for k in rows[0]}, len(rows)


for scene, sfx in (("humanoid3d_walk", ""), ("dog3d_pace", "_dog3d_pace")):
    n = 4096
    try:
        cls = {}
        for it in (10, 5):
            m = {}
            for part in ("a", "b"):
                d, launches = agg(os.path.join(src, "pmc_cls_%s_it%d%s" % (part, it, sfx), "pmc_counter_collection.csv"))
                m.update({k: v / n for k, v in d.items() if k != "dur_ns"}); m["kernel_ms_" + part] = d["dur_ns"] * 1e-6
            fp = m["SQ_INSTS_VALU_FMA_F32"] + m["SQ_INSTS_VALU_ADD_F32"] + m["SQ_INSTS_VALU_MUL_F32"] + m["SQ_INSTS_VALU_TRANS_F32"]
            cls[it] = {"valu_total": m["SQ_INSTS_VALU"], "fp32_arith (fma + add + mul + trans)": fp, "fma": m["SQ_INSTS_VALU_FMA_F32"], "add": m["SQ_INSTS_VALU_ADD_F32"], "mul": m["SQ_INSTS_VALU_MUL_F32"],
                       "trans": m["SQ_INSTS_VALU_TRANS_F32"], "int32": m["SQ_INSTS_VALU_INT32"], "int64": m["SQ_INSTS_VALU_INT64"], "cvt": m["SQ_INSTS_VALU_CVT"], "mfma": m["SQ_INSTS_MFMA"],
                       "other (v_mov, v_cndmask / v_cmp, DPP moves, readlane / permlane ...: whatever the class counters do not claim)": m["SQ_INSTS_VALU"] - fp - m["SQ_INSTS_VALU_INT32"] - m["SQ_INSTS_VALU_INT64"] - m["SQ_INSTS_VALU_CVT"] - m["SQ_INSTS_MFMA"],
                       "salu": m["SQ_INSTS_SALU"], "lds": m["SQ_INSTS_LDS"], "smem": m["SQ_INSTS_SMEM"], "branch": m["SQ_INSTS_BRANCH"], "kernel_ms": m["kernel_ms_a"]}
        sweep = {k: (cls[10][k] - cls[5][k]) / 5 * 10 for k in cls[10] if k != "kernel_ms"}
        rest = {k: cls[10][k] - sweep[k] for k in sweep}
        share = lambda d: {k: d[k] / d["valu_total"] for k in d if k not in ("valu_total", "salu", "lds", "smem", "branch", "fma", "add", "mul", "trans")}
        json.dump({"scene": scene, "envs": n, "unit": "wave-level instructions per env-step (one env-step = 20 updates = 20 SPD solves + 40 substeps)", "kernel_source_sha1": bench.kernel_source_sha1(),
                   "method": "rocprofv3 --pmc class counters of the production step kernel (one launch of 4096 envs, groups 1), at 10 and at 5 Gauss-Seidel iterations; the sweeps' share is "
                             "the difference scaled to 10 iterations (the contact states of the two runs differ slightly: a ~1 % effect), `everything but the sweeps` the remainder. "
                             "The hardware has no counters for v_mov / v_cndmask / DPP / readlane separately: `other` is the VALU total minus every class that has one.",
                   "whole_kernel": cls[10], "whole_kernel_share_of_valu": share(cls[10]), "sweeps_10_iterations": sweep, "sweeps_share_of_valu": share(sweep),
                   "everything_but_the_sweeps": rest, "rest_share_of_valu": share(rest), "sweeps_fraction_of_all_valu": sweep["valu_total"] / cls[10]["valu_total"],
                   "kernel_ms_at_10_and_5_iterations": [cls[10]["kernel_ms"], cls[5]["kernel_ms"]]},
                  open(os.path.join(dst, "%s_valu_classes%s.json" % (out, sfx)), "w"), indent=1)
        print(scene, "classes ok: other share", share(cls[10]))
    except Exception as ex:
        print("classes of", scene, "skipped:", repr(ex))
    try:
        w = {"scene": scene, "envs": n, "kernel_source_sha1": bench.kernel_source_sha1(), "unit": "per env-step; cycle-type counters in quad-cycles as the SQ reports them"}
        d, _ = agg(os.path.join(src, "pmc_lat_lds" + sfx, "pmc_counter_collection.csv"))
        w["lds"] = {"instructions": d["SQ_INSTS_LDS"] / n, "average_latency_cycles": d["LdsLatency"], "in_flight_cycles": d["LdsLatency"] * d["SQ_INSTS_LDS"] / n}
        w["SQ_WAIT_ANY"] = d["SQ_WAIT_ANY"] / n; w["SQ_WAVE_CYCLES"] = d["SQ_WAVE_CYCLES"] / n
        d, _ = agg(os.path.join(src, "pmc_lat_smem" + sfx, "pmc_counter_collection.csv"))
        w["smem"] = {"instructions": d["SQ_INSTS_SMEM"] / n, "average_latency_cycles": d["SmemLatency"], "in_flight_cycles": d["SmemLatency"] * d["SQ_INSTS_SMEM"] / n}
        d, _ = agg(os.path.join(src, "pmc_lat_vmem" + sfx, "pmc_counter_collection.csv"))
        w["vmem"] = {"instructions": d["SQ_INSTS_VMEM"] / n, "average_latency_cycles": d["VmemLatency"], "in_flight_cycles": d["VmemLatency"] * d["SQ_INSTS_VMEM"] / n}
        d, _ = agg(os.path.join(src, "pmc_lds" + sfx, "pmc_counter_collection.csv"))
        w["lds_detail"] = {k: v / n for k, v in d.items() if k != "dur_ns"}
        w["lds_detail"]["bank_conflict_fraction_of_active"] = d["SQ_LDS_BANK_CONFLICT"] / max(1.0, d["SQ_LDS_IDX_ACTIVE"])
        w["lds_detail"]["valu_lane_utilisation"] = d["SQ_THREAD_CYCLES_VALU"] / max(1.0, 64.0 * d["SQ_ACTIVE_INST_VALU"])
        json.dump(w, open(os.path.join(dst, "%s_wait_split%s.json" % (out, sfx)), "w"), indent=1)
        print(scene, "wait split:", json.dumps({k: w[k] for k in ("lds", "smem", "vmem", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES")}))
    except Exception as ex:
        print("wait split of", scene, "skipped:", repr(ex))
