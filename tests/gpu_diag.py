#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (uses the oracle): GPU-side rollout parity summaries (device vs oracle) per scene / precision -> JSON lines."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))   # ROOT = repo root (this file lives in tests/)
import numpy as np
import parity_common as pc

lib = os.path.join(ROOT, "deepmimic_amd", "csrc", "libdm_hip.so")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for name in ["humanoid3d_walk", "humanoid3d_spinkick", "dog3d_pace"]:
    for prec in (64, 32):
        dr, ds, ok = pc.rollout_compare(name, prec, lib, steps=steps)
        first_bad = int(np.argmax(dr > 1e-4)) if (dr > 1e-4).any() else -1
        print(json.dumps({"scene": name, "precision": prec, "steps": steps, "reward_mae": float(dr.mean()), "reward_max_abs": float(dr.max()),
                          "first_step_over_1e-4": first_bad, "state_max_abs": float(ds.max()), "flags_equal": bool(ok),
                          "dr_by_50": [float(dr[i:i + 50].max()) for i in range(0, steps, 50)]}))
# reset detail for spinkick fp64
t, o, env = pc.make_pair("humanoid3d_spinkick", 4, 64, lib)
times = np.array([0.0, 0.21, 0.8 * o.duration, 2.3 * o.duration])
env.reset(kin_times=times, max_times=np.inf)
st = env.get_state()
for e, tt in enumerate(times):
    o.reset(tt); p, v = o.sim_state()
    d = np.abs(st["vel"][e] - v); i = int(np.argmax(d))
    print("reset", e, tt, "pose", np.abs(st["pose"][e] - p).max(), "vel", d.max(), "idx", i, st["vel"][e][i], v[i])
