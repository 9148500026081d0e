#!/usr/bin/env python3
"""Per-phase shader-cycle breakdown of one control step (profiling tap `prof`, lane 0 of every wave)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from deepmimic_amd import model
from deepmimic_amd.core import BatchEnv
PH = ["kin_update+latch", "spd.kinematics", "spd.dynamics", "spd.chol+solve", "spd.err/clamp|sub.pre", "sub.kinematics", "sub.dynamics",
      "sub.chol+solve+vstar", "sub.collision", "sub.rows(J,Y)", "sub.A", "sub.PGS", "sub.backsolve+integrate", "emit(+reset)", "store", "load+action"]
name = sys.argv[1] if len(sys.argv) > 1 else "humanoid3d_walk"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
env = BatchEnv(model.load_asset(name), n, seed=1, test_mode=True, wave_packing=int(os.environ.get('PACK', '0')))
env.reset()
for _ in range(10):
    env.step(None, 1 / 600, 20, open_loop=True, auto_reset=True)
env.probe(3, 1 / 600)
p = env.debug("prof")
tot = p.sum(1)
print(json.dumps({"scene": name, "envs": n, "cycles_per_env_step_mean": float(tot.mean()), "max": float(tot.max()),
                  "phases": {PH[i]: [float(p[:, i].mean()), float(100 * p[:, i].sum() / tot.sum())] for i in range(16)}}, indent=1))
# distribution of per-wave totals (tail diagnostics): the last wave of a SIMD runs alone and latency-bound
q = np.percentile(tot, [1, 10, 50, 90, 99])
print(json.dumps({"wave_total_cycles_percentiles_1_10_50_90_99": [float(x) for x in q], "cv": float(tot.std() / tot.mean())}))
