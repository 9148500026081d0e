#!/usr/bin/env python3
"""Distribution of constraint rows / contacts per substep in the benchmark workload (open-loop, auto-reset)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from deepmimic_amd import model
from deepmimic_amd.core import BatchEnv
name = sys.argv[1] if len(sys.argv) > 1 else "humanoid3d_walk"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
env = BatchEnv(model.load_asset(name), n, seed=1234, test_mode=True)
env.reset()
out = {"scene": name, "envs": n, "samples": []}
for k in range(40):
    env.step(None, 1 / 600, 20, open_loop=True, auto_reset=True)
    if k % 8 == 7:
        env.probe(1, 1 / 1200)
        rows = env.debug("rows")
        R, nc = rows[:, 0], rows[:, 1]
        out["samples"].append({"step": k + 1, "rows_mean": float(R.mean()), "rows_p50": float(np.percentile(R, 50)), "rows_p90": float(np.percentile(R, 90)),
                               "rows_max": float(R.max()), "contacts_mean": float(nc.mean()), "frac_no_contact": float((nc == 0).mean())})
print(json.dumps(out, indent=1))
