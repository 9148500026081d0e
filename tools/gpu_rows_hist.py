#!/usr/bin/env python3
"""Constraint rows per substep of a scene in open-loop tracking (probe after every 4th control step): how many rows the sweep visits, and how
many of them lie beyond the rows a kernel class keeps in registers.   usage: python tools/gpu_rows_hist.py [scene]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from deepmimic_amd import model
from deepmimic_amd.core import BatchEnv
scene = sys.argv[1] if len(sys.argv) > 1 else "dog3d_pace"
t = model.load_asset(scene); n = 4096
env = BatchEnv(t, n, seed=1234, test_mode=True); env.reset()
Rs = []
for k in range(48):
    env.step(None, 1 / 600, 20, open_loop=True, auto_reset=True)
    if k >= 16 and k % 4 == 3:
        env.probe(1, 1 / 1200); Rs.append(env.debug("rows").copy())
R = np.concatenate(Rs)
rows, nc = R[:, 0].astype(np.int64), R[:, 1].astype(np.int64)
hist = np.bincount(rows, minlength=65)
print("%s: rows mean %.1f p10 %d p50 %d p90 %d p99 %d max %d; contacts mean %.1f max %d" % (scene, rows.mean(), *np.percentile(rows, [10, 50, 90, 99]), rows.max(), nc.mean(), nc.max()))
for cap in (32, 40, 48, 56):
    print("  rows beyond %d: %.1f %% of substeps, %.2f row visits per substep on average" % (cap, 100 * (rows > cap).mean(), np.maximum(rows - cap, 0).mean()))
print("  histogram (rows: share %%):", " ".join("%d:%.1f" % (i, 100 * h / hist.sum()) for i, h in enumerate(hist) if h))
