#!/usr/bin/env python3
"""Constraint rows per substep under three action regimes (open-loop tracking, zero actions, a random-init policy's mean action):
how often a character exceeds the 32 rows the two-per-wave kernel keeps in registers (then the pair falls back)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from deepmimic_amd import model
from deepmimic_amd.core import BatchEnv
SCENE = os.environ.get("SCENE", "humanoid3d_walk")
t = model.load_asset(SCENE); n = 4096
e0 = BatchEnv(t, 1); amean = (-e0.offsets_scales()["action_offset"]).astype(np.float32); e0.close()
for label, acts, ol in (("open-loop", None, True), ("zeros", np.zeros((n, 28), np.float32), False), ("a_mean", np.tile(amean, (n, 1)), False)):
    env = BatchEnv(t, n, seed=1234, test_mode=True); env.reset()
    Rs = []
    for k in range(48):
        env.step(acts, 1 / 600, 20, open_loop=ol, auto_reset=True)
        if k >= 16 and k % 4 == 3:
            env.probe(1, 1 / 1200); Rs.append(env.debug("rows")[:, 0].copy())
    R = np.concatenate(Rs)
    pair = np.maximum(R[0::2], R[1::2])
    print("%-10s rows mean %.1f p50 %d p90 %d p99 %d max %d | chars > 32 rows: %.2f %% | pairs with a char > 32: %.2f %%" %
          (label, R.mean(), np.percentile(R, 50), np.percentile(R, 90), np.percentile(R, 99), R.max(), 100 * (R > 32).mean(), 100 * (pair > 32).mean()))
    env.close()
