#!/usr/bin/env python3
"""Tail diagnostics of the step kernel: per-wave cycle totals (prof tap), which phases separate the slowest waves from the
median ones, and how throughput changes when more than one round of waves is resident (8192 / 16384 envs)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from deepmimic_amd import model
from deepmimic_amd.core import BatchEnv
PH = ["kin_update+latch", "spd.kinematics", "spd.dynamics", "spd.chol+solve", "spd.err/clamp|sub.pre", "sub.kinematics", "sub.dynamics",
      "sub.chol+solve+vstar", "sub.collision", "sub.rows(J,Y)", "sub.A", "sub.PGS", "sub.backsolve+integrate", "emit(+reset)", "store", "load+action"]
t = model.load_asset("humanoid3d_walk")
n = 4096
env = BatchEnv(t, n, seed=1, test_mode=True)
env.reset()
for _ in range(40):
    env.step(None, 1 / 600, 20, open_loop=True, auto_reset=True)
env.probe(3, 1 / 600)
p = env.debug("prof")
w = p[0::2]                      # duo: lane 0 of the wave writes env 2b's slot
tot = w.sum(1)
order = np.argsort(tot)
med = order[len(order) // 2 - 50: len(order) // 2 + 50]; top = order[-20:]
print("waves", len(tot), "mean %.3fM median %.3fM p90 %.3fM p99 %.3fM max %.3fM" % tuple(x / 1e6 for x in (tot.mean(), np.median(tot), np.percentile(tot, 90), np.percentile(tot, 99), tot.max())))
print("phase: median-waves mean | slowest-20 mean | delta (k cycles)")
for i in range(16):
    a, b = w[med, i].mean(), w[top, i].mean()
    print("  %-26s %9.0f %9.0f %+9.0f" % (PH[i], a / 1e3, b / 1e3, (b - a) / 1e3))
for nn in (4096, 8192, 16384):
    e2 = BatchEnv(t, nn, seed=1, test_mode=True); e2.reset()
    ms = e2.bench_rollout(30, 100) / 100
    print("envs %5d  %.3f ms/step  %.3f M env-steps/s" % (nn, ms, nn / ms / 1e3))
    e2.close()
