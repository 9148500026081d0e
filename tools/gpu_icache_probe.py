#!/usr/bin/env python3
"""Run a few control steps of 4096 envs, either phase-randomised (MODE=random) or all identical (MODE=identical)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from deepmimic_amd import model
from deepmimic_amd.core import BatchEnv
env = BatchEnv(model.load_asset("humanoid3d_walk"), 4096, seed=1234, test_mode=True)
if os.environ.get("MODE", "random") == "identical":
    env.reset(kin_times=0.3, max_times=np.inf)
    for k in range(8):
        env.reset(kin_times=0.3 + 0.01 * k, max_times=np.inf)
        print(env.bench_rollout(0, 1, auto_reset=False, open_loop=True))
else:
    env.reset(); env.bench_rollout(10, 1)
    for k in range(8):
        print(env.bench_rollout(0, 1))
