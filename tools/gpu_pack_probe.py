#!/usr/bin/env python3
"""A few control steps of 4096 humanoids with the wave packing given by PACK (1 or 2): run under rocprofv3 --pmc."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepmimic_amd import model
from deepmimic_amd.core import BatchEnv
env = BatchEnv(model.load_asset("humanoid3d_walk"), 4096, seed=1234, test_mode=True, wave_packing=int(os.environ.get("PACK", "1")))
env.reset(); env.bench_rollout(10, 1)
for k in range(6):
    print(env.bench_rollout(0, 1))
