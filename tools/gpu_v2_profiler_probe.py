#!/usr/bin/env python3
"""Which stage of a DM-physics v2 context faults under `rocprofv3 --pmc` (NOTES.md, round 5): prints a line after every stage.
    rocprofv3 --pmc SQ_WAVES --kernel-trace -d /tmp/p -- python tools/gpu_v2_profiler_probe.py [physics] [envs] [wave_packing]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepmimic_amd import model          # noqa: E402
from deepmimic_amd.core import BatchEnv  # noqa: E402

physics = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
pack = int(sys.argv[3]) if len(sys.argv) > 3 else 0
t = model.load_asset("humanoid3d_walk")
env = BatchEnv(t, n, precision=32, physics=physics, wave_packing=pack, seed=1234, test_mode=True)
print("created", flush=True)
env.reset()
env.get_state()
print("reset", flush=True)
env.step(None, 1.0 / 600.0, 1)
env.get_state()
print("one update", flush=True)
env.step(None, 1.0 / 600.0, 20)
env.get_state()
print("one control step", flush=True)
print("bench_rollout", env.bench_rollout(0, 3), flush=True)
