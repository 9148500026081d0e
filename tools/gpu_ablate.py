#!/usr/bin/env python3
"""What a phase costs in the PRODUCTION kernel (no taps, no spills): the same library timed with a runtime knob turned -- Gauss-Seidel
iterations 10 -> 1 (the sweeps), self collision off (pair tests + the rows of self contacts), contact cap 20 -> 4 (rows).  Every
measurement is the first `steps` control steps after a fresh reset to the same phases, so the states under the variants stay close.
usage: python tools/gpu_ablate.py [scene] [steps]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepmimic_amd import core, model, streams  # noqa: E402
scene = sys.argv[1] if len(sys.argv) > 1 else "dog3d_pace"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n = 4096
t = model.load_asset(scene)
variants = {"base": {}, "solver_iters_1": {"solver_iters": 1}, "no_self_collision": {"self_collision": False}, "max_contacts_4": {"max_contacts": 4},
            "solver_iters_1 + no_self_collision": {"solver_iters": 1, "self_collision": False}}
envs = {k: core.BatchEnv(t, n, seed=1234, test_mode=True, **kw) for k, kw in variants.items()}
res = {k: [] for k in envs}
for rep in range(6):
    for k, env in envs.items():
        env.reset(kin_times=streams.reset_phase(np.arange(n), env.duration))
        env.bench_rollout(0, 1)                                  # (first launch after a reset: warm)
        env.reset(kin_times=streams.reset_phase(np.arange(n), env.duration))
        res[k].append(env.bench_rollout(0, steps) / steps)
out = {"scene": scene, "envs": n, "steps_after_reset": steps, "kernel_ms": {k: float(np.median(v)) for k, v in res.items()}}
b = out["kernel_ms"]["base"]
out["sweeps_share"] = (b - out["kernel_ms"]["solver_iters_1"]) * 10 / 9 / b
out["self_collision_share"] = (b - out["kernel_ms"]["no_self_collision"]) / b
print(json.dumps(out))
