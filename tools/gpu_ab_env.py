"""A/B on ONE box of two settings of an environment switch that dm_create reads (DM_TREE, DM_DUO, ...): two contexts of the same
libdm_hip.so, timed alternately.  usage: python tools/gpu_ab_env.py VAR valueA valueB [scene] [envs] [steps]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepmimic_amd import core, model, streams  # noqa: E402


def main():
    var, va, vb = sys.argv[1:4]
    scene = sys.argv[4] if len(sys.argv) > 4 else "humanoid3d_walk"
    n = int(sys.argv[5]) if len(sys.argv) > 5 else 4096
    steps = int(sys.argv[6]) if len(sys.argv) > 6 else 100
    t = model.load_asset(scene)
    envs = {}
    for tag, val in (("A", va), ("B", vb)):
        os.environ[var] = val
        env = core.BatchEnv(t, n, seed=1234, test_mode=True)
        env.reset(kin_times=streams.reset_phase(np.arange(n), env.duration))
        env.bench_rollout(60, 1)
        envs[tag] = env
    res = {"A": [], "B": []}
    for rep in range(5):
        for tag in ("A", "B"):
            res[tag].append(envs[tag].bench_rollout(0, steps) / steps)
    out = {tag: {var: val, "kernel_ms_median": float(np.median(res[tag])), "kernel_ms_all": res[tag],
                 "env_steps_per_s": n / (float(np.median(res[tag])) * 1e-3)} for tag, val in (("A", va), ("B", vb))}
    out["scene"] = scene; out["envs"] = n
    out["B_over_A_time"] = out["B"]["kernel_ms_median"] / out["A"]["kernel_ms_median"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
