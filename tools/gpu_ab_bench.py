"""A/B on ONE box (the boxes of the pool differ by several percent): times the step kernel of two builds of libdm_hip.so
alternately.  usage: python tools/gpu_ab_bench.py libA.so libB.so [scene] [envs]   (paths relative to the repo root)"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepmimic_amd import core, model, streams  # noqa: E402


def load_raw(path):
    lib = C.CDLL(path)
    lib.dm_last_error.restype = C.c_char_p
    lib.dm_motion_duration.restype = C.c_double
    lib.dm_motion_duration.argtypes = [C.c_void_p]
    core._libs[path] = lib
    return lib


def main():
    a, b = [os.path.join(ROOT, p) for p in sys.argv[1:3]]
    scene = sys.argv[3] if len(sys.argv) > 3 else "humanoid3d_walk"
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
    t = model.load_asset(scene)
    envs = {}
    for tag, path in (("A", a), ("B", b)):
        load_raw(path)
        env = core.BatchEnv(t, n, seed=1234, lib_path=path, test_mode=True)
        env.reset(kin_times=streams.reset_phase(np.arange(n), env.duration))
        env.bench_rollout(60, 1)
        envs[tag] = env
    res = {"A": [], "B": []}
    for rep in range(5):
        for tag in ("A", "B"):
            res[tag].append(envs[tag].bench_rollout(0, 100) / 100)
    out = {tag: {"lib": os.path.relpath(p, ROOT), "kernel_ms_median": float(np.median(res[tag])), "kernel_ms_all": res[tag],
                 "env_steps_per_s": n / (float(np.median(res[tag])) * 1e-3)} for tag, p in (("A", a), ("B", b))}
    out["B_over_A_time"] = out["B"]["kernel_ms_median"] / out["A"]["kernel_ms_median"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
