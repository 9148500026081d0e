"""Throughput of the imitate scene with `--enable_rand_perturbs` on (the AMP instantiation of the step kernel carries the perturbation code) next to
the plain scene, on ONE box, alternating.  usage: python tools/gpu_perturb_bench.py [scene] [envs]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepmimic_amd import core, model, streams  # noqa: E402


def main():
    scene = sys.argv[1] if len(sys.argv) > 1 else "humanoid3d_walk"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    envs = {}
    for tag in ("plain", "perturbed", "perturbed_hard"):
        t = model.load_asset(scene)
        if tag != "plain":
            c = t.cfg
            c.enable_rand_perturbs = True
            # robustness-training settings: a 50-100 N push (the reference's default magnitudes and durations) every 1-2 s; "hard": 200-400 N every 0.5-1 s
            c.perturb_time_min, c.perturb_time_max = (1.0, 2.0) if tag == "perturbed" else (0.5, 1.0)
            if tag == "perturbed_hard":
                c.min_perturb, c.max_perturb = 200.0, 400.0
        env = core.BatchEnv(t, n, seed=1234, test_mode=True)
        env.reset(kin_times=streams.reset_phase(np.arange(n), env.duration))
        env.bench_rollout(60, 1)
        envs[tag] = env
    res = {k: [] for k in envs}
    for rep in range(3):
        for tag, env in envs.items():
            res[tag].append(env.bench_rollout(0, 100) / 100)
    out = {"scene": scene, "envs": n}
    for tag in envs:
        ms = float(np.median(res[tag]))
        out[tag] = {"kernel_ms_median": ms, "env_steps_per_s": n / (ms * 1e-3)}
    rows = envs["perturbed_hard"].get_perturb_state()
    out["perturbed_hard"]["envs_with_a_force_acting"] = int((rows[:, 3] > 0).sum() + 0)
    out["perturbed_over_plain_time"] = out["perturbed"]["kernel_ms_median"] / out["plain"]["kernel_ms_median"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
