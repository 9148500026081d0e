"""GPU-side parity report (run on the MI355X box): the BASELINE.json parity metric measured so that it means something.

For each scene: `--steps` control steps of `--envs` envs, open-loop mocap tracking (stream A1), THROUGH auto-resets that the
oracle mirrors draw for draw (tests/parity_common.auto_reset_rollout_compare), so every transition is a live one.  Reports
the reward error over live steps only (MAE, p99, max, where the max occurred), the state-vector error, how many steps were
live, for the fp64 algorithm build and the fp32 production kernels (both wave packings for the biped), free-running.
Writes one JSON object to stdout."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_common as pc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--envs", type=int, default=8)
    ap.add_argument("--scenes", default="humanoid3d_walk,humanoid3d_spinkick,dog3d_pace")
    a = ap.parse_args()
    lib = os.path.join(ROOT, "deepmimic_amd", "csrc", "libdm_hip.so")
    out = {}
    for name in a.scenes.split(","):
        for prec, pack in ((64, 0), (32, 2), (32, 1)):
            if name.startswith("dog") and pack == 1:
                continue
            dr, ds, alive, resets, ok = pc.auto_reset_rollout_compare(name, prec, lib, a.steps, a.envs, seed=11, wave_packing=pack)
            live = dr[alive]
            k, e = np.unravel_index(np.argmax(np.where(alive, dr, 0)), dr.shape)
            out["%s/fp%d/pack%d" % (name, prec, pack)] = dict(
                steps=a.steps, envs=a.envs, transitions=int(dr.size), live=int(alive.sum()), resets=int(resets), flags_ok=bool(ok),
                reward_mae_live=float(live.mean()), reward_p99_live=float(np.quantile(live, 0.99)), reward_max_live=float(live.max()),
                reward_max_at=[int(k), int(e)], frac_live_over_1e4=float((live > 1e-4).mean()),
                state_rel_mean=float(ds.mean()), state_rel_p99=float(np.quantile(ds, 0.99)), state_rel_max=float(ds.max()))
            # teacher-forced: one control step from the oracle's state, live steps only -- the per-step precision of the kernel
            dr, ds, alive, ok = pc.stepwise_live_compare(name, prec, lib, a.steps, a.envs, seed=12, wave_packing=pack)
            live, sl = dr[alive], ds[alive]
            k, e = np.unravel_index(np.argmax(np.where(alive, dr, 0)), dr.shape)
            srt = np.sort(live)
            out["%s/fp%d/pack%d/stepwise" % (name, prec, pack)] = dict(
                transitions=int(dr.size), live=int(alive.sum()), flags_ok=bool(ok),
                reward_mae_live=float(live.mean()), reward_p99_live=float(np.quantile(live, 0.99)), reward_p999_live=float(np.quantile(live, 0.999)),
                reward_max_live=float(live.max()), reward_max_at=[int(k), int(e)], n_live_over_1e4=int((live > 1e-4).sum()),
                worst5=[float(x) for x in srt[-5:]],
                state_rel_mean=float(sl.mean()), state_rel_p99=float(np.quantile(sl, 0.99)), state_rel_max=float(sl.max()))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
