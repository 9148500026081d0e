#!/usr/bin/env python3
"""Closed-loop diagnostics of one scene as ONE JSON document (profiles/r06_closed_loop_<scene>.json): a random-init on-device policy drives 4096 envs;
reported: episode ends and invalid episodes per env-step, the share of pair-substeps beyond 32 rows (borrowed lanes | 64-lane fallback), the rows-per-substep histogram of the
policy-made state distribution, the per-wave cycle totals of one profiled control step with the phases that separate the slowest waves from the median ones, and the host-timed
closed-loop and open-loop step of the same context.   usage: SCENE=humanoid3d_spinkick python tools/gpu_closed_loop_diag.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from deepmimic_amd import model, streams
from deepmimic_amd.core import BatchEnv
from deepmimic_amd.policy import Policy, random_weights
PH = ["kin_update+latch", "spd.kinematics", "spd.dynamics", "spd.chol+solve", "spd.err/clamp|sub.pre", "sub.kinematics", "sub.dynamics",
      "sub.chol+solve+vstar", "sub.collision", "sub.rows(J,Y)", "sub.A", "sub.PGS", "sub.backsolve+integrate", "emit(+reset)", "store", "load+action"]
SCENE = os.environ.get("SCENE", "humanoid3d_walk")
t = model.load_asset(SCENE); n = 4096
env = BatchEnv(t, n, seed=1234, test_mode=True)
env.reset(kin_times=streams.reset_phase(np.arange(n), env.duration))
duo = env.J <= 15 and t.goal_kind != 5
epw = 2 if duo else 1
open_ms = env.bench_rollout(60, 100) / 100
offs = env.offsets_scales()
w = random_weights(env.S, env.A, seed=0)
w["s_mean"] = -offs["state_offset"].astype(np.float32); w["s_std"] = (1.0 / offs["state_scale"]).astype(np.float32)
w["a_mean"] = -offs["action_offset"].astype(np.float32); w["a_std"] = (1.0 / offs["action_scale"]).astype(np.float32)
pol = Policy(w)
dev = torch.device("cuda")
f32, i32 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.int32, device=dev)
st = torch.zeros((n, env.S), **f32); ac = torch.zeros((n, env.A), **f32); rw = torch.zeros(n, **f32)
tm = torch.zeros(n, **i32); vd = torch.zeros(n, **i32); en = torch.zeros(n, **i32)
ptrs = (st.data_ptr(), rw.data_ptr(), tm.data_ptr(), vd.data_ptr(), en.data_ptr())
strm = env.own_stream()
env.step_device(0, *ptrs, n_updates=0); env.synchronize()


def loop(k0, k1):
    for k in range(k0, k1):
        pol.forward_device(st.data_ptr(), n, ac.data_ptr(), 0, sample=True, seed=1, step=k, stream=strm)
        env.step_device(ac.data_ptr(), *ptrs, timestep=1.0 / 600, n_updates=20, auto_reset=True)


loop(0, 60); env.synchronize()
c0 = np.array([env.debug("fallback").sum(), env.debug("borrowed").sum()])
ends, inval = 0.0, 0.0
t0 = time.perf_counter(); loop(60, 260); env.synchronize(); closed_ms = 1e3 * (time.perf_counter() - t0) / 200
c1 = np.array([env.debug("fallback").sum(), env.debug("borrowed").sum()])
for k in range(260, 280):
    loop(k, k + 1); env.synchronize()
    ends += float(en.float().mean().item()) / 20; inval += float(1.0 - vd.float().mean().item()) / 20
share = (c1 - c0) / epw / ((n // epw) * 40 * 200)
out = {"scene": SCENE, "envs": n, "kernel": "k_env_step_duo (two characters per wavefront)" if duo else "k_env_step (one character per wavefront)",
       "policy": "%d -> 1024 -> 512 -> %d, random init, sampled actions (dm_policy_forward)" % (env.S, env.A),
       "open_loop_ms_per_step_one_launch": open_ms, "closed_loop_ms_per_step_one_launch": closed_ms, "closed_over_open_rate": open_ms / closed_ms,
       "episode_ends_per_env_step": ends, "invalid_episodes_per_env_step": inval,
       "share_of_pair_substeps_on_the_64_lane_fallback": float(share[0]), "share_of_pair_substeps_on_borrowed_lanes": float(share[1])}
# one profiled control step on the policy's actions (tap build of the kernel: a ranking of phases, not the production timing)
env.probe(4, 1 / 600)
p = env.debug("prof"); wv = p[0::2] if duo else p
tot = wv.sum(1); order = np.argsort(tot)
med = order[len(order) // 2 - 50: len(order) // 2 + 50]; top = order[-20:]
out["wave_cycles"] = {"mean": float(tot.mean()), "median": float(np.median(tot)), "p90": float(np.percentile(tot, 90)), "p99": float(np.percentile(tot, 99)), "max": float(tot.max()),
                      "max_over_median": float(tot.max() / np.median(tot)),
                      "waves_by_time_over_median": {"<1.1": int((tot < 1.1 * np.median(tot)).sum()), "1.1-1.2": int(((tot >= 1.1 * np.median(tot)) & (tot < 1.2 * np.median(tot))).sum()),
                                                    "1.2-1.5": int(((tot >= 1.2 * np.median(tot)) & (tot < 1.5 * np.median(tot))).sum()), ">=1.5": int((tot >= 1.5 * np.median(tot)).sum())}}
out["phase_kcycles_median_waves_vs_slowest_20"] = {PH[i]: [float(wv[med, i].mean() / 1e3), float(wv[top, i].mean() / 1e3)] for i in range(16)}
# rows per substep of the policy-made states: one more substep with the latched torques through the tap build
env.probe(1, 1 / 1200)
r = env.debug("rows"); R, NC = r[:, 0].astype(int), r[:, 1].astype(int)
hist = np.bincount(np.clip(R, 0, 64), minlength=65)
out["rows_per_character"] = {"mean": float(R.mean()), "p50": int(np.percentile(R, 50)), "p90": int(np.percentile(R, 90)), "p99": int(np.percentile(R, 99)), "max": int(R.max()),
                             "histogram_by_4": {"%d-%d" % (4 * i, 4 * i + 3): int(hist[4 * i:4 * i + 4].sum()) for i in range(16)}, "more_than_32": float((R > 32).mean())}
if duo:
    pr = np.maximum(R[0::2], R[1::2]); hv = np.flatnonzero(R > 32)
    out["pairs"] = {"with_a_character_beyond_32_rows": float((pr > 32).mean()), "both_beyond_32": float(((R[0::2] > 32) & (R[1::2] > 32)).mean()),
                    "heavy_plus_partner_within_64_rows": float(((R[hv] + R[hv ^ 1]) <= 64).mean()) if hv.size else None,
                    "contacts_of_the_heavy_characters": {int(k): int(v) for k, v in zip(*np.unique(NC[R > 32], return_counts=True))}}
print(json.dumps(out, indent=1))
