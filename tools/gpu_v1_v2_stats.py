#!/usr/bin/env python3
"""How much does the choice between DM-physics v1 and v2 matter?  4096 humanoid3d_walk envs x 300 control steps of open-loop clip
tracking with auto-reset under each version (same seeds, same start phases): imitation reward, how long an episode lasts before the character falls,
the share of episodes that end in a fall, and the throughput of both wave packings.  -> JSON on stdout (profiles/r04_v1_v2_stats.json)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from deepmimic_amd import model, streams
from deepmimic_amd.core import BatchEnv

n, steps = 4096, 300
out = {"envs": n, "steps": steps, "workload": "open-loop clip tracking (stream A1), auto-reset with the early episode end, episode timer at its annealed end"}
for scene in ("humanoid3d_walk", "humanoid3d_spinkick", "dog3d_pace"):
    t = model.load_asset(scene)
    res = {}
    for phys in (1, 2):
        env = BatchEnv(t, n, seed=1234, test_mode=True, physics=phys)
        env.reset(kin_times=streams.reset_phase(np.arange(n), env.duration))
        rew, falls, ends, ep_len, cur = [], 0, 0, [], np.zeros(n)
        for k in range(steps):
            o = env.step(None, 1.0 / 600, 20, open_loop=True, auto_reset=True)
            rew.append(float(o["reward"].mean()))
            cur += 1
            e = o["episode_end"] != 0
            falls += int((o["terminate"][e] == 1).sum()); ends += int(e.sum())
            ep_len.extend(cur[e].tolist()); cur[e] = 0
        rate = {}
        for pk in (1, 2) if t.num_joints <= 15 else (1,):
            b = BatchEnv(t, n, seed=1234, test_mode=True, physics=phys, wave_packing=pk)
            b.reset(kin_times=streams.reset_phase(np.arange(n), b.duration))
            b.bench_rollout(60, 0)
            rate["wave_packing_%d" % pk] = n / (b.bench_rollout(0, 100) / 100 * 1e-3)
            b.close()
        res["v%d" % phys] = {"mean_reward": float(np.mean(rew)), "mean_reward_last_100_steps": float(np.mean(rew[-100:])), "episodes_ended": ends,
                             "ended_by_fall": falls, "fall_share": falls / max(1, ends), "mean_episode_steps": float(np.mean(ep_len)) if ep_len else None,
                             "median_episode_steps": float(np.median(ep_len)) if ep_len else None, "env_steps_per_s": rate}
        env.close()
    out[scene] = res
    print(scene, json.dumps(res), file=sys.stderr)
print(json.dumps(out))
