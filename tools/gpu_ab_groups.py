#!/usr/bin/env python3
"""does splitting a 4096-env batch into G independent groups -- each its own dm_ctx on its own HIP stream, so that a
group's control step k + 1 starts when ITS slowest wave is done instead of the whole batch's -- recover the wave-time tail of a one-round
launch?  Every group runs the C rollout loop (dm_bench_rollout) from its own host thread (ctypes drops the GIL); throughput = all envs x
steps / wall time between a common start and the last group's end.  G = 1 through the same harness is the baseline.
usage: python tools/gpu_ab_groups.py [scene] [envs] [steps] [lib]"""
import json, os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from deepmimic_amd import core, model, streams

scene = sys.argv[1] if len(sys.argv) > 1 else "humanoid3d_walk"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
lib = os.path.join(ROOT, sys.argv[4]) if len(sys.argv) > 4 else None
t = model.load_asset(scene)
out = {"scene": scene, "envs": n, "steps": steps, "groups": {}}
for G in (1, 2, 4, 8, 16, 32):
    per = n // G
    envs = []
    for g in range(G):
        e = core.BatchEnv(t, per, seed=1234, test_mode=True, env_id_offset=g * per, lib_path=lib)
        e.reset(kin_times=streams.reset_phase(np.arange(g * per, (g + 1) * per), e.duration))
        e.bench_rollout(60, 1)
        envs.append(e)
    reps = []
    for rep in range(3):
        gate = threading.Barrier(G + 1)
        ms = [0.0] * G
        def run(g):
            gate.wait()
            ms[g] = envs[g].bench_rollout(0, steps)
        th = [threading.Thread(target=run, args=(g,)) for g in range(G)]
        for x in th: x.start()
        gate.wait(); t0 = time.perf_counter()
        for x in th: x.join()
        wall = time.perf_counter() - t0
        reps.append({"wall_s": wall, "env_steps_per_s": n * steps / wall, "group_ms_per_step_max": max(ms) / steps, "group_ms_per_step_mean": float(np.mean(ms)) / steps})
    best = max(reps, key=lambda r: r["env_steps_per_s"])
    out["groups"][str(G)] = best
    print("G %2d  %7.0f env-steps/s (wall)  slowest group %.4f ms/step, mean %.4f" % (G, best["env_steps_per_s"], best["group_ms_per_step_max"], best["group_ms_per_step_mean"]), file=sys.stderr)
    for e in envs: e.close()
print(json.dumps(out))
