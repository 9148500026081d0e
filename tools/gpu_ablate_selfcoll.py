#!/usr/bin/env python
"""What self collision costs per scene: the fixed-action rollout with the pair list empty (`self_collision=False`) against the default, same box,
interleaved repeats.  (Not a parity mode: without the pairs the rare link-link contacts vanish.)  Prints one JSON object; run on the GPU box."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepmimic_amd import model, streams             # noqa: E402
from deepmimic_amd.core import BatchEnv              # noqa: E402

out = {}
n = 4096
for scene in ("dog3d_pace", "humanoid3d_walk"):
    t = model.load_asset(scene)
    envs = {sc: BatchEnv(t, n, seed=1, test_mode=True, self_collision=sc) for sc in (True, False)}
    for e in envs.values():
        e.reset(kin_times=streams.reset_phase(np.arange(n), e.duration)); e.bench_rollout(60, 0)
    ms = {True: [], False: []}
    for rep in range(4):
        for sc in (True, False):
            ms[sc].append(envs[sc].bench_rollout(0, 100) / 100)
    out[scene] = {"ms_per_step_self_collision_on": float(np.median(ms[True])), "ms_per_step_self_collision_off": float(np.median(ms[False])),
                  "all_on": ms[True], "all_off": ms[False]}
    out[scene]["share_of_step"] = 1.0 - out[scene]["ms_per_step_self_collision_off"] / out[scene]["ms_per_step_self_collision_on"]
    for e in envs.values():
        e.close()
print(json.dumps(out))
