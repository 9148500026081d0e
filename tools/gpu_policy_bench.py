#!/usr/bin/env python
"""Time the on-device policy (dm_policy.h) alone and inside the closed 30 Hz loop (policy -> control step), 4096 humanoids.
Prints one JSON object; run on the GPU box."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepmimic_amd import model                      # noqa: E402
from deepmimic_amd.core import BatchEnv              # noqa: E402
from deepmimic_amd.policy import Policy, random_weights   # noqa: E402

n = int(os.environ.get("ENVS", "4096")); iters = 200
t = model.load_asset("humanoid3d_walk")
env = BatchEnv(t, n, seed=1)
# policy and env on ONE explicit stream (torch's default stream has the null handle, which dm_set_stream reads as "the ctx's own stream":
# the two would then run unordered against each other)
tstream = torch.cuda.Stream(); torch.cuda.set_stream(tstream)
stream = tstream.cuda_stream
env.set_stream(stream); env.reset()
offs = env.offsets_scales()
w = random_weights(env.S, env.A, seed=0)
w["s_mean"] = -offs["state_offset"].astype(np.float32); w["s_std"] = (1.0 / offs["state_scale"]).astype(np.float32)
w["a_mean"] = -offs["action_offset"].astype(np.float32); w["a_std"] = (1.0 / offs["action_scale"]).astype(np.float32)
pol = Policy(w)
dev = torch.device("cuda")
st = torch.zeros((n, env.S), dtype=torch.float32, device=dev); ac = torch.zeros((n, env.A), dtype=torch.float32, device=dev)
rw = torch.zeros(n, dtype=torch.float32, device=dev); tm = torch.zeros(n, dtype=torch.int32, device=dev)
vd = torch.zeros(n, dtype=torch.int32, device=dev); en = torch.zeros(n, dtype=torch.int32, device=dev)
env.step_device(0, st.data_ptr(), rw.data_ptr(), tm.data_ptr(), vd.data_ptr(), en.data_ptr(), n_updates=0)
for k in range(20):
    pol.forward_device(st.data_ptr(), n, ac.data_ptr(), 0, sample=True, seed=1, step=k, stream=stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for k in range(iters):
    pol.forward_device(st.data_ptr(), n, ac.data_ptr(), 0, sample=True, seed=1, step=k, stream=stream)
e1.record(); torch.cuda.synchronize()
pol_ms = e0.elapsed_time(e1) / iters
flops = 2.0 * n * (env.S * 1024 + 1024 * 512 + 512 * env.A)
out = {"envs": n, "policy_ms": pol_ms, "policy_tflops": flops / (pol_ms * 1e-3) / 1e12, "policy_flops_per_call": flops,
       "mfma_peak_tflops_bf16_dense": 2500.0}
out["policy_frac_of_peak"] = out["policy_tflops"] / 2500.0
steps = 100
for k in range(10):
    pol.forward_device(st.data_ptr(), n, ac.data_ptr(), 0, sample=True, seed=1, step=k, stream=stream)
    env.step_device(ac.data_ptr(), st.data_ptr(), rw.data_ptr(), tm.data_ptr(), vd.data_ptr(), en.data_ptr(), auto_reset=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(steps):
    pol.forward_device(st.data_ptr(), n, ac.data_ptr(), 0, sample=True, seed=1, step=k, stream=stream)
    env.step_device(ac.data_ptr(), st.data_ptr(), rw.data_ptr(), tm.data_ptr(), vd.data_ptr(), en.data_ptr(), auto_reset=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
out["closed_loop_env_steps_per_s"] = n * steps / dt; out["closed_loop_ms_per_step"] = 1e3 * dt / steps
out["mean_reward"] = float(rw.mean().item())
# the same loop with every env simulated to the action boundary even after its episode ended (DM_END_EPISODE_EARLY off): what the
# heavy-contact tail of fallen characters costs
torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(steps):
    pol.forward_device(st.data_ptr(), n, ac.data_ptr(), 0, sample=True, seed=1, step=k, stream=stream)
    env.step_device(ac.data_ptr(), st.data_ptr(), rw.data_ptr(), tm.data_ptr(), vd.data_ptr(), en.data_ptr(), auto_reset=True, end_early=False)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
out["closed_loop_no_early_end_env_steps_per_s"] = n * steps / dt
# same closed loop with the manifold cap lowered to 9 contacts per character (dm_create_info.max_contacts): rows <= 4 + 27 < 32, so
# no pair ever leaves the two-per-wave register path
env9 = BatchEnv(t, n, seed=1, max_contacts=9)
env9.set_stream(stream); env9.reset()
env9.step_device(0, st.data_ptr(), rw.data_ptr(), tm.data_ptr(), vd.data_ptr(), en.data_ptr(), n_updates=0)
for ph in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(steps):
        pol.forward_device(st.data_ptr(), n, ac.data_ptr(), 0, sample=True, seed=1, step=k, stream=stream)
        env9.step_device(ac.data_ptr(), st.data_ptr(), rw.data_ptr(), tm.data_ptr(), vd.data_ptr(), en.data_ptr(), auto_reset=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
out["closed_loop_max_contacts_9_env_steps_per_s"] = n * steps / dt
# the closed loop as TWO env groups (deepmimic_amd/groups.py): policy(A) -> step(A) on group A's stream while B is stepping, and vice versa.  One Policy
# object per group (its activation buffers are per object); actions / observations are the groups' rows of the same tensors.
from deepmimic_amd.groups import EnvGroups            # noqa: E402
grp = EnvGroups(t, n, groups=2, seed=1)
grp.reset()
pols = [Policy(w) for _ in range(grp.G)]
gst = [e.own_stream() for e in grp.envs]
ptr = lambda tns, g, width: tns.data_ptr() + 4 * grp.start[g] * width


def grouped(k, **kw):
    for g in range(grp.G):
        pols[g].forward_device(ptr(st, g, env.S), grp.count[g], ptr(ac, g, env.A), 0, sample=True, seed=1, step=k, env_id_offset=grp.first[g], stream=gst[g])
        grp.step_group_device(g, ac.data_ptr(), st.data_ptr(), rw.data_ptr(), tm.data_ptr(), vd.data_ptr(), en.data_ptr(), auto_reset=True, **kw)


torch.cuda.synchronize()
for g in range(grp.G):
    grp.step_group_device(g, 0, st.data_ptr(), rw.data_ptr(), tm.data_ptr(), vd.data_ptr(), en.data_ptr(), n_updates=0)
for k in range(10):
    grouped(k)
grp.synchronize(); torch.cuda.synchronize(); t0 = time.perf_counter()
for k in range(steps):
    grouped(k)
grp.synchronize(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
out["closed_loop_2_groups_env_steps_per_s"] = n * steps / dt
print(json.dumps(out))
