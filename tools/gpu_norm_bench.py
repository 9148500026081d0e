#!/usr/bin/env python
"""Device normaliser (dm_norm.h): time of record() / update() / normalize() on record blocks of n x 227 fp32, achieved HBM rate
(algorithmic bytes: record reads n x size x 4; normalize reads and writes that much), and what recording every control step costs the
closed loop.  Prints one JSON object; run on the GPU box."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepmimic_amd import model                      # noqa: E402
from deepmimic_amd.core import BatchEnv              # noqa: E402
from deepmimic_amd.normalizer import DeviceNormalizer  # noqa: E402
from deepmimic_amd.policy import Policy, random_weights   # noqa: E402

S = 227
out = {"size": S, "hbm_peak_gbs": 8000.0, "record": {}, "normalize": {}}
nrm = DeviceNormalizer(S, None)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


for n in (4096, 32768, 262144, 1048576):
    x = torch.randn((n, S), device="cuda"); o = torch.empty_like(x)
    ms = timed(lambda: nrm.record_device(x.data_ptr(), n), 50)
    out["record"][str(n)] = {"ms": ms, "gbs": n * S * 4 / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": n * S * 4 / (ms * 1e-3) / 1e9 / 8000.0}
    ms = timed(lambda: nrm.normalize_device(x.data_ptr(), n, o.data_ptr()), 50)
    out["normalize"][str(n)] = {"ms": ms, "gbs": 2 * n * S * 4 / (ms * 1e-3) / 1e9, "frac_of_hbm_peak": 2 * n * S * 4 / (ms * 1e-3) / 1e9 / 8000.0}
    ms = timed(lambda: o.copy_(x), 50)                      # the device's own copy of the same block, as the yardstick of a read + write stream
    out["normalize"][str(n)]["torch_copy_gbs"] = 2 * n * S * 4 / (ms * 1e-3) / 1e9
    ms = timed(lambda: x.sum(), 50)                         # ... and a library reduction as the yardstick of a read-only stream
    out["record"][str(n)]["torch_sum_gbs"] = n * S * 4 / (ms * 1e-3) / 1e9
    del x, o
xs = torch.randn((64, S), device="cuda")
out["record64_plus_update_ms"] = timed(lambda: (nrm.record_device(xs.data_ptr(), 64), nrm.update()), 50)
# closed loop, 4096 humanoids, one stream: policy -> step [-> record]; update every 32 steps (an "iteration")
n = 4096
t = model.load_asset("humanoid3d_walk")
env = BatchEnv(t, n, seed=1)
ts = torch.cuda.Stream(); torch.cuda.set_stream(ts); h = ts.cuda_stream
env.set_stream(h); env.reset()
offs = env.offsets_scales()
w = random_weights(env.S, env.A, seed=0)
w["a_mean"] = -offs["action_offset"].astype(np.float32); w["a_std"] = (1.0 / offs["action_scale"]).astype(np.float32)
pol = Policy(w)
sn = DeviceNormalizer(env.S, offs["state_norm_groups"]); sn.set_stream(h)
sn.set_mean_std(-offs["state_offset"], 1.0 / offs["state_scale"]); sn.bind_policy(pol)
st = torch.zeros((n, env.S), device="cuda"); ac = torch.zeros((n, env.A), device="cuda"); rw = torch.zeros(n, device="cuda")
tm, vd, en = (torch.zeros(n, dtype=torch.int32, device="cuda") for _ in range(3))
env.step_device(0, st.data_ptr(), rw.data_ptr(), tm.data_ptr(), vd.data_ptr(), en.data_ptr(), n_updates=0)


def loop(steps, record):
    for k in range(steps):
        pol.forward_device(st.data_ptr(), n, ac.data_ptr(), 0, sample=True, seed=1, step=k, stream=h)
        env.step_device(ac.data_ptr(), st.data_ptr(), rw.data_ptr(), tm.data_ptr(), vd.data_ptr(), en.data_ptr(), auto_reset=True)
        if record:
            sn.record_device(st.data_ptr(), n)
            if k % 32 == 31:
                sn.update()                  # (not bound to the policy inside the timed loops: both variants then follow the same trajectories)


for record in (False, True, False, True):
    loop(20, record); torch.cuda.synchronize(); t0 = time.perf_counter()
    loop(128, record); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out.setdefault("closed_loop_env_steps_per_s", {}).setdefault("with_record" if record else "plain", []).append(n * 128 / dt)
sn.bind_policy(pol)
loop(8, True); torch.cuda.synchronize()
out["normalizer_after"] = {"count": sn.count, "mean_abs_mean": float(np.abs(sn.mean).mean()), "mean_std": float(sn.std.mean()), "finite": bool(np.isfinite(sn.mean).all() and np.isfinite(sn.std).all())}
print(json.dumps(out))
