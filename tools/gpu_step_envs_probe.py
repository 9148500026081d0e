#!/usr/bin/env python
"""What one `dm_step_envs` round trip costs next to the kernel it launches (the broker's owner loop, deepmimic_amd/broker.py): a 64 / 256-env
one-per-wave context, host-array calls, 200 control steps each.  Prints one JSON object; run on the GPU box."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepmimic_amd import model                      # noqa: E402
from deepmimic_amd.core import BatchEnv              # noqa: E402

t = model.load_asset("humanoid3d_walk")
out = {}
for N in (64, 256):
    env = BatchEnv(t, N, seed=1, wave_packing=1)
    env.reset()
    rng = np.random.default_rng(0)
    acts = (0.1 * rng.normal(size=(N, env.A))).astype(np.float32)
    iters = 200
    res = {}

    def timeit(fn):
        for _ in range(10):
            fn()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        return 1e3 * (time.perf_counter() - t0) / iters

    dev = torch.device("cuda")
    st = torch.zeros((N, env.S), dtype=torch.float32, device=dev); ac = torch.from_numpy(acts).to(dev); rw = torch.zeros(N, dtype=torch.float32, device=dev)
    tm, vd, en = (torch.zeros(N, dtype=torch.int32, device=dev) for _ in range(3))

    def dev_step():
        env.step_device(ac.data_ptr(), st.data_ptr(), rw.data_ptr(), tm.data_ptr(), vd.data_ptr(), en.data_ptr(), auto_reset=True)
        env.synchronize()
    res["kernel_plus_sync_ms"] = timeit(dev_step)
    res["step_host_arrays_ms"] = timeit(lambda: env.step(acts, 1.0 / 600, 20, auto_reset=True))
    ids = np.arange(N, dtype=np.int32)
    res["step_envs_all_ms"] = timeit(lambda: env.step_envs(ids, acts, 1.0 / 600, 20, auto_reset=True))
    res["step_envs_all_no_clocks_ms"] = timeit(lambda: env.step_envs(ids, acts, 1.0 / 600, 20, auto_reset=True, want_clocks=False))
    res["step_envs_16_ms"] = timeit(lambda: env.step_envs(ids[:16], acts[:16], 1.0 / 600, 20, auto_reset=True))

    def spaced(gap_ms):                      # the broker's rhythm: the device idles between launches while the workers do their host work
        tot = 0.0
        for k in range(110):
            t1 = time.perf_counter()
            while 1e3 * (time.perf_counter() - t1) < gap_ms:
                pass
            t0 = time.perf_counter()
            env.step_envs(ids[:16], acts[:16], 1.0 / 600, 20, auto_reset=True)
            if k >= 10:
                tot += time.perf_counter() - t0
        return 1e3 * tot / 100
    for gap in (0.5, 1.5, 5.0):
        res["step_envs_16_after_%.1f_ms_idle_ms" % gap] = spaced(gap)
    res["get_state_ms"] = timeit(lambda: env.get_state())
    res["query_ms"] = timeit(lambda: env.query())
    out[str(N)] = res
    env.close()
print(json.dumps(out))
