#!/usr/bin/env python3
"""Which way of driving two env groups overlaps their kernels?  (a) ctx-own streams, one host thread alternating the launches; (b) torch streams,
one host thread; (c) ctx-own streams, one host thread per group (C loop); (d) torch streams, one thread per group (Python loop)."""
import os, sys, threading, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from deepmimic_amd import model, streams
from deepmimic_amd.groups import EnvGroups
n, K = 4096, 200
t = model.load_asset("humanoid3d_walk")
torch.zeros(1, device="cuda")
res = {}
def make(G):
    g = EnvGroups(t, n, groups=G, seed=1234, test_mode=True)
    g.reset(kin_times=streams.reset_phase(np.arange(n), g.duration))
    return g
S = torch.zeros((n, 227), device="cuda"); R = torch.zeros(n, device="cuda"); T = torch.zeros(n, dtype=torch.int32, device="cuda"); V = torch.zeros_like(T); E = torch.zeros_like(T)
ptrs = (0, S.data_ptr(), R.data_ptr(), T.data_ptr(), V.data_ptr(), E.data_ptr())
kw = dict(auto_reset=True, open_loop=True)
def loop_alt(g, k):
    for _ in range(k):
        for i in range(g.G): g.step_group_device(i, *ptrs, **kw)
def run(tag, g, fn):
    fn(g, 60); torch.cuda.synchronize(); g.synchronize()
    t0 = time.perf_counter(); fn(g, K); g.synchronize(); torch.cuda.synchronize(); el = time.perf_counter() - t0
    res[tag] = n * K / el; print("%-46s %8.0f env-steps/s" % (tag, res[tag]), flush=True)
g1 = make(1); run("G=1 own stream", g1, loop_alt); g1.close()
g = make(2); run("(a) G=2 own streams, one thread alternating", g, loop_alt)
def thr_py(g, k):
    th = [threading.Thread(target=lambda i=i: [g.step_group_device(i, *ptrs, **kw) for _ in range(k)]) for i in range(g.G)]
    [x.start() for x in th]; [x.join() for x in th]
run("(c') G=2 own streams, python thread per group", g, thr_py)
def thr_c(g, k): g.bench_rollout(0, k)
run("(c) G=2 own streams, C loop thread per group", g, thr_c)
ts = [torch.cuda.Stream() for _ in range(2)]
g.set_streams([x.cuda_stream for x in ts])
run("(b) G=2 torch streams, one thread alternating", g, loop_alt)
run("(d) G=2 torch streams, python thread per group", g, thr_py)
g.close()
print(json.dumps(res))
