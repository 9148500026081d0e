#!/usr/bin/env python3
"""Tail diagnostics of the step kernel in the CLOSED loop (random-init policy, sampled actions): per-wave cycle totals and the
phases that separate the slowest waves from the median ones, plus the rows-per-substep picture.  Run on the GPU box."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from deepmimic_amd import model
from deepmimic_amd.core import BatchEnv
from deepmimic_amd.policy import Policy, random_weights
PH = ["kin_update+latch", "spd.kinematics", "spd.dynamics", "spd.chol+solve", "spd.err/clamp|sub.pre", "sub.kinematics", "sub.dynamics",
      "sub.chol+solve+vstar", "sub.collision", "sub.rows(J,Y)", "sub.A", "sub.PGS", "sub.backsolve+integrate", "emit(+reset)", "store", "load+action"]
t = model.load_asset("humanoid3d_walk")
n = int(os.environ.get("ENVS", "4096"))
env = BatchEnv(t, n, seed=1)
env.reset()
offs = env.offsets_scales()
w = random_weights(env.S, env.A, seed=0)
w["s_mean"] = -offs["state_offset"].astype(np.float32); w["s_std"] = (1.0 / offs["state_scale"]).astype(np.float32)
w["a_mean"] = -offs["action_offset"].astype(np.float32); w["a_std"] = (1.0 / offs["action_scale"]).astype(np.float32)
pol = Policy(w)
dev = torch.device("cuda")
st = torch.zeros((n, env.S), dtype=torch.float32, device=dev); ac = torch.zeros((n, env.A), dtype=torch.float32, device=dev)
out = env.step(None, 1 / 600, 0)
state = out["state"]
for k in range(60):
    st.copy_(torch.from_numpy(np.ascontiguousarray(state, dtype=np.float32)))
    pol.forward_device(st.data_ptr(), n, ac.data_ptr(), 0, sample=True, seed=1, step=k, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    acts = ac.cpu().numpy()
    out = env.step(acts, 1 / 600, 20, auto_reset=True)
    state = out["state"]
env.probe(4, 1 / 600)
p = env.debug("prof")
wv = p[0::2]
tot = wv.sum(1)
order = np.argsort(tot)
med = order[len(order) // 2 - 50: len(order) // 2 + 50]; top = order[-20:]
print("closed loop, envs", n, "waves", len(tot), "mean %.3fM median %.3fM p90 %.3fM p99 %.3fM max %.3fM" % tuple(x / 1e6 for x in (tot.mean(), np.median(tot), np.percentile(tot, 90), np.percentile(tot, 99), tot.max())))
print("phase: median-waves mean | slowest-20 mean | delta (k cycles)")
for i in range(16):
    a, b = wv[med, i].mean(), wv[top, i].mean()
    print("  %-26s %9.0f %9.0f %+9.0f" % (PH[i], a / 1e3, b / 1e3, (b - a) / 1e3))
hist, edges = np.histogram(tot / np.median(tot), bins=[0, 0.9, 1.0, 1.1, 1.2, 1.3, 1.5, 1.75, 2.0, 3.0])
print("wave time / median:", dict(zip(["<%.2f" % e for e in edges[1:]], [int(h) for h in hist])))
