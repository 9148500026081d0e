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
SCENE = os.environ.get("SCENE", "humanoid3d_walk")
t = model.load_asset(SCENE)
n = int(os.environ.get("ENVS", "4096"))
env = BatchEnv(t, n, seed=1, wave_packing=int(os.environ.get("PACK", "0")))
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
import time
ends, inval, fbd, tms, clk_adv = [], [], [], [], []
fb0 = env.debug("fallback")
for k in range(100):
    st.copy_(torch.from_numpy(np.ascontiguousarray(state, dtype=np.float32)))
    pol.forward_device(st.data_ptr(), n, ac.data_ptr(), 0, sample=True, seed=1, step=k, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    acts = ac.cpu().numpy()
    c0 = env.get_state()["clocks"][:, 3] if k >= 60 else None
    t0 = time.perf_counter()
    out = env.step(acts, 1 / 600, 20, auto_reset=True)
    tms.append(time.perf_counter() - t0)
    state = out["state"]
    if k >= 60:
        fb1 = env.debug("fallback"); fbd.append((fb1 - fb0).sum() / 2); fb0 = fb1      # pair-substeps on the fallback (both characters of a pair count it)
        ends.append(out["episode_end"].mean()); inval.append(1.0 - out["valid"].mean())
    else:
        fb0 = env.debug("fallback")
print("%s closed loop, steps 60..99: episode ends per env-step %.4f, invalid %.5f, fallback pair-substeps per step %.1f of %d (%.3f %%), host-timed step %.3f ms (min %.3f)"
      % (SCENE, np.mean(ends), np.mean(inval), np.mean(fbd), n // 2 * 40, 100 * np.mean(fbd) / (n // 2 * 40), 1e3 * np.mean(tms[60:]), 1e3 * np.min(tms[60:])))
env.probe(4, 1 / 600)
rows_dbg = None
p = env.debug("prof")
wv = p[0::2] if int(os.environ.get("PACK", "0")) != 1 else p      # (two characters per wave: the wave's counters are in the even env's row)
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

# rows per substep of the policy-driven state distribution: one more substep with the latched torques through the tap build (R, contacts per character)
env.probe(1, 1 / 1200)
rw = env.debug("rows"); R, NC = rw[:, 0].astype(int), rw[:, 1].astype(int)
pairR = np.maximum(R[0::2], R[1::2])
print("rows per character: mean %.1f p50 %d p90 %d p99 %d max %d | characters > 32 rows %.2f %% | pairs with a character > 32 rows %.2f %% | both > 32: %.2f %%"
      % (R.mean(), np.percentile(R, 50), np.percentile(R, 90), np.percentile(R, 99), R.max(), 100 * (R > 32).mean(), 100 * (pairR > 32).mean(), 100 * ((R[0::2] > 32) & (R[1::2] > 32)).mean()))
print("contacts of the characters > 32 rows:", dict(zip(*[x.tolist() for x in np.unique(NC[R > 32], return_counts=True)])))
hv = np.flatnonzero(R > 32)
if hv.size:
    partner = R[hv ^ 1]
    print("heavy + partner rows <= 64: %.1f %% of the heavy characters; partner rows mean %.1f" % (100 * ((R[hv] + partner) <= 64).mean(), partner.mean()))
