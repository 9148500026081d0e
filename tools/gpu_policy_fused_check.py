#!/usr/bin/env python
"""GPU: the one-launch actor (k_policy_fused) against the per-layer kernels (DM_POLICY_LAYERED=1) on the same inputs -- actions must be bit-identical
(same products, same accumulation order, same bf16 rounding points), log-probabilities equal up to the order of one 32-term sum.  Prints one JSON line."""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepmimic_amd.policy import Policy, random_weights
out = []
for S, A, n in ((227, 28, 4096), (347, 58, 1000), (241, 28, 77)):
    w = random_weights(S, A, seed=1)
    rng = np.random.default_rng(2)
    for k, d in (("b1", 1024), ("b2", 512), ("b3", A)):
        w[k] = (0.1 * rng.normal(size=d)).astype(np.float32)
    w["s_mean"] = rng.normal(size=S).astype(np.float32); w["s_std"] = (1 + rng.random(S)).astype(np.float32)
    pol = Policy(w, s_clip=5.0)
    x = torch.tensor(2 * rng.normal(size=(n, S)), dtype=torch.float32, device="cuda")
    res = {}
    for tag, env in (("fused", None), ("layered", "1")):
        if env: os.environ["DM_POLICY_LAYERED"] = env
        else: os.environ.pop("DM_POLICY_LAYERED", None)
        a = torch.zeros((n, A), device="cuda"); lp = torch.zeros(n, device="cuda")
        pol.forward_device(x.data_ptr(), n, a.data_ptr(), lp.data_ptr(), sample=True, seed=3, step=5, env_id_offset=10)
        torch.cuda.synchronize()
        res[tag] = (a.cpu().numpy(), lp.cpu().numpy())
    os.environ.pop("DM_POLICY_LAYERED", None)
    out.append({"S": S, "A": A, "rows": n, "actions_bit_identical": bool(np.array_equal(res["fused"][0], res["layered"][0])),
                "max_abs_logp_diff": float(np.abs(res["fused"][1] - res["layered"][1]).max()), "max_abs_logp": float(np.abs(res["layered"][1]).max())})
print(json.dumps(out))
