"""Same-box A/B of two builds of libdm_hip.so on BOTH workloads of the bench line: the fixed-action (open-loop) rollout and the closed loop
(on-device policy -> control step, one stream).  usage: python tools/gpu_ab_closed.py libA.so libB.so [envs]   (paths relative to the repo root)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from deepmimic_amd import core, model, streams          # noqa: E402
from deepmimic_amd.policy import Policy, random_weights  # noqa: E402
from gpu_ab_bench import load_raw                        # noqa: E402


def main():
    a, b = [os.path.join(ROOT, p) for p in sys.argv[1:3]]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
    t = model.load_asset(os.environ.get("SCENE", "humanoid3d_walk"))
    dev = torch.device("cuda")
    tstream = torch.cuda.Stream(); torch.cuda.set_stream(tstream)
    ctx = {}
    for tag, path in (("A", a), ("B", b)):
        load_raw(path)
        env = core.BatchEnv(t, n, seed=1234, lib_path=path, test_mode=True)
        env.reset(kin_times=streams.reset_phase(np.arange(n), env.duration))
        env.bench_rollout(60, 1)
        cl = core.BatchEnv(t, n, seed=1234, lib_path=path, test_mode=True)
        cl.set_stream(tstream.cuda_stream); cl.reset(kin_times=streams.reset_phase(np.arange(n), cl.duration))
        offs = cl.offsets_scales()
        w = random_weights(cl.S, cl.A, seed=0)
        w["s_mean"] = -offs["state_offset"].astype(np.float32); w["s_std"] = (1.0 / offs["state_scale"]).astype(np.float32)
        w["a_mean"] = -offs["action_offset"].astype(np.float32); w["a_std"] = (1.0 / offs["action_scale"]).astype(np.float32)
        pol = Policy(w, lib_path=path)
        bufs = dict(st=torch.zeros((n, cl.S), device=dev), ac=torch.zeros((n, cl.A), device=dev), rw=torch.zeros(n, device=dev),
                    tm=torch.zeros(n, dtype=torch.int32, device=dev), vd=torch.zeros(n, dtype=torch.int32, device=dev), en=torch.zeros(n, dtype=torch.int32, device=dev))
        cl.step_device(0, bufs["st"].data_ptr(), bufs["rw"].data_ptr(), bufs["tm"].data_ptr(), bufs["vd"].data_ptr(), bufs["en"].data_ptr(), n_updates=0)
        ctx[tag] = (env, cl, pol, bufs, [0])

    def closed(tag, steps):
        env, cl, pol, B, k = ctx[tag]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps):
            pol.forward_device(B["st"].data_ptr(), n, B["ac"].data_ptr(), 0, sample=True, seed=1, step=k[0], stream=tstream.cuda_stream); k[0] += 1
            cl.step_device(B["ac"].data_ptr(), B["st"].data_ptr(), B["rw"].data_ptr(), B["tm"].data_ptr(), B["vd"].data_ptr(), B["en"].data_ptr(), auto_reset=True)
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / steps

    for tag in ("A", "B"):
        closed(tag, 60)                                   # into the policy-driven episode mixture
    res = {"A": {"open": [], "closed": []}, "B": {"open": [], "closed": []}}
    for rep in range(5):
        for tag in ("A", "B"):
            res[tag]["open"].append(ctx[tag][0].bench_rollout(0, 100) / 100)
            res[tag]["closed"].append(closed(tag, 100))
    out = {}
    for tag, p in (("A", a), ("B", b)):
        o, c = float(np.median(res[tag]["open"])), float(np.median(res[tag]["closed"]))
        out[tag] = {"lib": os.path.relpath(p, ROOT), "open_ms": o, "closed_ms": c, "open_env_steps_per_s": n / (o * 1e-3), "closed_env_steps_per_s": n / (c * 1e-3),
                    "closed_over_open_rate": o / c, "mean_reward_closed": float(ctx[tag][3]["rw"].mean().item())}
    out["B_over_A_open_time"] = out["B"]["open_ms"] / out["A"]["open_ms"]; out["B_over_A_closed_time"] = out["B"]["closed_ms"] / out["A"]["closed_ms"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
