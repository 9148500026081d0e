"""Same-box timing of several builds of libdm_hip.so (paths relative to the repo root) on one scene / wave packing.
usage: [DM_AB_AMP=1] python tools/gpu_ab_libs.py scene envs wave_packing lib [lib ...]"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepmimic_amd import core, model, streams  # noqa: E402
scene, n, pack = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
t = model.load_asset(scene)
if os.environ.get("DM_AB_AMP") == "1": t.cfg.scene = "imitate_amp"          # the same asset as --scene imitate_amp (AMP instantiation of the kernel)
envs = {}
for p in sys.argv[4:]:
    path = os.path.join(ROOT, p)
    env = core.BatchEnv(t, n, seed=1234, test_mode=True, wave_packing=pack, lib_path=path, physics=int(os.environ.get("DM_AB_PHYSICS", "1")))
    env.reset(kin_times=streams.reset_phase(np.arange(n), env.duration))
    env.bench_rollout(60, 1)
    envs[p] = env
res = {k: [] for k in envs}
for rep in range(5):
    for tag, env in envs.items():
        res[tag].append(env.bench_rollout(0, 100) / 100)
# outputs of the variants on an identical 40-step rollout (a variant that only moves data must reproduce them bit for bit)
chk = {}
for tag, env in envs.items():
    env.reset(kin_times=streams.reset_phase(np.arange(n), env.duration))
    out = None
    for _ in range(40): out = env.step(None, 1.0 / 600, 20, auto_reset=True, open_loop=True)
    chk[tag] = {"reward_sum": float(np.asarray(out["reward"], dtype=np.float64).sum()), "state_abs_sum": float(np.abs(np.asarray(out["state"], dtype=np.float64)).sum())}
print(json.dumps({"scene": scene, "envs": n, "wave_packing": pack, **{tag: {"kernel_ms_median": float(np.median(v)), "env_steps_per_s": n / (float(np.median(v)) * 1e-3)} for tag, v in res.items()}, "checksums": chk}))
