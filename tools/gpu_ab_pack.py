"""Same-box comparison of the humanoid step kernels: two characters per wavefront (dense factor) against one per wavefront with the dense
factor (DM_TREE=0) and with the branch-sparse factor on the compiled topology (DM_TREE=1).  usage: python tools/gpu_ab_pack.py [scene] [envs]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepmimic_amd import core, model, streams  # noqa: E402
scene = sys.argv[1] if len(sys.argv) > 1 else "humanoid3d_walk"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
t = model.load_asset(scene)
envs = {}
for tag, pack, tree in (("duo", 2, "1"), ("one_dense", 1, "0"), ("one_tree", 1, "1")):
    os.environ["DM_TREE"] = tree
    env = core.BatchEnv(t, n, seed=1234, test_mode=True, wave_packing=pack)
    env.reset(kin_times=streams.reset_phase(np.arange(n), env.duration))
    env.bench_rollout(60, 1)
    envs[tag] = env
res = {k: [] for k in envs}
for rep in range(5):
    for tag, env in envs.items():
        res[tag].append(env.bench_rollout(0, 100) / 100)
print(json.dumps({"scene": scene, "envs": n, **{tag: {"kernel_ms_median": float(np.median(v)), "env_steps_per_s": n / (float(np.median(v)) * 1e-3)} for tag, v in res.items()}}))
