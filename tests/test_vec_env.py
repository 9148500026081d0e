"""deepmimic_amd.vec_env.TorchVecEnv (device-resident batched env) against the host-pointer path of the same C-ABI."""
import numpy as np
import pytest

from deepmimic_amd import model
from deepmimic_amd.core import BatchEnv


def test_vec_env_needs_a_gpu():
    from deepmimic_amd.vec_env import TorchVecEnv
    with pytest.raises(RuntimeError, match="GPU"):
        TorchVecEnv(model.load_asset("humanoid3d_walk"), 2, device="cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("asset", ["humanoid3d_walk", "amp_target_zombie"])
def test_vec_env_matches_host_path(hip_lib, asset):
    """same seed, same actions: the torch-tensor path (step_device, async on torch's stream) and BatchEnv.step (host buffers) give the
    same observations, rewards and flags through 40 control steps with auto-reset"""
    import torch
    from deepmimic_amd.vec_env import TorchVecEnv
    t = model.load_asset(asset)
    n = 64
    ve = TorchVecEnv(t, n, seed=3, lib_path=hip_lib, amp_obs=True)
    ref = BatchEnv(t, n, seed=3, lib_path=hip_lib)
    obs = ve.reset(); ref.reset()
    q = ref.query()
    assert np.array_equal(obs.cpu().numpy(), q["state"])
    rng = np.random.default_rng(0)
    ends = 0
    for k in range(40):
        a = (0.2 * rng.normal(size=(n, ve.act_dim))).astype(np.float32)
        o, r, d, info = ve.step(torch.from_numpy(a).to(ve.device))
        out = ref.step(a, ve.timestep, ve.updates, auto_reset=True, amp=bool(ve.amp_obs is not None))
        assert np.array_equal(o.cpu().numpy(), out["state"]) and np.array_equal(r.cpu().numpy(), out["reward"])
        assert np.array_equal(d.cpu().numpy(), out["episode_end"].astype(bool)) and np.array_equal(info["terminate"].cpu().numpy(), out["terminate"])
        if ve.goal_dim:
            assert np.array_equal(info["goal"].cpu().numpy(), out["goal"])
        if ve.amp_obs is not None:
            assert np.array_equal(info["amp_obs"].cpu().numpy(), out["amp_obs"])
        ends += int(d.sum().item())
    assert ends > 0
    ve.close(); ref.close()
