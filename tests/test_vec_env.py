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


@pytest.mark.gpu
def test_set_stream_of_torchs_default_stream_orders_the_launch(hip_lib):
    """torch's default stream has the null handle; BatchEnv.set_stream maps it to dm_set_stream_default, so a torch op enqueued right after
    step_device on the default stream sees the step's outputs (a 2 ms kernel: an unordered read would see the buffer's previous content)"""
    import torch
    t = model.load_asset("humanoid3d_walk")
    n = 4096
    env = BatchEnv(t, n, seed=5, lib_path=hip_lib)
    torch.cuda.set_stream(torch.cuda.default_stream())           # whatever an earlier test left current
    assert torch.cuda.current_stream().cuda_stream == 0
    env.set_stream(torch.cuda.current_stream().cuda_stream)
    env.reset()
    dev = torch.device("cuda")
    st = torch.full((n, env.S), -7.0, dtype=torch.float32, device=dev); rw = torch.full((n,), -7.0, dtype=torch.float32, device=dev)
    tm = torch.zeros(n, dtype=torch.int32, device=dev); vd = torch.zeros(n, dtype=torch.int32, device=dev); en = torch.zeros(n, dtype=torch.int32, device=dev)
    for k in range(5):
        st.fill_(-7.0); rw.fill_(-7.0)
        env.step_device(0, st.data_ptr(), rw.data_ptr(), tm.data_ptr(), vd.data_ptr(), en.data_ptr(), open_loop=True, auto_reset=True)
        seen_s, seen_r = st.clone(), rw.clone()                  # enqueued on the default stream right behind the launch, no synchronize in between
        torch.cuda.synchronize()
        assert torch.equal(seen_s, st) and torch.equal(seen_r, rw)
        assert float(seen_r.min()) >= 0.0 and float(seen_s.abs().max()) < 1e3        # the step's outputs, not the -7 fill
    env.close()


@pytest.mark.gpu
def test_vec_env_follows_the_current_stream(hip_lib):
    """the launches ride on whatever torch stream is current at the call (no stream of their own, no events): alternating between the default
    stream and a side stream, the results stay those of the host-pointer path"""
    import torch
    from deepmimic_amd.vec_env import TorchVecEnv
    t = model.load_asset("humanoid3d_walk")
    n = 64
    torch.cuda.set_stream(torch.cuda.default_stream())
    ve = TorchVecEnv(t, n, seed=8, lib_path=hip_lib)
    ref = BatchEnv(t, n, seed=8, lib_path=hip_lib)
    ve.reset(); ref.reset()
    side = torch.cuda.Stream()
    rng = np.random.default_rng(2)
    for k in range(8):
        a = (0.2 * rng.normal(size=(n, ve.act_dim))).astype(np.float32)
        with torch.cuda.stream(side if k % 2 else torch.cuda.default_stream()):
            at = torch.from_numpy(a).to(ve.device)
            o, r, d, _ = ve.step(at)
            o2 = (o * 1.0).cpu().numpy()                 # a torch op behind the launch on the same stream, then the copy: sees the step's output
            r2 = r.cpu().numpy()
        out = ref.step(a, ve.timestep, ve.updates, auto_reset=True)
        assert np.array_equal(o2, out["state"]) and np.array_equal(r2, out["reward"]), k
    ve.close(); ref.close()


@pytest.mark.gpu
def test_vec_env_groups_match_single_launch(hip_lib):
    """TorchVecEnvGroups (two env groups on their own streams, round 4): env by env the same observations, rewards and done flags as TorchVecEnv over
    the whole batch, both through the synchronous step() and through per-group step_group() calls issued out of order"""
    import torch
    from deepmimic_amd.vec_env import TorchVecEnv, TorchVecEnvGroups
    t = model.load_asset("humanoid3d_walk")
    n = 128
    one = TorchVecEnv(t, n, seed=4, lib_path=hip_lib)
    grp = TorchVecEnvGroups(t, n, groups=2, seed=4, lib_path=hip_lib)
    assert grp.G == 2 and grp.rows(1) == slice(64, 128)
    assert torch.equal(one.reset(), grp.reset())
    rng = np.random.default_rng(1)
    ends = 0
    for k in range(30):
        a = torch.from_numpy((0.2 * rng.normal(size=(n, one.act_dim))).astype(np.float32)).to(one.device)
        o1, r1, d1, _ = one.step(a)
        if k % 2 == 0:
            o2, r2, d2, _ = grp.step(a)
        else:                                        # per-group calls, group 1 first; outputs consumed after joining both streams
            torch.cuda.current_stream().synchronize()
            parts = {}
            for g in (1, 0):
                parts[g] = grp.step_group(g, a)
            for g in (0, 1):
                grp.stream(g).synchronize()
            o2, r2 = grp.obs, grp.reward
            d2 = torch.cat([parts[0][2], parts[1][2]])
        torch.cuda.synchronize()
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2), k
        ends += int(d1.sum().item())
    assert ends > 0
    one.close(); grp.close()


@pytest.mark.gpu
def test_vec_env_groups_amp_and_goal_rows(hip_lib):
    """the AMP observation and RecordGoal arrays of TorchVecEnvGroups are whole-batch like the others: env by env what TorchVecEnv reports"""
    import torch
    from deepmimic_amd.vec_env import TorchVecEnv, TorchVecEnvGroups
    t = model.load_asset("amp_heading_zombie")
    n = 64
    one = TorchVecEnv(t, n, seed=9, amp_obs=True, lib_path=hip_lib)
    grp = TorchVecEnvGroups(t, n, groups=2, seed=9, amp_obs=True, lib_path=hip_lib)
    assert grp.G == 2 and grp.goal_dim == one.goal_dim > 0 and grp.amp_obs is not None
    assert torch.equal(one.reset(), grp.reset())
    rng = np.random.default_rng(2)
    for k in range(6):
        a = torch.from_numpy((0.2 * rng.normal(size=(n, one.act_dim))).astype(np.float32)).to(one.device)
        o1, r1, d1, i1 = one.step(a)
        o2, r2, d2, i2 = grp.step(a)
        torch.cuda.synchronize()
        assert torch.equal(o1, o2) and torch.equal(r1, r2) and torch.equal(d1, d2), k
        assert torch.equal(i1["amp_obs"], i2["amp_obs"]) and torch.equal(i1["goal"], i2["goal"]), k
    assert float(i2["amp_obs"][n // 2:].abs().max()) > 0 and float(i2["goal"][n // 2:].abs().max()) > 0
    one.close(); grp.close()
