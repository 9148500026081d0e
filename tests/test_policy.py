"""On-device policy inference (deepmimic_amd/csrc/dm_policy.h) against a plain numpy / torch fp32
statement of the same actor.  CPU: emulator build (lane-exchange emulation of the MFMA); GPU: the real matrix cores."""
import numpy as np
import pytest

from deepmimic_amd import streams
from deepmimic_amd.policy import Policy, random_weights, reference_forward


def make(S, A, H1, H2, seed, with_norm=True):
    w = random_weights(S, A, H1, H2, seed=seed, init_output_scale=0.3)
    rng = np.random.default_rng(seed + 1)
    w["b1"] = rng.normal(size=H1).astype(np.float32) * 0.1; w["b2"] = rng.normal(size=H2).astype(np.float32) * 0.1
    w["b3"] = rng.normal(size=A).astype(np.float32) * 0.1
    if with_norm:
        w["s_mean"] = rng.normal(size=S).astype(np.float32); w["s_std"] = rng.uniform(0.5, 2.0, size=S).astype(np.float32)
        w["a_mean"] = rng.normal(size=A).astype(np.float32); w["a_std"] = rng.uniform(0.5, 2.0, size=A).astype(np.float32)
    return w


def test_policy_emulator_matches_bf16_reference(emu_lib):
    """small widths (the emulated MFMA is slow): asymmetric random weights, ragged row count (M % 32 != 0), S % 32 != 0, A < 32"""
    S, A, H1, H2 = 45, 7, 64, 128
    w = make(S, A, H1, H2, 3)
    pol = Policy(w, lib_path=emu_lib, s_clip=5.0)
    s = np.random.default_rng(0).normal(size=(37, S)).astype(np.float32) * 2 + 0.5
    a, lp = pol.forward_host(s)
    want, _ = reference_forward(w, s, s_clip=5.0, bf16=True)
    assert np.abs(a - want).max() < 2e-3, np.abs(a - want).max()
    want32, _ = reference_forward(w, s, s_clip=5.0, bf16=False)
    assert np.abs(a - want32).max() < 0.1
    # mode: logp is the constant -sum(logstd) - A/2 log(2 pi)
    assert np.allclose(lp, -w["logstd"].sum() - 0.5 * A * np.log(2 * np.pi), atol=1e-5)


def test_policy_emulator_tiled_gemm_and_one_wave_kernels_agree(emu_lib, monkeypatch):
    """widths that are multiples of 128 take the LDS-tiled four-wave GEMM (k_policy_gemm): both layers tiled, a row count that leaves
    ragged 128-row blocks; the one-wave kernels (DM_POLICY_ONE_WAVE) give the same bf16 activations, hence the same actions"""
    S, A, H1, H2 = 70, 9, 256, 128
    w = make(S, A, H1, H2, 8)
    s = np.random.default_rng(4).normal(size=(200, S)).astype(np.float32) * 1.5
    pol = Policy(w, lib_path=emu_lib, s_clip=5.0)
    want, _ = reference_forward(w, s, s_clip=5.0, bf16=True)
    for tile in ("128", "64"):                   # both tile heights (the launch picks by batch size; forced here)
        monkeypatch.setenv("DM_POLICY_TILE", tile)
        a_t, _ = pol.forward_host(s)
        assert np.abs(a_t - want).max() < 2e-3, (tile, np.abs(a_t - want).max())
    monkeypatch.setenv("DM_POLICY_ONE_WAVE", "1")
    a_o, _ = pol.forward_host(s)
    assert np.abs(a_t - a_o).max() < 1e-5        # same products, another summation order inside an fp32 accumulator


def test_policy_emulator_sampling_uses_the_philox_stream(emu_lib):
    S, A, H1, H2 = 32, 5, 64, 64
    w = make(S, A, H1, H2, 5, with_norm=False)
    pol = Policy(w, lib_path=emu_lib)
    s = np.random.default_rng(1).normal(size=(16, S)).astype(np.float32)
    a0, _ = pol.forward_host(s)
    a1, lp = pol.forward_host(s, sample=True, seed=0xD33B, step=3, env_id_offset=100)
    z = (a1 - a0) / np.exp(w["logstd"])
    # same generator as deepmimic_amd/streams.py (key 0xD33B + env id, counter step * A + j), fp32 Box-Muller on 24-bit uniforms
    want = streams.normal_noise(100 + np.arange(16), 3, A, sigma=1.0)
    assert np.abs(z - want).max() < 2e-3, np.abs(z - want).max()
    assert np.allclose(lp, (-0.5 * z ** 2 - w["logstd"]).sum(1) - 0.5 * A * np.log(2 * np.pi), atol=2e-3)


def test_policy_emulator_logp_more_than_32_actions(emu_lib):
    """A = 36 (humanoid with spherical hips as 4-vectors) / 58 (dog): the Gaussian head spans two 32-column blocks; logp must be the
    sum over ALL action components (a per-block partial sum used to win a last-writer race)."""
    for A in (36, 58):
        S, H1, H2 = 40, 64, 64
        w = make(S, A, H1, H2, 9, with_norm=False)
        w["logstd"] = np.random.default_rng(3).normal(size=A).astype(np.float32) * 0.3 - 1.0
        pol = Policy(w, lib_path=emu_lib)
        s = np.random.default_rng(4).normal(size=(19, S)).astype(np.float32)
        a0, lp0 = pol.forward_host(s)
        assert np.allclose(lp0, -w["logstd"].sum() - 0.5 * A * np.log(2 * np.pi), atol=1e-4)
        a1, lp = pol.forward_host(s, sample=True, seed=11, step=2, env_id_offset=7)
        z = (a1 - a0) / np.exp(w["logstd"])
        assert np.abs(z).max() > 0.5                                     # noise reached every block
        assert np.abs(z[:, 32:]).max() > 0.5
        want = (-0.5 * z ** 2 - w["logstd"]).sum(1) - 0.5 * A * np.log(2 * np.pi)
        assert np.abs(lp - want).max() < 5e-3, (A, np.abs(lp - want).max())


def _coin(seed, env_ids, step):
    """the exploration coin of dm_policy_forward_ex: Philox4x32-10, key (seed_lo + env, seed_hi), counter (step, 1, 0, 0), 24-bit uniform"""
    env_ids = np.asarray(env_ids, dtype=np.int64)
    ctr = np.zeros((env_ids.size, 4), np.uint32); ctr[:, 0] = step; ctr[:, 1] = 1
    key = np.zeros((env_ids.size, 2), np.uint32); key[:, 0] = ((seed & 0xFFFFFFFF) + env_ids) & 0xFFFFFFFF; key[:, 1] = (seed >> 32) & 0xFFFFFFFF
    r = streams.philox4x32_10(ctr, key)
    return ((r[:, 0] >> 8).astype(np.float64) + 0.5) / 16777216.0


def _check_forward_ex(lib, on_gpu):
    """pg_agent.py:214-221 for a batch: goal block concatenated behind the state block, exploration coin per row, EXP flags"""
    S, G, A, H1, H2, n = 45, 6, 7, 64, 128, 300
    w = make(S + G, A, H1, H2, 12)
    pol = Policy(w, lib_path=lib, s_clip=5.0)
    rng = np.random.default_rng(2)
    s = rng.normal(size=(n, S)).astype(np.float32) * 2 + 0.5; g = rng.normal(size=(n, G)).astype(np.float32)
    cat = np.concatenate([s, g], axis=1)

    def run(goals, rate, sample, **kw):
        if not on_gpu:
            return pol.forward_host_ex(s if goals is not None else cat, goals, rate, sample, **kw)
        import torch
        sd = torch.from_numpy(s if goals is not None else cat).cuda(); gd = None if goals is None else torch.from_numpy(goals).cuda()
        a = torch.zeros((n, A), device="cuda"); lp = torch.zeros(n, device="cuda"); fl = torch.zeros(n, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        pol.forward_device_ex(sd.data_ptr(), n, a.data_ptr(), 0 if gd is None else gd.data_ptr(), 0 if gd is None else G, lp.data_ptr(), fl.data_ptr(), rate, sample, **kw)
        torch.cuda.synchronize()
        return a.cpu().numpy(), lp.cpu().numpy(), fl.cpu().numpy()
    # goal block == concatenated input, bit for bit; both equal the bf16 statement
    a_cat, lp_cat, _ = run(None, 1.0, False)
    a_g, lp_g, fl = run(g, 1.0, False)
    assert np.array_equal(a_cat, a_g) and np.array_equal(lp_cat, lp_g) and not fl.any()
    want, _ = reference_forward(w, cat, s_clip=5.0, bf16=True)
    assert np.abs(a_g - want).max() < 2e-3
    # exploration: rows whose coin < rate take the sampled action (same noise as sample = 1 everywhere), the others the mode
    kw = dict(seed=0xABCDEF0123, step=17, env_id_offset=40)
    a_all, lp_all, fl_all = run(g, 1.0, True, **kw)
    assert fl_all.all()
    a_mix, lp_mix, fl_mix = run(g, 0.3, True, **kw)
    coin = _coin(0xABCDEF0123, 40 + np.arange(n), 17)
    assert np.array_equal(fl_mix != 0, coin < 0.3) and 0.15 < fl_mix.mean() < 0.45
    ex = fl_mix != 0
    assert np.array_equal(a_mix[ex], a_all[ex]) and np.array_equal(lp_mix[ex], lp_all[ex])
    assert np.array_equal(a_mix[~ex], a_g[~ex]) and np.array_equal(lp_mix[~ex], lp_g[~ex])
    a0, _, fl0 = run(g, 0.0, True, **kw)
    assert np.array_equal(a0, a_g) and not fl0.any()
    with pytest.raises(RuntimeError, match="exp_rate"):
        run(g, 1.5, True)
    pol.close()


def test_policy_forward_ex_goal_block_and_exploration_emulator(emu_lib):
    _check_forward_ex(emu_lib, False)


@pytest.mark.gpu
def test_policy_forward_ex_goal_block_and_exploration_gpu(hip_lib):
    _check_forward_ex(hip_lib, True)


def test_policy_as_critic_one_output(emu_lib):
    """the critic of learning/pg_agent.py:179-188 is the same net with ONE linear output, un-normalised by val_norm (a_mean / a_std here): A = 1 works"""
    S, H1, H2 = 45, 64, 64
    w = make(S, 1, H1, H2, 21)
    crit = Policy(w, lib_path=emu_lib, s_clip=5.0)
    s = np.random.default_rng(6).normal(size=(21, S)).astype(np.float32)
    v, _ = crit.forward_host(s)
    want, _ = reference_forward(w, s, s_clip=5.0, bf16=True)
    assert v.shape == (21, 1) and np.abs(v - want).max() < 2e-3
    crit.close()


def test_policy_rejects_bad_shapes(emu_lib):
    w = random_weights(10, 4, 96, 64)
    with pytest.raises(RuntimeError, match="multiples of 64"):
        Policy(w, lib_path=emu_lib)


@pytest.mark.gpu
@pytest.mark.parametrize("S,A,n", [(227, 28, 4096), (347, 58, 1000), (197, 36, 33)])
def test_policy_gpu_matches_reference(hip_lib, S, A, n):
    """reference architecture 1024-512 at the humanoid / dog sizes, through the C-ABI on device buffers"""
    import torch
    w = make(S, A, 1024, 512, 11)
    pol = Policy(w, lib_path=hip_lib, s_clip=10.0)
    s = (np.random.default_rng(2).normal(size=(n, S)) * 1.5 + 0.3).astype(np.float32)
    ts = torch.from_numpy(s).cuda(); ta = torch.zeros((n, A), dtype=torch.float32, device="cuda"); tl = torch.zeros(n, dtype=torch.float32, device="cuda")
    pol.forward_device(ts.data_ptr(), n, ta.data_ptr(), tl.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    a = ta.cpu().numpy()
    want_bf, _ = reference_forward(w, s, s_clip=10.0, bf16=True)
    want_32, _ = reference_forward(w, s, s_clip=10.0, bf16=False)
    scale = np.abs(want_32).max()
    assert np.abs(a - want_bf).max() < 2e-3 * scale, (np.abs(a - want_bf).max(), scale)      # same rounding points: accumulation order only
    assert np.abs(a - want_32).max() < 2e-2 * scale, (np.abs(a - want_32).max(), scale)      # bf16 operands vs fp32: stated tolerance 2 % of range
    # torch fp32 reference of the same op
    tw = {k: torch.from_numpy(v).cuda() for k, v in w.items()}
    x = torch.clamp((ts - tw["s_mean"]) / tw["s_std"], -10, 10)
    h = torch.relu(x @ tw["w1"] + tw["b1"]); h = torch.relu(h @ tw["w2"] + tw["b2"])
    ref = (h @ tw["w3"] + tw["b3"]) * tw["a_std"] + tw["a_mean"]
    assert (ta - ref).abs().max().item() < 2e-2 * scale
    # mode: logp is the constant -sum(logstd) - A/2 log(2 pi) for every row
    assert np.allclose(tl.cpu().numpy(), -w["logstd"].sum() - 0.5 * A * np.log(2 * np.pi), atol=1e-4)
    # sampled actions: noise stream reproducible on the host
    pol.forward_device(ts.data_ptr(), n, ta.data_ptr(), tl.data_ptr(), sample=True, seed=7, step=9, env_id_offset=5,
                       stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    z = (ta.cpu().numpy() - a) / (np.exp(w["logstd"]) * w["a_std"])
    ids = 5 + np.arange(n)
    ctr_key = streams.normal_noise(ids - 0xD33B + 7, 9, A, sigma=1.0)      # streams keys with 0xD33B + id; the kernel with seed + id
    assert np.abs(z - ctr_key).max() < 5e-3
    # sampled logp over ALL A components (A = 36 and 58 span two column blocks of the head)
    want_lp = (-0.5 * z.astype(np.float64) ** 2 - w["logstd"]).sum(1) - 0.5 * A * np.log(2 * np.pi)
    assert np.abs(tl.cpu().numpy() - want_lp).max() < 2e-2 * max(1.0, np.abs(want_lp).max() / 10), np.abs(tl.cpu().numpy() - want_lp).max()


@pytest.mark.gpu
def test_closed_loop_rollout_stays_on_device(hip_lib):
    """observation -> policy -> control step, 10 steps, 4096 envs, no host round trip of states or actions"""
    import torch
    from deepmimic_amd import model
    from deepmimic_amd.core import BatchEnv
    t = model.load_asset("humanoid3d_walk")
    n = 4096
    env = BatchEnv(t, n, lib_path=hip_lib)
    torch.cuda.set_stream(torch.cuda.Stream())        # an explicit stream: the null handle of torch's default stream would mean "the ctx's own stream"
    env.set_stream(torch.cuda.current_stream().cuda_stream)
    env.reset()
    offs = env.offsets_scales()
    w = random_weights(env.S, env.A, seed=0)
    w["s_mean"] = -offs["state_offset"].astype(np.float32); w["s_std"] = (1.0 / offs["state_scale"]).astype(np.float32)
    w["a_mean"] = -offs["action_offset"].astype(np.float32); w["a_std"] = (1.0 / offs["action_scale"]).astype(np.float32)
    pol = Policy(w, lib_path=hip_lib)
    dev = torch.device("cuda")
    st = torch.zeros((n, env.S), dtype=torch.float32, device=dev); ac = torch.zeros((n, env.A), dtype=torch.float32, device=dev)
    rw = torch.zeros(n, dtype=torch.float32, device=dev); tm = torch.zeros(n, dtype=torch.int32, device=dev)
    vd = torch.zeros(n, dtype=torch.int32, device=dev); en = torch.zeros(n, dtype=torch.int32, device=dev)
    env.step_device(0, st.data_ptr(), rw.data_ptr(), tm.data_ptr(), vd.data_ptr(), en.data_ptr(), n_updates=0)     # first observation
    stream = torch.cuda.current_stream().cuda_stream
    tot = 0.0
    for k in range(10):
        pol.forward_device(st.data_ptr(), n, ac.data_ptr(), 0, sample=True, seed=1, step=k, stream=stream)
        env.step_device(ac.data_ptr(), st.data_ptr(), rw.data_ptr(), tm.data_ptr(), vd.data_ptr(), en.data_ptr(), auto_reset=True)
        tot += float(rw.mean().item())
    assert torch.isfinite(st).all() and torch.isfinite(ac).all() and 0.0 < tot / 10 < 1.0
    torch.cuda.synchronize(); torch.cuda.set_stream(torch.cuda.default_stream())      # leave the process as found
