"""Per-env parity ON THE CONFIGURATIONS bench.py TIMES (BASELINE.json configs 2 / 3 / 4, and the last shard of config 5): 4096 envs in one
launch -- every wave slot of the chip occupied once, 8 x 20 KB of LDS per CU, block indices up to 2047 -- as one context and as the bench's two
env groups on two streams (the concurrent `dm_bench_rollout` route).

(a) rows [o, o + 64) of the 4096-env batch are BIT-IDENTICAL to a 64-env context created with env_id_offset = o, through auto-resets, for the
    first, a middle (straddling the two groups' boundary) and the last waves;
(b) sampled envs of the 4096 batch against the oracle, one control step at a time from the device's own states (parity_common.sampled_compare),
    with the fixed fp32 bounds of test_parity_gpu.test_stepwise_300_steps_live.
Reference rows: scenes/SceneImitate.cpp:7-127 (reward), sim/CtController.cpp:281-478 (state vector), via the oracle."""
import numpy as np
import pytest

import parity_common as pc
from deepmimic_amd import model, streams
from deepmimic_amd.core import BatchEnv
from deepmimic_amd.groups import EnvGroups

pytestmark = pytest.mark.gpu

N = 4096
KEYS = ("state", "reward", "terminate", "valid", "episode_end")
SCENES = ["humanoid3d_walk", "humanoid3d_spinkick", "dog3d_pace"]


def _make(kind, t, n, off, lib, precision=32):
    kw = dict(seed=1234, precision=precision, lib_path=lib, test_mode=True, env_id_offset=off)        # bench.py's own construction
    env = BatchEnv(t, n, **kw) if kind == "one" else EnvGroups(t, n, groups=2, **kw)
    env.reset(kin_times=streams.reset_phase(off + np.arange(n), env.duration))
    return env


def _step(env):
    return env.step(None, pc.DT, 20, open_loop=True, auto_reset=True)


@pytest.mark.parametrize("shard", [0, 7])
@pytest.mark.parametrize("kind", ["one", "groups2"])
@pytest.mark.parametrize("scene", SCENES)
def test_rows_of_4096_bit_identical_to_64_env_contexts(hip_lib, scene, kind, shard):
    if shard and (scene != "humanoid3d_walk"):
        pytest.skip("config 5 (the 8-GPU shards) is humanoid3d_walk")
    t = model.load_asset(scene)
    base = shard * N                                  # global env id of the shard's first env (rank 7 of config 5: 28 672)
    big = _make(kind, t, N, base, hip_lib)
    if kind == "groups2":
        assert big.G == 2
    offs = (0, 1022 * 2, 4032)                        # first waves | the last waves of group 0 and the first of group 1 | last waves
    small = [_make("one", t, 64, base + o, hip_lib) for o in offs]
    ends = 0

    def lockstep(k):
        nonlocal ends
        for _ in range(k):
            a = _step(big)
            for o, s in zip(offs, small):
                b = _step(s)
                for key in KEYS:
                    assert np.array_equal(a[key][o:o + 64], b[key]), (scene, kind, o, key)
                ends += int(b["episode_end"].sum())

    lockstep(12)
    # the route the bench times: the C rollout loop, no host arrays; with two groups, both loops at once from two host threads
    big.bench_rollout(0, 60)
    for s in small:
        s.bench_rollout(0, 60)
    sa = big.get_state()
    for o, s in zip(offs, small):
        sb = s.get_state()
        for key in sb:
            assert np.array_equal(sa[key][o:o + 64], sb[key]), (scene, kind, o, key)
    lockstep(5)
    assert ends > 0, "the compared rows must have crossed episode ends"
    big.close()
    for s in small:
        s.close()


# sampled global rows: both lanes halves, low / middle / high block indices, the very last env; 16 envs
SAMPLE = [0, 1, 130, 517, 771, 1024, 1285, 1800, 2047, 2048, 2303, 2822, 3079, 3590, 4094, 4095]


@pytest.mark.parametrize("kind", ["one", "groups2"])
@pytest.mark.parametrize("scene,steps", [("humanoid3d_walk", 150), ("humanoid3d_spinkick", 100), ("dog3d_pace", 60)])
def test_sampled_envs_of_4096_vs_oracle(hip_lib, scene, steps, kind):
    t = model.load_asset(scene)
    env = _make(kind, t, N, 0, hip_lib)
    for _ in range(30):                                # into the steady-state episode mixture the bench times
        _step(env)
    dr, ds, alive, ok, ends, d32 = pc.sampled_compare(env.get_state, lambda: _step(env), t, SAMPLE, steps, conditioning=True)
    live, sl = dr[alive], ds[alive & np.isfinite(ds)]
    print("%s %s: live %d/%d, ends %d, reward MAE %.2e p99 %.2e max %.2e n>1e-4 %d; state mean %.2e p99 %.2e max %.2e"
          % (scene, kind, alive.sum(), dr.size, ends, live.mean(), np.quantile(live, 0.99), live.max(), (live > 1e-4).sum(), sl.mean(), np.quantile(sl, 0.99), sl.max()))
    big = np.argwhere(alive & (dr > 1e-4))
    print("  steps beyond 1e-4 (device vs fp64 oracle | the oracle's own fp32 build vs fp64 oracle): " + ", ".join("%.1e|%.1e" % (dr[k, j], d32[k, j]) for k, j in big))
    assert ok, "terminate / valid / episode_end differ from the oracle"
    assert alive.mean() > 0.5 and ends > 0
    assert dr[~alive].max(initial=0.0) < 1e-6          # both sides report 0 for a fallen character
    # the fixed fp32 bounds of test_stepwise_300_steps_live: MAE, 99th percentile, share of steps beyond 1e-4 ...
    assert live.mean() < 1e-5 and np.quantile(live, 0.99) < 1e-4 and (live > 1e-4).mean() < 0.01
    assert sl.mean() < 5e-3 and np.quantile(sl, 0.99) < 5e-2
    # ... and the maximum: < 5e-3 there (8 oracle-driven envs).  Here the sample follows the DEVICE's own trajectories through their falls; a step beyond 1e-3 is
    # accepted (up to 2e-2, at most 0.2 % of the live steps) only where single precision itself does not hold the step: the oracle narrowed to float, started from
    # the same state, misses its fp64 self by > 2e-5 -- ten times its typical 2e-6 (measured pairs are printed above: walk 8.5e-3|1.3e-4, 2.5e-3|5.9e-5, dog 2.1e-3|2.1e-3)
    worst = np.argwhere(alive & (dr > 1e-3))
    assert len(worst) <= 0.002 * alive.sum() + 1
    for k, j in worst:
        assert d32[k, j] > 2e-5 and dr[k, j] < 2e-2, (k, j, dr[k, j], d32[k, j])
    env.close()


# ---------------------------------------------------------------------------------------------------------------------------
# Round 6: the same two checks where a LEARNER drives the envs -- explicit actions through dm_step_batch(actions_dev) (cDeepMimicCore::SetAction,
# env/deepmimic_env.py:88 -> sim/CtPDController.cpp:97-166 on the device) at the measured shape, on the state distribution the actions make:
#   feed "a2"      stream A2 (deepmimic_amd/streams.py): mocap tracking + Philox N(0, 0.05^2) noise keyed by (global env id, control step), made on the host;
#   feed "policy"  the on-device actor (dm_policy_forward, random init, SAMPLED actions) reading the states the step kernel wrote: bench.py's closed loop.
#                  Its characters tumble: pairs of the two-per-wave kernel with a character beyond 32 constraint rows run the 64-lane ClsBipedFb fallback
#                  (dm_device_duo.h), counted per env by dm_get_debug "fallback".
# The 4096-env batch is stepped through DEVICE pointers (torch tensors), the 64-env contexts and the oracles get the same action rows as host arrays.
class _Driven:
    """a 4096-env batch (one context or two groups) stepped with explicit actions held in device memory"""

    def __init__(self, kind, t, base, lib, feed, precision=32):
        import torch
        from deepmimic_amd.policy import Policy, random_weights
        self.torch, self.t, self.base, self.feed = torch, t, base, feed
        self.env = _make(kind, t, N, base, lib, precision)
        e = self.env
        self.on_gpu = torch.cuda.is_available()        # (the emulator harness drives this class with host tensors: tools/emu_check_driven.py)
        dev = torch.device("cuda" if self.on_gpu else "cpu")
        f32, i32 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.int32, device=dev)
        self.st = torch.zeros((N, e.S), **f32); self.ac = torch.zeros((N, e.A), **f32); self.rw = torch.zeros(N, **f32)
        self.tm = torch.zeros(N, **i32); self.vd = torch.zeros(N, **i32); self.en = torch.zeros(N, **i32)
        self.ptrs = (self.st.data_ptr(), self.rw.data_ptr(), self.tm.data_ptr(), self.vd.data_ptr(), self.en.data_ptr())
        self.pol = None
        if feed == "policy":
            one = e.envs[0] if hasattr(e, "envs") else e
            offs = one.offsets_scales()
            w = random_weights(e.S, e.A, seed=0)                    # bench.py closed_loop's actor
            w["s_mean"] = -offs["state_offset"].astype(np.float32); w["s_std"] = (1.0 / offs["state_scale"]).astype(np.float32)
            w["a_mean"] = -offs["action_offset"].astype(np.float32); w["a_std"] = (1.0 / offs["action_scale"]).astype(np.float32)
            self.pol = Policy(w, lib_path=lib)
        else:
            from oracle_lib import Oracle
            self.orc = Oracle(t)
        e.step_device(0, *self.ptrs, n_updates=0)                   # RecordState of the reset envs
        self.sync()

    def sync(self):
        self.env.synchronize()
        if self.on_gpu:
            self.torch.cuda.synchronize()

    def actions(self, k, st0=None):
        """[N, A] float32 actions of control step k (left in self.ac on the device too)"""
        if self.feed == "policy":
            self.pol.forward_device(self.st.data_ptr(), N, self.ac.data_ptr(), 0, sample=True, seed=1, step=k, env_id_offset=self.base, stream=0)
            self.sync()
            return self.ac.cpu().numpy()
        st0 = self.env.get_state() if st0 is None else st0
        a = streams.stream_a2(pc.tracking_actions(self.t, st0["clocks"][:, 0], oracle=self.orc), self.base + np.arange(N), k).astype(np.float32)
        self.ac.copy_(self.torch.from_numpy(a))
        self.sync()
        return a

    def step(self, acts=None):
        """one control step on the actions in self.ac (`acts` is their host copy: already uploaded by actions())"""
        self.env.step_device(self.ac.data_ptr(), *self.ptrs, timestep=pc.DT, n_updates=20, auto_reset=True)
        self.sync()
        return {"state": self.st.cpu().numpy(), "reward": self.rw.cpu().numpy(), "terminate": self.tm.cpu().numpy(), "valid": self.vd.cpu().numpy(),
                "episode_end": self.en.cpu().numpy()}

    def close(self):
        if self.pol is not None:
            self.pol.close()
        self.env.close()


@pytest.mark.parametrize("feed", ["a2", "policy"])
@pytest.mark.parametrize("kind", ["one", "groups2"])
@pytest.mark.parametrize("scene", SCENES)
def test_action_fed_rows_of_4096_bit_identical_to_64_env_contexts(hip_lib, scene, kind, feed):
    t = model.load_asset(scene)
    big = _Driven(kind, t, 0, hip_lib, feed)
    offs = (0, N // 2 - 4, N - 64)                    # 4096: 0 | 2044 (the last waves of group 0 and the first of group 1) | 4032
    small = [_make("one", t, 64, o, hip_lib) for o in offs]
    ends, fb = 0, np.zeros(3)
    for k in range(36):
        acts = big.actions(k)
        a = big.step()
        for j, (o, s) in enumerate(zip(offs, small)):
            b = s.step(acts[o:o + 64], pc.DT, 20, auto_reset=True)
            for key in KEYS:
                assert np.array_equal(a[key][o:o + 64], b[key]), (scene, kind, feed, k, o, key)
            ends += int(b["episode_end"].sum())
    sa = big.env.get_state()
    fa, xa = big.env.debug("fallback"), big.env.debug("borrowed")
    for j, (o, s) in enumerate(zip(offs, small)):
        sb = s.get_state()
        for key in sb:
            assert np.array_equal(sa[key][o:o + 64], sb[key]), (scene, kind, feed, o, key)
        # the same substeps left the 32-row register path in either batch size: on borrowed lanes (one character beyond 32 rows, the pair within 64: DuoSim's
        # lane-borrowing path, round 6) or on the 64-lane fallback
        assert np.array_equal(fa[o:o + 64], s.debug("fallback")) and np.array_equal(xa[o:o + 64], s.debug("borrowed"))
        fb[j] = fa[o:o + 64].sum() + xa[o:o + 64].sum()
    print("%s %s %s: episode ends %d, substeps of the compared rows beyond 32 rows %s (whole batch per env-step: %.4f on borrowed lanes, %.4f on the fallback)"
          % (scene, kind, feed, ends, fb, xa.sum() / (N * 36), fa.sum() / (N * 36)))
    assert ends > 0, "the compared rows must have crossed episode ends"
    if feed == "policy" and scene == "humanoid3d_spinkick":
        assert fb.sum() > 0, "the compared rows must have left the 32-row register path"
    big.close()
    for s in small:
        s.close()


# 32 sampled global rows: both halves of a wavefront, low / middle / high block indices, the boundary of the two groups, the very last env
SAMPLE_DRIVEN = sorted(set(SAMPLE + [2, 3, 64, 257, 898, 1411, 1666, 2046, 2049, 2560, 3301, 3333, 3838, 4000, 4032, 4093]))


@pytest.mark.parametrize("feed", ["a2", "policy"])
@pytest.mark.parametrize("kind", ["one", "groups2"])
@pytest.mark.parametrize("scene,steps", [("humanoid3d_walk", 60), ("humanoid3d_spinkick", 60), ("dog3d_pace", 30)])
def test_action_fed_sampled_envs_of_4096_vs_oracle(hip_lib, scene, steps, kind, feed):
    t = model.load_asset(scene)
    big = _Driven(kind, t, 0, hip_lib, feed)
    for k in range(30):                                # into the episode mixture the actions make (a random policy's characters fall within a second)
        big.actions(k); big.step()
    ids = np.array([i for i in SAMPLE_DRIVEN if i < N])
    cnt = lambda: big.env.debug("fallback") + big.env.debug("borrowed")      # substeps beyond the 32-row register path: 64-lane fallback + borrowed lanes (round 6)
    fb_prev = [cnt()]
    fb_steps = np.zeros((steps, ids.size))

    def on_step(k, st0, out):
        f = cnt()
        # a reset does not touch the counters; dm_set_state is never called here: the difference is this step's substeps of either kind
        fb_steps[k] = (f - fb_prev[0])[ids]; fb_prev[0] = f

    dr, ds, alive, ok, ends, d32 = pc.sampled_compare(big.env.get_state, big.step, t, ids, steps, conditioning=True,
                                                      actions=lambda k, st0: big.actions(30 + k, st0), on_step=on_step)
    live, sl = dr[alive], ds[alive & np.isfinite(ds)]
    on_fb = alive & (fb_steps > 0)
    print("%s %s %s: live %d/%d, ends %d, reward MAE %.2e p99 %.2e max %.2e n>1e-4 %d; state mean %.2e p99 %.2e max %.2e; sampled steps with substeps beyond 32 rows %d (MAE there %.2e, max %.2e)"
          % (scene, kind, feed, alive.sum(), dr.size, ends, live.mean(), np.quantile(live, 0.99), live.max(), (live > 1e-4).sum(), sl.mean(), np.quantile(sl, 0.99), sl.max(),
             on_fb.sum(), dr[on_fb].mean() if on_fb.any() else 0.0, dr[on_fb].max(initial=0.0)))
    big_ = np.argwhere(alive & (dr > 1e-4))
    print("  steps beyond 1e-4 (device vs fp64 oracle | the oracle's own fp32 build vs fp64 oracle | substeps beyond 32 rows): " + ", ".join("%.1e|%.1e|%d" % (dr[k, j], d32[k, j], fb_steps[k, j]) for k, j in big_))
    assert ok, "terminate / valid / episode_end differ from the oracle"
    assert alive.mean() > 0.5 and ends > 0
    assert dr[~alive].max(initial=0.0) < 1e-6
    if feed == "policy" and scene == "humanoid3d_spinkick":
        # spinkick under the random actor: 0.35 % of the pair-substeps of the batch have a character beyond 32 rows (two flat feet + two or three self contacts = 34 / 37
        # rows), concentrated in about 1 % of the pairs (profiles/r06_closed_loop_spinkick.json): they run on borrowed lanes (the 64-lane fallback itself -- a character
        # beyond 48 rows or a pair beyond 64 -- is held to the oracle by test_parity_gpu.test_duo_heavy_contact_fallback*); walk: 0.001 %, the dog has no two-per-wave kernel
        assert on_fb.sum() >= 1, "no sampled step left the 32-row register path"
        assert dr[on_fb].mean() < 3e-5 and dr[on_fb].max() < 1e-3          # measured: MAE 9.9e-6, max 1.3e-4 over 19 such steps
    if feed == "a2":
        # noisy tracking stays on the tracking distribution: the fixed fp32 bounds of test_sampled_envs_of_4096_vs_oracle (measured: MAE 4.9e-6 / 1.3e-6 / 2.3e-6,
        # p99 6.8e-5 / 2.3e-5 / 1.9e-5, share beyond 1e-4 0.9 / 0.3 / 0.2 %)
        assert live.mean() < 1e-5 and np.quantile(live, 0.99) < 1e-4 and (live > 1e-4).mean() < 0.01
        assert sl.mean() < 5e-3 and np.quantile(sl, 0.99) < 5e-2
    else:
        # a random actor's characters tumble: every control step holds contact events (a corner crossing the contact threshold, a pair entering the capsule margin) on
        # which two correct single-precision evaluations take different sides, and 20 updates amplify it.  Measured on MI355X (32 envs x 60 / 30 steps): reward MAE
        # 1.5e-5 / 6.3e-6 / 3.3e-6 (walk / spinkick / dog), p99 1.8e-4 / 1.3e-4 / 1.7e-5, 1.3 / 1.2 / 0.5 % of the live steps beyond 1e-4; the fp64 build of the same
        # kernels on the same kind of states equals the oracle to 1e-7 (test_policy_fed_fp64_kernels_equal_the_oracle below): precision, not algorithm
        assert live.mean() < 3e-5 and np.quantile(live, 0.99) < 4e-4 and (live > 1e-4).mean() < 0.03
        assert sl.mean() < 2e-2 and np.quantile(sl, 0.99) < 0.3
    # steps beyond 1e-3: at most 0.5 % of the live steps (+ 1), none beyond 2e-2
    worst = np.argwhere(alive & (dr > 1e-3))
    assert len(worst) <= 0.005 * alive.sum() + 1, len(worst)
    assert dr.max() < 2e-2
    big.close()


@pytest.mark.parametrize("scene", ["humanoid3d_walk", "humanoid3d_spinkick"])
def test_policy_fed_fp64_kernels_equal_the_oracle(hip_lib, scene):
    """The fp64 build of the same two-per-wave kernels on policy-made states (tumbling characters, resets every ~15 steps, pairs on the 64-lane fallback): the
    typical sampled control step equals the oracle to 1e-9, i.e. the algorithm is the oracle's.  What remains are steps on which the DYNAMICS amplify rounding: a
    tumbling character in stiff multi-contact is an unstable system -- isolated on the emulator (walk, one env of a 128-env batch): |velocity difference| 5e-12 after
    1 update, 2e-10 after 3, 3e-8 after 12, 7e-6 after 16, 7e-3 after 20, a factor ~2.3 per update from fp64 rounding noise (libm / FMA contraction), reward difference
    3.9e-6.  The same steps started from fp32 rounding (1e-7) are the tail of the fp32 figures above: precision on unstable states, not a different algorithm."""
    global N
    n_keep = N
    try:
        N = 512                                         # (the fp64 kernels hold one wave per SIMD)
        t = model.load_asset(scene)
        big = _Driven("one", t, 0, hip_lib, "policy", precision=64)
        for k in range(30):
            big.actions(k); big.step()
        ids = np.arange(0, N, 11)[:40]
        fb0 = big.env.debug("fallback").sum() + big.env.debug("borrowed").sum()
        dr, ds, alive, ok, ends = pc.sampled_compare(big.env.get_state, big.step, t, ids, 30, actions=lambda k, st0: big.actions(30 + k, st0))
        fb = big.env.debug("fallback").sum() + big.env.debug("borrowed").sum() - fb0
        live = dr[alive]
        print("%s fp64: live %d/%d, ends %d, |reward diff| median %.2e p90 %.2e p99 %.2e max %.2e, max rel state diff %.2e, substeps of the batch beyond 32 rows %d"
              % (scene, alive.sum(), dr.size, ends, np.median(live), np.quantile(live, 0.9), np.quantile(live, 0.99), live.max(), np.nanmax(ds), fb))
        assert ok and ends > 0 and alive.mean() > 0.5
        if scene == "humanoid3d_spinkick":
            assert fb > 0
        # measured on MI355X: max 3.3e-4 / 1.5e-4 (walk / spinkick: the unstable steps of the docstring)
        assert np.median(live) < 1e-7 and np.quantile(live, 0.9) < 1e-6 and np.quantile(live, 0.99) < 1e-4 and live.max() < 5e-3
        big.close()
    finally:
        N = n_keep
