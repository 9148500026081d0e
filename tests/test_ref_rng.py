"""DM_RNG=reference: the one-env cDeepMimicCore facade draws its clip times and episode limits from the reference's own
generator in the reference's call order.  Checked bit for bit against the reference's compiled cMathUtil / cRand / cTimer (oracle/_ref:
util/MathUtil.cpp, util/Rand.cpp, util/Timer.cpp built unmodified; ref_glue.cpp:ref_rng_session issues the calls in the order read off
DeepMimicCore.cpp, scenes/Scene.cpp, RLSceneSimChar.cpp, SceneImitate.cpp, sim/Ground.cpp) over 50 resets, for the uniform timer with and
without a range, the exponential timer, annealing in between, test mode, and a negative seed."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import ref_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPAT = os.path.join(ROOT, "deepmimic_amd", "compat")
pytestmark = pytest.mark.skipif(not ref_lib.ref_available(), reason="oracle/_ref neither prebuilt nor buildable (no /root/reference)")


def _core_module():
    if COMPAT not in sys.path:
        sys.path.insert(0, COMPAT)
    from DeepMimicCore import DeepMimicCore
    return DeepMimicCore


def _session(ref, seed, ttype, tparams, dur, n, test_mode, test_max):
    out = np.zeros((n, 2)); init = np.zeros(2)
    tp = np.ascontiguousarray(tparams, dtype=np.float64)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    ref.ref_rng_session(C.c_int(seed), C.c_int(ttype), dp(tp), C.c_double(dur), n, int(test_mode), C.c_double(test_max), dp(out), dp(init))
    return out, init


CASES = [
    # seed, timer type, (lim min, lim max, exp), (end min, end max, end exp), anneal samples, test mode
    (7, "uniform", (0.5, 0.5, 1.0), (20.0, 20.0, 1.0), 32000000, False),          # the shipped imitate arg files: no timer draw ever happens
    (123456789, "uniform", (0.5, 2.5, 1.0), (5.0, 20.0, 1.0), 1000, False),       # a real range, annealed between resets
    (-3, "uniform", (1.0, 4.0, 1.0), (1.0, 4.0, 1.0), 0, True),                   # negative seed (int -> unsigned long), test mode pins the limit
    (99, "exp", (0.5, 8.0, 2.0), (0.5, 30.0, 6.0), 500, False),                   # --timer_type exp
]


@pytest.mark.parametrize("seed,ttype,begin,end,anneal,test_mode", CASES)
def test_facade_reset_draws_equal_compiled_reference(emu_lib, monkeypatch, seed, ttype, begin, end, anneal, test_mode):
    from deepmimic_amd import model
    monkeypatch.setenv("DM_HIP_LIB", emu_lib); monkeypatch.setenv("DM_PRECISION", "64"); monkeypatch.delenv("DM_RNG", raising=False)
    ref = ref_lib.load("ref")
    mod = _core_module()
    t = model.load_asset("humanoid3d_walk")
    c = t.cfg
    c.scene = "imitate_amp"                       # (also serves RecordAMPObsExpert from the scene generator)
    c.timer_type = ttype
    c.time_lim_min, c.time_lim_max, c.time_lim_exp = begin
    c.time_end_lim_min, c.time_end_lim_max, c.time_end_lim_exp = end
    c.anneal_samples = anneal
    core = mod.cDeepMimicCore(False)
    core.SeedRand(seed); core.LoadTables(t, 10); core.Init()
    assert core._ref_active
    n = 50
    counts = [0] + [int(anneal * k / n) if anneal else 0 for k in range(n)]      # sample count in force at Init and at each reset
    got, params = [], []
    clk0 = core._env.get_state()["clocks"][0]
    init_limit = float(clk0[4])
    assert float(clk0[0]) == 0.0                  # after Init the clip stands at time 0; the driver resets before it steps
    if test_mode:
        core.SetMode(1)
    for k in range(n):
        core.SetSampleCount(counts[k + 1])
        core.Reset()
        clk = core._env.get_state()["clocks"][0]
        got.append((float(clk[0]), float(clk[4])))
    for sc in counts:
        lo, hi = model.timer_limits(c, False, sc)
        params.append((lo, hi, model.timer_exp(c, False, sc)))
    want, init = _session(ref, seed, 1 if ttype == "exp" else 0, params, float(core._env.duration), n, test_mode, float(end[1]))
    got = np.array(got)
    assert np.array_equal(got, want), np.abs(got - want).max()
    assert init_limit == init[0]
    assert len(np.unique(got[:, 0])) == n and got[:, 0].min() >= 0 and got[:, 0].max() < core._env.duration
    if ttype == "uniform" and begin[0] != begin[1] and not test_mode:
        assert len(np.unique(got[:, 1])) > n // 2
    # the scene generator (cScene::mRand, seeded by the second draw of the session) serves the expert sample's clip time
    import types
    seen = {}
    orig = core._env.amp_expert
    core._env.amp_expert = types.MethodType(lambda self, n_, times=None, gh=None: (seen.setdefault("t", float(times[0])), orig(n_, times, gh))[1], core._env)
    core.RecordAMPObsExpert(0)
    assert seen["t"] == init[1]


def test_refrand_matches_compiled_crand_calls(emu_lib):
    """every dm_refrand_* entry point against the compiled cRand through cMathUtil, interleaved, 2000 calls"""
    from deepmimic_amd.core import RefRand
    ref = ref_lib.load("ref")
    ref.ref_math_rand.restype = C.c_double
    r = RefRand(0, lib_path=emu_lib)
    for seed in (1, 2 ** 31 + 5, 2 ** 64 - 3):
        r.seed(seed); r.rand_int(); ref.ref_math_seed(C.c_ulong(seed))         # cMathUtil::SeedRand = Seed + one RandInt
        rng = np.random.default_rng(3)
        for i in range(2000):
            op = int(rng.integers(0, 6))
            a, b = sorted(rng.normal(size=2) * 10)
            if op == 0: x, y = r.rand_double(a, b), ref.ref_math_rand(0, C.c_double(a), C.c_double(b))
            elif op == 1: x, y = r.rand_exp(abs(a) + 0.1), ref.ref_math_rand(1, C.c_double(abs(a) + 0.1), C.c_double(0))
            elif op == 2: x, y = r.rand_norm(a, abs(b) + 0.1), ref.ref_math_rand(2, C.c_double(a), C.c_double(abs(b) + 0.1))
            elif op == 3: x, y = float(r.rand_int()), ref.ref_math_rand(3, C.c_double(0), C.c_double(0))
            elif op == 4: x, y = float(r.rand_int_range(-5, 40)), ref.ref_math_rand(4, C.c_double(-5), C.c_double(40))
            else: x, y = float(r.rand_uint()), ref.ref_math_rand(5, C.c_double(0), C.c_double(0))
            assert x == y, (seed, i, op, x, y)


def test_counter_mode_is_still_available(emu_lib, monkeypatch):
    from deepmimic_amd import model
    monkeypatch.setenv("DM_HIP_LIB", emu_lib); monkeypatch.setenv("DM_PRECISION", "64"); monkeypatch.setenv("DM_RNG", "counter")
    mod = _core_module()
    core = mod.cDeepMimicCore(False)
    core.SeedRand(5); core.LoadTables(model.load_asset("humanoid3d_walk"), 10); core.Init()
    assert not core._ref_active and core._grand is None
    assert float(core._env.get_state()["clocks"][0][0]) > 0.0          # Init reset the env with the device's counter-based draw
