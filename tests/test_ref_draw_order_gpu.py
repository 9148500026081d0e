"""The draw tape on the HIP kernels: the one-env cDeepMimicCore facade with `DM_RNG=reference` on libdm_hip.so against the log of the reference's compiled scene
classes (tests/golden/ref_draws.npz, written by tests/golden/make_ref_draw_golden.py where the reference checkout exists; tests/test_ref_draw_order.py is the
live comparison on the CPU emulator build of the same kernels).  fp64 kernels: every draw-determined value EQUAL, and the compiled scenes' CalcReward / RecordGoal /
RecordState / RecordAMPObsAgent of every action boundary of the logged sessions within 1e-3 (the log was taken on the emulator's trajectory; the GPU's fp64 trajectory
leaves it by rounding, which contact amplifies over an episode: 5e-4 in a velocity feature after 140 updates of a falling character); fp32 production kernels: the draws, equal to float accuracy (their scene parameters are floats)."""
import os

import numpy as np
import pytest

import test_ref_draw_order as T
from deepmimic_amd import model

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_draws.npz")


@pytest.mark.gpu
@pytest.mark.parametrize("key", sorted(T.GOLDEN_SESSIONS))
@pytest.mark.parametrize("precision", ["64", "32"])
def test_draw_tape_on_hip_kernels_equals_reference_log(hip_lib, monkeypatch, key, precision):
    asset, args, seed, n_resets, steps, anneal = T.GOLDEN_SESSIONS[key]
    store = np.load(GOLDEN)
    if precision == "32":
        if key in T.GOLDEN_POLICY_SCALE:
            pytest.skip("episodes of this session end in falls: an fp32 character does not fall at the update the logged fp64 one did")
        n_resets = 3      # the fp32 character drifts from the logged fp64 one: short, time-limited episodes only
    T._run(T._core_module(), hip_lib, args(), seed, monkeypatch, n_resets=n_resets, steps=steps, anneal_at=anneal, tables=T.golden_tables(asset),
           provider=T.Replay(store, key, T.GOLDEN_KINDS[key], values=precision == "64"), precision=precision, exact=precision == "64", pos_tol=1e-6,
           policy_scale=T.GOLDEN_POLICY_SCALE.get(key, 0.0), val_tol=500.0)
