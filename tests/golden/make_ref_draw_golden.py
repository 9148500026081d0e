#!/usr/bin/env python3
"""Writes tests/golden/ref_draws.npz: what the reference's compiled scene classes (oracle/_ref, driven by tests/test_ref_draw_order.py RefSession) drew and
held -- clip, clip time, episode limit, yaw, target heading / speed / timer, strike target and hit state, expert samples -- over the sessions of SESSIONS,
logged in call order, so that the GPU box (no reference checkout) can hold the HIP kernels' draw tape to the same numbers (tests/test_ref_draw_order_gpu.py).
The sessions run on the CPU emulator build of the kernels in fp64; every logged value is asserted equal to the device's on the way (the CPU test's checks).

    python tests/golden/make_ref_draw_golden.py        (needs /root/reference and oracle/_ref)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["DM_ALLOW_EMULATOR"] = "1"

import ref_lib  # noqa: E402
import test_ref_draw_order as T  # noqa: E402
from deepmimic_amd import model  # noqa: E402


class Env:
    def setenv(self, k, v): os.environ[k] = v
    def delenv(self, k, raising=False): os.environ.pop(k, None)


def main():
    emu = os.path.join(ROOT, "tests", "emu", "libdm_emu.so")
    mod = T._core_module()
    store = {}
    for key, (asset, args, seed, n_resets, steps, anneal) in T.GOLDEN_SESSIONS.items():
        rec = T.Recorder(T.RefSession(ref_lib.load("ref"), args(), seed))
        T._run(mod, emu, args(), seed, Env(), n_resets=n_resets, steps=steps, anneal_at=anneal, tables=T.golden_tables(asset), provider=rec, policy_scale=T.GOLDEN_POLICY_SCALE.get(key, 0.0),
               pos_tol=1e-6 if key == "dribble" else 1e-9)
        rec.save(None, key, store)
        print(key, len(rec.tag), "calls logged")
    np.savez_compressed(os.path.join(HERE, "ref_draws.npz"), **store)


if __name__ == "__main__":
    main()
