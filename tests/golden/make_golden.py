#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_rollouts.json from the fp64 CPU oracle (TEST INFRASTRUCTURE).

The reference ships no tests or golden vectors and cannot be built or imported here, so these fixtures do
not pin the oracle to the reference; they freeze the oracle's own outputs on the reference's shipped data files (skeleton,
controller and motion clips compiled into deepmimic_amd/assets) so that any later change to oracle/ or to the asset
compiler is caught, and so that the GPU parity tests have committed vectors to check against on a box without the oracle
sources' provenance.  Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from deepmimic_amd import model  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

CASES = [("humanoid3d_walk", 0.0), ("humanoid3d_walk", 0.37), ("humanoid3d_spinkick", 0.0), ("dog3d_pace", 0.2),
         ("humanoid3d_run", 0.1), ("humanoid3d_backflip", 0.0), ("dog3d_spin", 0.5)]


def one_case(name, t0, steps=10):
    steps = 30 if name == "dog3d_spin" else steps      # long enough to cross a phase wrap (root heading sync)
    t = model.load_asset(name)
    o = Oracle(t)
    o.reset(t0)
    s0 = o.record_state()
    rewards, heights, ncontact = [], [], []
    for k in range(steps):
        kp, _, _ = o.kin_state()
        o.set_action(o.pose_to_action(kp))
        for u in range(20):
            o.update(1.0 / 600)
        rewards.append(o.calc_reward()); heights.append(float(o.sim_state()[0][1])); ncontact.append(int(o.contacts().sum()))
    p, v = o.sim_state()
    return {"scene": name, "t0": t0, "steps": steps, "dims": [o.J, o.P, o.A, o.S, o.F], "duration": o.duration,
            "state0": [float(x) for x in s0], "rewards": [float(r) for r in rewards], "root_height": heights,
            "links_in_contact": ncontact, "final_pose": [float(x) for x in p], "final_vel": [float(x) for x in v],
            "final_state": [float(x) for x in o.record_state()], "terminate": o.check_terminate()}


if __name__ == "__main__":
    out = [one_case(n, t0) for n, t0 in CASES]
    with open(os.path.join(HERE, "oracle_rollouts.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("wrote %d cases" % len(out))
