#!/usr/bin/env python
"""Dump a fixed-action rollout of the HIP path as a replay bundle for a real DeepMimicCore + Bullet 2.88 build.

SURVEY.md 8(c)(3): parity against Bullet itself cannot be measured where Bullet cannot be built, so this tool writes
everything a reference build needs to re-run config 1 and diff: the scene args, the action fed at every control step
and the character state after it, the latter in the reference's own WriteState JSON ({"Pose","Vel"},
anim/Character.cpp:434-443 -- loadable with cCharacter::ReadState / --state_files).

    python tools/replay_dump.py --asset humanoid3d_walk --steps 300 --stream A1 --out gpurun_out/replay_walk

Which rigid-body step to diff against Bullet: `--physics 2` (DM-physics v2: Bullet's manifold semantics as recalled -- one new support
point per narrowphase call into a persistent <= 4-point manifold per link, both rows of every revolute limit) is the
specification meant to be CLOSER to Bullet 2.88 and is what a Bullet owner should diff first; `--physics 1` (the default of the library,
the fast path of the benchmarks) regenerates the analytic contact set every substep.  The bundle records which one produced it
(`meta` of the bundle).

On the reference side (a maintainer's ~20-line driver over the SWIG module): ParseArgs(scene_args); Init(); Reset() with
kin time 0 (or ReadState(state_0000.json)); per step k: SetAction(0, actions[k]); 20 x Update(1/600); WriteState and
compare with state_%04d.json, CalcReward with rewards[k].
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepmimic_amd import formats, model, streams          # noqa: E402
from deepmimic_amd.core import BatchEnv                    # noqa: E402


def run(tables, scene_args, steps, stream, precision, out_dir, lib_path=None, t0=0.0, physics=1):
    env = BatchEnv(tables, 1, precision=precision, lib_path=lib_path, wave_packing=1, physics=physics)
    env.reset(kin_times=[t0], max_times=np.inf)
    st = env.get_state()
    poses, vels, actions, rewards, term = [st["pose"][0].copy()], [st["vel"][0].copy()], [], [], []
    for k in range(steps):
        if stream == "A0":
            a = streams.stream_a0(1, env.A)[0]
        else:
            # the device encodes the clip pose at its own kin time; recover that action from a scratch env's PD target
            probe = BatchEnv(tables, 1, precision=64, lib_path=lib_path, wave_packing=1)
            probe.reset(kin_times=[st["clocks"][0][0]], max_times=np.inf)
            probe.step(None, 1.0 / 600, 1, open_loop=True)
            a = streams.pose_to_action(tables, probe.get_state()["tar"][0])
            probe.close()
            if stream == "A2":
                a = streams.stream_a2(a[None], [0], k)[0]
        a = a.astype(np.float32).astype(np.float64)
        out = env.step(a[None], 1.0 / 600, 20)
        st = env.get_state()
        actions.append(a); rewards.append(float(out["reward"][0])); term.append(int(out["terminate"][0]))
        poses.append(st["pose"][0].copy()); vels.append(st["vel"][0].copy())
    formats.write_replay_bundle(out_dir, scene_args, actions, poses, vels, rewards, term, 1.0 / 600, 20,
                                meta={"stream": stream, "precision": precision, "t0": t0, "producer": "deepmimic_amd HIP path", "physics": "DM-physics v%d" % physics})
    return np.array(actions), np.array(poses), np.array(vels), np.array(rewards), np.array(term)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asset", default="humanoid3d_walk")
    ap.add_argument("--arg-file", default=None, help="reference arg file (resolved under --data-root) instead of --asset")
    ap.add_argument("--data-root", default=".")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--stream", choices=["A0", "A1", "A2"], default="A1")
    ap.add_argument("--precision", type=int, default=32)
    ap.add_argument("--physics", type=int, default=1, choices=[1, 2], help="DM-physics version of the rigid-body step (2: the one to diff against Bullet first)")
    ap.add_argument("--out", required=True)
    ap.add_argument("--lib", default=None)
    a = ap.parse_args()
    if a.arg_file:
        args = ["--arg_file", a.arg_file]
        tables = model.load_scene_from_args(args, a.data_root)
    else:
        args = ["--asset", a.asset]
        tables = model.load_asset(a.asset)
    _, poses, _, rewards, term = run(tables, args, a.steps, a.stream, a.precision, a.out, a.lib, physics=a.physics)
    print("wrote %d states to %s; mean reward %.4f; terminated at step %s" %
          (len(poses), a.out, rewards.mean(), (int(np.argmax(term != 0)) if (term != 0).any() else None)))


if __name__ == "__main__":
    main()
