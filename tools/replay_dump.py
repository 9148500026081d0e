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

Two outputs:
* `--bundle out.npz` (round 5): the rollout as the REFERENCE'S OWN DRIVER sees it -- recorded through this repository's drop-in `DeepMimicCore`
  module on the SWIG surface only, with the loop of tools/ref_replay.py (`drive`): `SeedRand(--seed)`, `ParseArgs`, `Init`, `SetMode`, `Reset`,
  then per action boundary RecordState / RecordGoal / CalcReward / GetTime / SetAction and per update the episode flags, through episode ends
  and resets.  The drop-in draws its reset clip times from the reference's generator in the reference's order (DM_RNG=reference), so
  `tools/ref_replay.py out.npz` on a DeepMimicCore + Bullet build starts every episode at the same clip time and its printed differences are
  Bullet's rigid-body step against the producer's, nothing else.
* `--out dir` (rounds 1-4): actions + one `{"Pose","Vel"}` snapshot per control step in the reference's WriteState format, for `--state_files`.
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


def record(scene_args, tables, steps, stream, seed, bundle_path, mode=0, precision=32, physics=1, lib_path=None, data_root=None):
    """Drive the drop-in `DeepMimicCore` (deepmimic_amd/compat) with tools/ref_replay.drive and write what the SWIG surface reported.
    Returns the bundle dict (also written to `bundle_path` as .npz when given)."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p_ in (os.path.join(root, "deepmimic_amd", "compat"), os.path.join(root, "tools")):
        if p_ not in sys.path:
            sys.path.insert(0, p_)
    import ref_replay
    os.environ["DM_PHYSICS"] = str(int(physics)); os.environ["DM_PRECISION"] = str(int(precision)); os.environ["DM_RNG"] = "reference"
    if lib_path:
        os.environ["DM_HIP_LIB"] = lib_path
    if data_root:
        os.environ["DM_DATA_ROOT"] = data_root
    meta = {"format": "deepmimic-replay-2", "scene_args": list(scene_args), "seed": int(seed), "mode": int(mode), "timestep": 1.0 / 600, "stream": stream,
            "precision": int(precision), "physics": "DM-physics v%d" % physics, "producer": "deepmimic_amd drop-in DeepMimicCore (HIP path)"}
    core = ref_replay.make_core(meta)
    env = core._env
    J, S, A = int(tables.num_joints), int(core.GetStateSize(0)), int(core.GetActionSize(0))
    p0 = 1 if tables.enable_phase_input else 0
    meta["state_blocks"] = [p0, p0 + 1 + 9 * J, p0 + 1 + 15 * J]        # sim/CtController.cpp:281-478: [phase] | root height, J x (pos 3, rot 6) | J x (vel 3, ang vel 3) [| task block]
    kin = model.KinSampler(tables) if stream in ("A1", "A2") else None
    rng = np.random.default_rng(int(seed) + 77)
    rec = {"states": [], "goals": [], "rewards": [], "times": [], "updates_before": [], "actions": [], "end_steps": [], "end_flags": [], "end_states": [], "end_rewards": []}

    def action_of(k, state, goal):
        if stream == "A0":
            a = np.zeros(A)
        elif stream == "R":                                   # seeded N(0, 0.1^2): any scene (goal scenes have no single clip pose to track)
            a = 0.1 * rng.normal(size=A)
        else:
            st = env.get_state()
            a = streams.pose_to_action(tables, kin.pose(float(st["clocks"][0][0]), st["kin"][0][0:3], st["kin"][0][3:7]))
            if stream == "A2":
                a = streams.stream_a2(a[None], [0], k)[0]
        a = np.asarray(a, dtype=np.float32).astype(np.float64)          # what crosses the boundary
        rec["actions"].append(a)
        return a

    def on_boundary(k, r):
        rec["states"].append(r["state"]); rec["goals"].append(r["goal"]); rec["rewards"].append(r["reward"]); rec["times"].append(r["time"]); rec["updates_before"].append(r["updates_before"])

    def on_end(k, r):
        rec["end_steps"].append(k); rec["end_flags"].append([k, r["updates"], int(r["episode_end"]), int(r["valid"]), r["terminate"]])
        rec["end_states"].append(r["state"]); rec["end_rewards"].append(r["reward"])

    ref_replay.drive(core, steps, meta["timestep"], action_of, on_boundary, on_end)
    core.Shutdown()
    G = int(len(rec["goals"][0]))
    b = {"meta": np.array(json.dumps(meta)), "actions": np.array(rec["actions"]).reshape(steps, A), "states": np.array(rec["states"]).reshape(steps, S),
         "goals": np.array(rec["goals"]).reshape(steps, G), "rewards": np.array(rec["rewards"]), "times": np.array(rec["times"]),
         "updates_before": np.array(rec["updates_before"], dtype=np.int32), "end_steps": np.array(rec["end_steps"], dtype=np.int32),
         "end_flags": np.array(rec["end_flags"], dtype=np.int32).reshape(-1, 5), "end_states": np.array(rec["end_states"]).reshape(-1, S), "end_rewards": np.array(rec["end_rewards"])}
    if bundle_path:
        os.makedirs(os.path.dirname(os.path.abspath(bundle_path)), exist_ok=True)
        np.savez_compressed(bundle_path, **b)
    b["meta"] = meta
    return b


# the committed set (tests/golden/replay/): BASELINE.json's three characters x both rigid-body steps x the three action streams of the measurement contract
GOLDEN = [(n, af, v, s) for n, af in (("walk", "args/run_humanoid3d_walk_args.txt"), ("spinkick", "args/train_humanoid3d_spinkick_args.txt"), ("dog", "args/train_dog3d_pace_args.txt"))
          for v in (1, 2) for s in ("A0", "A1", "A2")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asset", default="humanoid3d_walk")
    ap.add_argument("--arg-file", default=None, help="reference arg file (resolved under --data-root; the in-tree compiled copy when the file is not there) instead of --asset")
    ap.add_argument("--data-root", default=".")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--stream", choices=["A0", "A1", "A2", "R"], default="A1")
    ap.add_argument("--precision", type=int, default=32)
    ap.add_argument("--physics", type=int, default=1, choices=[1, 2], help="DM-physics version of the rigid-body step (2: the one to diff against Bullet first)")
    ap.add_argument("--out", default=None, help="directory for the WriteState-format bundle (actions.json + state_%%04d.json)")
    ap.add_argument("--bundle", default=None, help="the driver-level bundle (.npz) for tools/ref_replay.py; needs --arg-file (what the reference's ParseArgs is given)")
    ap.add_argument("--seed", type=int, default=1, help="--bundle: SeedRand(seed) on both sides")
    ap.add_argument("--mode", type=int, default=0, help="--bundle: cRLScene::eMode (0 train: the arg file's annealed episode limits, 1 test)")
    ap.add_argument("--golden", default=None, help="write the committed set (walk / spinkick / dog x v1 / v2 x A0 / A1 / A2, 60 steps, seed 1) into this directory")
    ap.add_argument("--lib", default=None)
    a = ap.parse_args()
    if a.golden:
        for name, af, v, sname in GOLDEN:
            tables = model.load_scene_from_args(["--arg_file", af], a.data_root) if os.path.exists(os.path.join(a.data_root, af)) else model.load_asset(model.ARG_FILE_ASSETS[af])
            b = record(["--arg_file", af], tables, 60, sname, 1, os.path.join(a.golden, "%s_v%d_%s.npz" % (name, v, sname)), 0, 32, v, a.lib)
            print("%s_v%d_%s: %d episode ends, mean reward %.4f" % (name, v, sname, len(b["end_steps"]), float(np.mean(b["rewards"]))))
        return
    if a.arg_file:
        args = ["--arg_file", a.arg_file]
        tables = model.load_scene_from_args(args, a.data_root) if os.path.exists(os.path.join(a.data_root, a.arg_file)) else model.load_asset(model.ARG_FILE_ASSETS[a.arg_file])
    else:
        args = ["--asset", a.asset]
        tables = model.load_asset(a.asset)
    if a.bundle:
        if not a.arg_file:
            raise SystemExit("--bundle records what ParseArgs is given on both sides: pass --arg-file (e.g. args/run_humanoid3d_walk_args.txt)")
        b = record(args, tables, a.steps, a.stream, a.seed, a.bundle, a.mode, a.precision, a.physics, a.lib, a.data_root if a.data_root != "." else None)
        print("wrote %s: %d control steps, %d episode ends, mean reward %.4f" % (a.bundle, a.steps, len(b["end_steps"]), float(np.mean(b["rewards"]))))
    if a.out:
        _, poses, _, rewards, term = run(tables, args, a.steps, a.stream, a.precision, a.out, a.lib, physics=a.physics)
        print("wrote %d states to %s; mean reward %.4f; terminated at step %s" %
              (len(poses), a.out, rewards.mean(), (int(np.argmax(term != 0)) if (term != 0).any() else None)))


if __name__ == "__main__":
    main()
