"""On-disk formats either side of the hot path, byte-compatible with the reference's readers.

* character state snapshot `{"Pose": [...], "Vel": [...]}` -- cCharacter::WriteState / ReadState / BuildStateJson
  (anim/Character.cpp:320-385,434-443); what `--state_files` (scenes/SceneSimChar.cpp:382-401) loads.
* motion clip `{"Loop": "wrap"|"none", "EnableCycleSyncRoot*": bool, "Frames": [[dt, pose...], ...]}` -- cMotion::Output
  (anim/Motion.cpp:581-646); the last frame's duration is written as 0 as the reference does.
* replay bundle (ours): `actions.json` + one state snapshot per control step, so that somebody with a DeepMimicCore +
  Bullet 2.88 build can feed the same actions to the real reference and diff (the parity leg this repo cannot run).
"""
import json
import os
from typing import Optional, Sequence

import numpy as np


def _vec(v) -> str:
    return "[" + ", ".join(repr(float(x)) for x in np.asarray(v, dtype=np.float64).ravel()) + "]"


def state_json(pose, vel) -> str:
    return "{\n\"Pose\":" + _vec(pose) + ",\n\"Vel\":" + _vec(vel) + "\n}"


def write_state(path: str, pose, vel) -> None:
    with open(path, "w") as f:
        f.write(state_json(pose, vel))


def read_state(path: str, pose_dim: Optional[int] = None):
    """-> (pose, vel); either may be None when its key is absent (ReadState applies only the keys present)."""
    with open(path) as f:
        d = json.load(f)
    out = []
    for key in ("Pose", "Vel"):
        v = d.get(key)
        if v is None:
            out.append(None)
            continue
        v = np.asarray(v, dtype=np.float64)
        if pose_dim is not None and v.shape != (pose_dim,):
            raise ValueError("%s has %d entries, character has %d dofs" % (key, v.size, pose_dim))
        out.append(v)
    return tuple(out)


def write_motion(path: str, frames, loop: bool, sync_root_pos: bool = True, sync_root_rot: bool = False,
                 sync_root_height: bool = False) -> None:
    """frames [F, 1+P]: column 0 = frame duration (as in the clip files), then the pose."""
    frames = np.asarray(frames, dtype=np.float64)
    b = lambda x: "true" if x else "false"
    with open(path, "w") as f:
        f.write("{\n\"Loop\": \"%s\",\n" % ("wrap" if loop else "none"))
        f.write("\"EnableCycleSyncRootPos\": %s,\n\"EnableCycleSyncRootRot\": %s,\n\"EnableCycleSyncRootHeight\": %s,\n" %
                (b(sync_root_pos), b(sync_root_rot), b(sync_root_height)))
        f.write("\n\"Frames\":\n[\n")
        rows = []
        for i, fr in enumerate(frames):
            fr = fr.copy()
            if i == len(frames) - 1:
                fr[0] = 0.0
            rows.append(_vec(fr))
        f.write(",\n".join(rows))
        f.write("\n]\n}")


def read_motion(path: str):
    with open(path) as f:
        d = json.load(f)
    loop = d.get("Loop", "none")
    if loop not in ("none", "wrap"):
        raise ValueError("unsupported loop mode %r" % loop)
    return np.asarray(d["Frames"], dtype=np.float64), loop == "wrap"


def write_replay_bundle(out_dir: str, scene_args: Sequence[str], actions, poses, vels, rewards, terminate,
                        timestep: float, updates_per_step: int, meta: Optional[dict] = None) -> None:
    """actions [K, A] fed at control step k; poses/vels [K+1, P] (index 0 = state after Reset, k+1 = after step k)."""
    os.makedirs(out_dir, exist_ok=True)
    actions = np.asarray(actions, dtype=np.float64)
    with open(os.path.join(out_dir, "actions.json"), "w") as f:
        json.dump({"scene_args": list(scene_args), "timestep": timestep, "updates_per_step": updates_per_step,
                   "actions": actions.tolist(), "rewards": np.asarray(rewards, dtype=np.float64).tolist(),
                   "terminate": [int(x) for x in terminate], "meta": meta or {}}, f)
    for k in range(len(poses)):
        write_state(os.path.join(out_dir, "state_%04d.json" % k), poses[k], vels[k])


def read_replay_bundle(out_dir: str):
    with open(os.path.join(out_dir, "actions.json")) as f:
        d = json.load(f)
    poses, vels, k = [], [], 0
    while os.path.exists(os.path.join(out_dir, "state_%04d.json" % k)):
        p, v = read_state(os.path.join(out_dir, "state_%04d.json" % k))
        poses.append(p); vels.append(v); k += 1
    d["poses"], d["vels"] = np.array(poses), np.array(vels)
    d["actions"] = np.array(d["actions"])
    return d
