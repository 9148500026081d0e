#!/usr/bin/env python3
"""Compile the reference's data files for the BASELINE configs into in-tree scene tables.

Reads /root/reference/{args,data} (only available in the build container) and writes
deepmimic_amd/assets/<name>.json in the `deepmimic_amd.scene_tables.v1` format (flat
numeric matrices in the reference's in-memory layout).  The GPU box has no /root/reference,
so tests / bench / smoke load these compiled tables instead.
"""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from deepmimic_amd import model

REF = os.environ.get("DM_REFERENCE", "/root/reference")
SCENES = {
    # name: (arg file, motion override)
    "humanoid3d_walk": ("args/run_humanoid3d_walk_args.txt", None),
    "humanoid3d_spinkick": ("args/train_humanoid3d_spinkick_args.txt", None),
    "dog3d_pace": ("args/train_dog3d_pace_args.txt", None),
    "humanoid3d_run": ("args/run_humanoid3d_run_args.txt", None),
    "humanoid3d_backflip": ("args/run_humanoid3d_backflip_args.txt", None),
    "dog3d_spin": ("args/run_dog3d_spin_args.txt", None),          # sync_char_root_rot = true
    # goal-conditioned AMP task scenes: `--kin_ctrl clips` datasets, enable_rand_rot_reset
    "amp_heading_zombie": ("args/train_amp_heading_humanoid3d_zombie_args.txt", None),
    "amp_target_zombie": ("args/train_amp_target_humanoid3d_zombie_args.txt", None),
    # a 4-clip dataset (two looping, two non-looping clips) under the heading task: exercises clip selection by weight
    "amp_heading_clips4": ("args/train_amp_heading_getup_humanoid3d_locomotion_getup_args.txt", ["--scene", "heading_amp"]),
    # heading_amp_getup as shipped (run, walk + the two get-up clips; getup_motion_ids 2 3)
    "amp_heading_getup": ("args/train_amp_heading_getup_humanoid3d_locomotion_getup_args.txt", None),
    # strike_amp: the shipped dataset (humanoid3d_clips_walk_punch.txt) names sie / amass clips that are not in the repository; the scene
    # keys are the arg file's, the dataset is a stand-in made of the two shipped clips of the same kind (tools/datasets/)
    # dribble_amp: the ball is a free rigid sphere in the world
    "amp_dribble_zombie": ("args/train_amp_dribble_humanoid3d_zombie_args.txt", None),
    "amp_strike_punch": ("args/train_amp_strike_humanoid3d_walk_punch_args.txt",
                         ["--motion_file", os.path.join(os.path.dirname(os.path.abspath(__file__)), "datasets", "humanoid3d_clips_walk_punch_local.txt")]),
}

def main():
    out_dir = model.ASSET_DIR
    os.makedirs(out_dir, exist_ok=True)
    for name, (arg_file, extra) in SCENES.items():
        t = model.load_scene_from_args(["--arg_file", arg_file] + (extra or []), data_root=REF)
        path = os.path.join(out_dir, name + ".json")
        with open(path, "w") as f:
            json.dump(t.to_json(), f, separators=(",", ":"))
        print("%-22s J=%d P=%d A=%d S=%d F=%d loop=%s -> %s" % (
            name, t.num_joints, t.pose_dim, t.action_dim, t.state_dim, t.frames.shape[0], t.loop, path))

if __name__ == "__main__":
    main()
