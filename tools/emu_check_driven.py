#!/usr/bin/env python3
"""CPU harness for the action-fed tests of tests/test_parity_4096.py: the same test bodies on the emulator build with a small batch, so that their Python
is debugged before a GPU call is spent on them.  usage: emu_check_driven.py [N=128] [scene] [feed] [kind]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["DM_ALLOW_EMULATOR"] = "1"
import test_parity_4096 as T
T.N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
scene = sys.argv[2] if len(sys.argv) > 2 else "humanoid3d_walk"
feed = sys.argv[3] if len(sys.argv) > 3 else "policy"
kind = sys.argv[4] if len(sys.argv) > 4 else "groups2"
lib = os.path.join(ROOT, "tests", "emu", "libdm_emu.so")
which = sys.argv[5] if len(sys.argv) > 5 else "both"
if which in ("both", "rows"):
    T.test_action_fed_rows_of_4096_bit_identical_to_64_env_contexts.__wrapped__ if False else None
    T.test_action_fed_rows_of_4096_bit_identical_to_64_env_contexts(lib, scene, kind, feed)
if which in ("both", "oracle"):
    T.test_action_fed_sampled_envs_of_4096_vs_oracle(lib, scene, 12, kind, feed)
print("ok")
