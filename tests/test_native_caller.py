"""A C99 program with no Python in the process drives the C-ABI (the reference's native caller is DeepMimicCore/Main.cpp:38-75,
97-124): tests/native/smoke.c is compiled with `gcc -std=c99 -Iinclude`, linked against the library, fed the flat scene tables as a blob
(tools/dump_tables.py) and must print exactly the rewards / flags / observation checksum the ctypes binding gets from the same library.
CPU: against the emulator build of the sources (same exported symbols); GPU: against libdm_hip.so."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _build(lib, out):
    d, name = os.path.split(lib)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "smoke.c"),
                           "-o", out, lib, "-Wl,-rpath," + d])


def _check(lib, tmp_path, scene, n, steps, precision):
    import dump_tables
    from deepmimic_amd import model
    from deepmimic_amd.core import BatchEnv
    t = model.load_asset(scene)
    blob = str(tmp_path / (scene + ".dmtbl")); exe = str(tmp_path / "smoke")
    assert dump_tables.dump(t, blob) > 1000
    _build(lib, exe)
    p = subprocess.run([exe, blob, str(n), str(steps), str(precision)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.strip().splitlines()
    assert lines[-1] == "ok" and lines[0].startswith("dims S %d " % t.state_dim)
    env = BatchEnv(t, n, seed=1234, precision=precision, lib_path=lib)
    env.reset(kin_times=env.duration * np.arange(n) / n, max_times=1e300)
    for k in range(steps):
        out = env.step(None, 1.0 / 600, 20, open_loop=True)
        toks = lines[1 + k].split()
        assert toks[0] == "step" and int(toks[1]) == k
        for e in range(n):
            r, fl = toks[2 + e].split("/")
            assert np.float32(float(r)) == out["reward"][e], (k, e)
            assert fl == "%d%d%d" % (out["terminate"][e], out["valid"][e], out["episode_end"][e])
        cs = 0.0
        for v in out["state"].ravel().tolist():                                          # the C loop's order: sequential, row major (numpy's sum is pairwise)
            cs += v
        assert float(toks[-1]) == cs
    return lines


def test_blob_round_trip_header():
    """the blob's pointer table covers every pointer member that is set, its struct image is sizeof(dm_scene_tables) of the header"""
    import struct
    import dump_tables
    from deepmimic_amd import core, model
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "t.dmtbl")
        dump_tables.dump(model.load_asset("amp_heading_getup"), path)            # a multi-clip dataset: the clip arrays travel too
        b = open(path, "rb").read()
    assert b[:8] == dump_tables.MAGIC
    abi, ssize, n = struct.unpack_from("<iiQ", b, 8)
    assert abi == core.ABI_VERSION and ssize == __import__("ctypes").sizeof(core._SceneTables) and n == 8      # joint_mat, body_defs, pd_params, frames, fall_mask, clip_*


@pytest.mark.parametrize("scene,n,precision", [("humanoid3d_walk", 4, 64), ("dog3d_pace", 2, 64)])
def test_c_caller_matches_python_binding_emulator(emu_lib, tmp_path, scene, n, precision):
    _check(emu_lib, tmp_path, scene, n, 3, precision)


@pytest.mark.gpu
@pytest.mark.parametrize("scene,n", [("humanoid3d_walk", 64), ("dog3d_pace", 16)])
def test_c_caller_matches_python_binding_gpu(hip_lib, tmp_path, scene, n):
    lines = _check(hip_lib, tmp_path, scene, n, 10, 32)
    assert " emulator 0" in lines[0]


@pytest.mark.skipif(not os.path.isdir("/root/reference/args"), reason="reference checkout not present (its arg file is the input)")
def test_c_caller_from_the_references_arg_file(emu_lib, tmp_path):
    """the whole native route: the reference's own arg file -> dm_scene_load (C++ in the library) -> dm_create -> dm_step_batch, in a C99 process; prints what
    the blob route prints for the tables Python builds from the same arg file"""
    import dump_tables
    from deepmimic_amd import model
    exe = str(tmp_path / "smoke"); blob = str(tmp_path / "walk.dmtbl")
    _build(emu_lib, exe)
    dump_tables.dump(model.load_scene_from_args(["--arg_file", "args/run_humanoid3d_walk_args.txt"], data_root="/root/reference"), blob)
    a = subprocess.run([exe, "--args", "args/run_humanoid3d_walk_args.txt", "--data-root", "/root/reference", "3", "2", "64"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    b = subprocess.run([exe, blob, "3", "2", "64"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert a.returncode == 0 and b.returncode == 0, a.stderr + b.stderr
    assert a.stdout == b.stdout and a.stdout.strip().endswith("ok")
