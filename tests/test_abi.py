"""The C-ABI library loads and exports every symbol include/dm_hip.h declares (no compute calls: runs without a GPU)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "dm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dm_[a-z_]+)\s*\(", src)))


def test_header_declares_the_boundary():
    names = declared_functions()
    for must in ("dm_create", "dm_destroy", "dm_reset", "dm_set_action", "dm_update", "dm_query", "dm_step_batch",
                 "dm_build_offsets_scales", "dm_set_time_limits", "dm_last_error"):
        assert must in names


def test_hip_library_exports_every_declared_symbol(hip_lib):
    lib = ctypes.CDLL(hip_lib)
    for name in declared_functions():
        assert hasattr(lib, name), "libdm_hip.so does not export %s" % name


def test_create_without_gpu_fails_loudly(hip_lib):
    """No CPU fallback: on a box without a HIP device dm_create must return an error, not compute on the host."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is visible")
    from deepmimic_amd import model
    from deepmimic_amd.core import BatchEnv
    try:
        BatchEnv(model.load_asset("humanoid3d_walk"), 1, lib_path=hip_lib)
    except RuntimeError as ex:
        assert "no HIP device" in str(ex) or "hip" in str(ex).lower()
    else:
        raise AssertionError("BatchEnv was created without a GPU")


def test_binding_mirrors_the_header_structs(emu_lib, tmp_path):
    """the ctypes structs of deepmimic_amd/core.py are written by hand: their sizes and a field-offset sample must be what a C compiler makes of
    include/dm_hip.h, and binding, header and library must agree on DM_ABI_VERSION"""
    import ctypes as C
    import re
    import subprocess
    from deepmimic_amd import core
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "dm_hip.h"\nint main(void){ printf("%d %zu %zu %zu %zu %zu\\n", DM_ABI_VERSION, '
                   'sizeof(dm_create_info), sizeof(dm_scene_tables), offsetof(dm_scene_tables, scene_goal), offsetof(dm_scene_tables, ball_radius), '
                   'offsetof(dm_scene_tables, perturb_part_mask)); return 0; }\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    ver, s_info, s_tab, o_goal, o_ball, o_mask = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert ver == core.ABI_VERSION == int(re.search(r"#define DM_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "dm_hip.h")).read()).group(1))
    assert (s_info, s_tab) == (C.sizeof(core._CreateInfo), C.sizeof(core._SceneTables))
    assert (o_goal, o_ball, o_mask) == (core._SceneTables.scene_goal.offset, core._SceneTables.ball_radius.offset, core._SceneTables.perturb_part_mask.offset)
    lib = core.load_library(emu_lib)
    sizes = (C.c_int32 * 2)()
    assert lib.dm_abi_version() == ver and lib.dm_struct_sizes(sizes) == 0 and (sizes[0], sizes[1]) == (s_info, s_tab)
