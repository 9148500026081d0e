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
