"""Running observation statistics on the device: the reference learner's `Normalizer` (learning/normalizer.py) over records that stay in HBM
(libdm_hip.so `dm_norm_*`, deepmimic_amd/csrc/dm_norm.h).  Same method names and meaning as the reference class --
record / update / set_mean_std / normalize, attributes mean / std / mean_sq / count -- so the learner-side code reads the same; `record_device`
takes the raw device pointer of a control step's record block (e.g. `TorchVecEnv.obs.data_ptr()`), `bind_policy` makes the statistics the
observation normaliser of a `deepmimic_amd.policy.Policy`.  Multi-worker: the pending sums are one device array (`pending_ptr`) to all-reduce
(SUM) over the ranks before `update()`, which is what `MPIUtil.reduce_sum` does in the reference (normalizer.py:48-50)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np

from .core import DM_DEVICE_PTRS, load_library

NORM_GROUP_SINGLE, NORM_GROUP_NONE = 0, -1          # learning/normalizer.py:10-11 (= sim/CharController.h)


class DeviceNormalizer:
    def __init__(self, size: int, groups_ids=None, eps: float = 0.02, clip: float = np.inf, device_id: int = 0, lib_path: Optional[str] = None):
        self.lib = load_library(lib_path)
        L = self.lib
        L.dm_norm_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_double, C.c_void_p]
        L.dm_norm_record.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.dm_norm_pending.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.dm_norm_update.argtypes = [C.c_void_p, C.c_void_p]
        L.dm_norm_set.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
        L.dm_norm_get.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.dm_norm_normalize.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.dm_norm_destroy.argtypes = [C.c_void_p]
        L.dm_policy_bind_obs_normalizer.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        self.size, self.eps, self.clip = int(size), float(eps), float(clip)
        g = None if groups_ids is None else np.ascontiguousarray(groups_ids, dtype=np.int32).ravel()
        if g is not None and g.size != self.size:
            raise ValueError("groups_ids must have `size` entries")
        self.h = C.c_void_p()
        self._chk(L.dm_norm_create(int(device_id), self.size, None if g is None else g.ctypes.data, self.eps, self.clip if np.isfinite(self.clip) else 0.0, C.byref(self.h)))
        self.stream = 0

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("libdm_hip: %s" % self.lib.dm_last_error().decode())

    def set_stream(self, handle: int):
        """hipStream_t handle the calls are ordered on (0 = the null stream)"""
        self.stream = int(handle)

    # ---- the reference's interface
    def record(self, x):
        """host rows (learning/normalizer.py:33-45); fp32 like the device records"""
        x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, self.size)
        self._chk(self.lib.dm_norm_record(self.h, x.ctypes.data, int(x.shape[0]), 0, C.c_void_p(self.stream)))

    def record_device(self, ptr: int, n: int):
        """n rows of `size` fp32 at device address `ptr` (a control step's observations), asynchronous on the stream"""
        self._chk(self.lib.dm_norm_record(self.h, C.c_void_p(int(ptr)), int(n), DM_DEVICE_PTRS, C.c_void_p(self.stream)))

    def pending_ptr(self):
        """(device address, length) of {new_count, new_sum, new_sum_sq} as float64: all-reduce it over the workers before update()"""
        p, n = C.c_void_p(), C.c_int()
        self._chk(self.lib.dm_norm_pending(self.h, C.byref(p), C.byref(n)))
        return int(p.value), int(n.value)

    def update(self):
        self._chk(self.lib.dm_norm_update(self.h, C.c_void_p(self.stream)))

    def set_mean_std(self, mean, std, count: int = -1):
        m = np.ascontiguousarray(mean, dtype=np.float64).ravel(); s = np.ascontiguousarray(std, dtype=np.float64).ravel()
        if m.size != self.size or s.size != self.size:
            raise ValueError("Normalizer shape mismatch, expecting size %d, but got %d and %d" % (self.size, m.size, s.size))
        self._chk(self.lib.dm_norm_set(self.h, m.ctypes.data, s.ctypes.data, int(count), C.c_void_p(self.stream)))

    def _get(self):
        m, s, q = np.zeros(self.size), np.zeros(self.size), np.zeros(self.size); c = C.c_int64()
        self._chk(self.lib.dm_norm_get(self.h, m.ctypes.data, s.ctypes.data, q.ctypes.data, C.byref(c), C.c_void_p(self.stream)))
        return m, s, q, int(c.value)

    mean = property(lambda self: self._get()[0])
    std = property(lambda self: self._get()[1])
    mean_sq = property(lambda self: self._get()[2])
    count = property(lambda self: self._get()[3])

    def get_size(self):
        return self.size

    def normalize_device(self, x_ptr: int, n: int, out_ptr: int):
        self._chk(self.lib.dm_norm_normalize(self.h, C.c_void_p(int(x_ptr)), int(n), C.c_void_p(int(out_ptr)), C.c_void_p(self.stream)))

    def bind_policy(self, policy, first_column: int = 0):
        """columns [first_column, first_column + size) of the policy's observation normaliser := these statistics (device-to-device, ordered on the
        stream); s_norm at 0, g_norm at the state size"""
        self._chk(self.lib.dm_policy_bind_obs_normalizer(policy.h, self.h, int(first_column), C.c_void_p(self.stream)))

    def close(self):
        if getattr(self, "h", None):
            self.lib.dm_norm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
