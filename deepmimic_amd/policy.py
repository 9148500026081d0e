"""On-device policy inference: ctypes binding of `dm_policy_*` (include/dm_hip.h).

The actor of the reference's PPO / PG agents (learning/pg_agent.py:141-188, learning/nets/fc_2layers_1024units.py,
learning/normalizer.py) evaluated by hand-written MFMA kernels (deepmimic_amd/csrc/dm_policy.h) on device buffers, so a
rollout loop `states -> actions -> BatchEnv.step_device` never touches the host.  No CPU fallback.
"""
import ctypes as C
from typing import Optional

import numpy as np

from .core import load_library


class _PolicyParams(C.Structure):
    _fields_ = [("state_dim", C.c_int), ("hidden1", C.c_int), ("hidden2", C.c_int), ("action_dim", C.c_int)] + \
               [(k, C.POINTER(C.c_float)) for k in ("w1", "b1", "w2", "b2", "w3", "b3", "s_mean", "s_std", "a_mean", "a_std", "logstd")] + \
               [("s_clip", C.c_double)]


def _fp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_float))


class Policy:
    """weights: dict with w1 [S,H1], b1 [H1], w2 [H1,H2], b2 [H2], w3 [H2,A], b3 [A] (tf.layers.dense layout) and optional
    s_mean, s_std, a_mean, a_std, logstd."""

    def __init__(self, weights: dict, device_id: int = 0, s_clip: float = 0.0, lib_path: Optional[str] = None):
        self.lib = load_library(lib_path)
        self.lib.dm_policy_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64,
                                               C.c_uint32, C.c_int, C.c_void_p]
        w = {k: (None if weights.get(k) is None else np.ascontiguousarray(weights[k], dtype=np.float32)) for k in
             ("w1", "b1", "w2", "b2", "w3", "b3", "s_mean", "s_std", "a_mean", "a_std", "logstd")}
        self.S, self.H1 = w["w1"].shape
        self.H2, self.A = w["w3"].shape
        if w["w2"].shape != (self.H1, self.H2) or w["b1"].shape != (self.H1,) or w["b2"].shape != (self.H2,) or w["b3"].shape != (self.A,):
            raise ValueError("inconsistent layer shapes")
        pp = _PolicyParams(self.S, self.H1, self.H2, self.A, *[_fp(w[k]) for k in ("w1", "b1", "w2", "b2", "w3", "b3", "s_mean", "s_std", "a_mean", "a_std", "logstd")],
                           float(s_clip))
        self.h = C.c_void_p()
        if self.lib.dm_policy_create(int(device_id), C.byref(pp), C.byref(self.h)) != 0:
            raise RuntimeError("libdm_hip: %s" % self.lib.dm_last_error().decode())

    @classmethod
    def from_checkpoint(cls, prefix: str, state_dim: Optional[int] = None, **kw):
        """the actor of a reference checkpoint (`--model_files <prefix>`: learning/rl_world.py:67-85, learning/tf_agent.py:36-48; read without TensorFlow by
        deepmimic_amd/tf_checkpoint.py) with its state / action normalisers.  An agent with a goal takes [state, goal] rows (forward_device_ex's goal block);
        its g_norm rides behind s_norm."""
        from . import tf_checkpoint
        w = tf_checkpoint.actor_weights(prefix, state_dim=state_dim)
        if "g_mean" in w:
            w["s_mean"] = np.concatenate([w["s_mean"], w["g_mean"]]); w["s_std"] = np.concatenate([w["s_std"], w["g_std"]])
        return cls(w, **kw)

    def forward_device(self, states_ptr: int, n: int, actions_ptr: int, logp_ptr: int = 0, sample: bool = False, seed: int = 0,
                       step: int = 0, env_id_offset: int = 0, stream: int = 0):
        """raw device pointers (e.g. torch.Tensor.data_ptr()); asynchronous on `stream` (a hipStream_t handle, 0 = null stream)."""
        vp = lambda p: C.c_void_p(p) if p else None
        if self.lib.dm_policy_forward(self.h, vp(states_ptr), int(n), vp(actions_ptr), vp(logp_ptr), int(bool(sample)),
                                      C.c_uint64(int(seed) & (2 ** 64 - 1)), C.c_uint32(int(step) & 0xFFFFFFFF), int(env_id_offset), vp(stream)) != 0:
            raise RuntimeError("libdm_hip: %s" % self.lib.dm_last_error().decode())

    def forward_device_ex(self, states_ptr: int, n: int, actions_ptr: int, goals_ptr: int = 0, goal_dim: int = 0, logp_ptr: int = 0, exp_flags_ptr: int = 0,
                          exp_rate: float = 1.0, sample: bool = False, seed: int = 0, step: int = 0, env_id_offset: int = 0, stream: int = 0):
        """`_decide_action` of learning/pg_agent.py:214-221 for a batch (include/dm_hip.h dm_policy_forward_ex): goal block as its own input,
        per-row exploration coin with probability `exp_rate`, EXP flags out"""
        vp = lambda p: C.c_void_p(p) if p else None
        self.lib.dm_policy_forward_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int,
                                                  C.c_uint64, C.c_uint32, C.c_int, C.c_void_p]
        if self.lib.dm_policy_forward_ex(self.h, vp(states_ptr), vp(goals_ptr), int(goal_dim), int(n), vp(actions_ptr), vp(logp_ptr), vp(exp_flags_ptr),
                                         C.c_double(exp_rate), int(bool(sample)), C.c_uint64(int(seed) & (2 ** 64 - 1)), C.c_uint32(int(step) & 0xFFFFFFFF),
                                         int(env_id_offset), vp(stream)) != 0:
            raise RuntimeError("libdm_hip: %s" % self.lib.dm_last_error().decode())

    def forward_host_ex(self, states, goals=None, exp_rate=1.0, sample=False, seed=0, step=0, env_id_offset=0):
        """emulator-build convenience for forward_device_ex: returns (actions, logp, exp_flags)"""
        s = np.ascontiguousarray(states, dtype=np.float32); n = s.shape[0]
        g = None if goals is None else np.ascontiguousarray(goals, dtype=np.float32)
        a = np.zeros((n, self.A), np.float32); lp = np.zeros(n, np.float32); fl = np.zeros(n, np.int32)
        self.forward_device_ex(s.ctypes.data, n, a.ctypes.data, 0 if g is None else g.ctypes.data, 0 if g is None else g.shape[1], lp.ctypes.data, fl.ctypes.data,
                               exp_rate, sample, seed, step, env_id_offset)
        return a, lp, fl

    def forward_host(self, states, sample=False, seed=0, step=0, env_id_offset=0):
        """Convenience for tests on the CPU emulator build, where "device" memory is host memory."""
        s = np.ascontiguousarray(states, dtype=np.float32)
        n = s.shape[0]
        a = np.zeros((n, self.A), np.float32); lp = np.zeros(n, np.float32)
        self.forward_device(s.ctypes.data, n, a.ctypes.data, lp.ctypes.data, sample, seed, step, env_id_offset)
        return a, lp

    def close(self):
        if getattr(self, "h", None):
            self.lib.dm_policy_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def reference_forward(weights: dict, states, s_clip=np.inf, bf16=False):
    """Plain numpy statement of the same actor (fp32; bf16=True rounds operands to bfloat16 at the points the kernels do).
    Returns the mode action and the normalised mean."""
    def r(x):
        if not bf16:
            return x.astype(np.float32)
        u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.astype(np.uint32).view(np.float32)
    S = weights["w1"].shape[0]
    s = np.asarray(states, dtype=np.float32)
    sm = weights.get("s_mean"); ss = weights.get("s_std")
    x = (s - (0 if sm is None else sm.astype(np.float32))) * (np.float32(1) / (np.float32(1) if ss is None else ss.astype(np.float32)))
    x = np.clip(x, -s_clip, s_clip)
    h = np.maximum(r(x).astype(np.float64) @ r(weights["w1"]).astype(np.float64) + weights["b1"], 0).astype(np.float32)
    h = np.maximum(r(h).astype(np.float64) @ r(weights["w2"]).astype(np.float64) + weights["b2"], 0).astype(np.float32)
    m = (r(h).astype(np.float64) @ r(weights["w3"]).astype(np.float64) + weights["b3"]).astype(np.float32)
    am = weights.get("a_mean"); as_ = weights.get("a_std")
    a = m * (1 if as_ is None else as_) + (0 if am is None else am)
    return a.astype(np.float32), m


def random_weights(S, A, H1=1024, H2=512, seed=0, init_output_scale=0.01, noise=0.05):
    """Random-init weights of the reference architecture: Xavier-uniform hidden layers (learning/tf_util.py:27-39), uniform
    (+-init_output_scale) output layer, logstd = log(noise) (pg_agent.py:147-158)."""
    rng = np.random.default_rng(seed)
    def xav(i, o):
        lim = np.sqrt(6.0 / (i + o)); return rng.uniform(-lim, lim, size=(i, o)).astype(np.float32)
    return dict(w1=xav(S, H1), b1=np.zeros(H1, np.float32), w2=xav(H1, H2), b2=np.zeros(H2, np.float32),
                w3=rng.uniform(-init_output_scale, init_output_scale, size=(H2, A)).astype(np.float32), b3=np.zeros(A, np.float32),
                logstd=np.full(A, np.log(noise), np.float32))
