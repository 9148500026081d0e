"""Deterministic synthetic action streams of the measurement contract (SURVEY.md 8(d) "Synthetic inputs").

A0  all zeros (precedent: the reference's own native driver feeds zeros, DeepMimicCore/Main.cpp:119-120): PD targets are
    identity rotations, the character collapses within about a second -- exercises contact and fall termination.
A1  open-loop mocap tracking: the action encoding of the reference pose at the env's clip time (computed on the device
    by `dm_step_batch(..., DM_OPEN_LOOP)`, or by the caller from a kinematic pose).
A2  A1 + N(0, 0.05^2) per component (the exploration noise of data/agents/ct_agent_humanoid_ppo.txt "ExpParams.Noise"),
    from a counter-based generator so that the stream depends only on (global env id, control step, component):
    Philox4x32-10, key = (0xD33B + env id, 0), counter = (step * A + j, 0, 0, 0), Box-Muller on the first two words.

Host-side numpy only; nothing here touches the device or the oracle.
"""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
_LO = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter [...,4] uint32, key [...,2] uint32 -> [...,4] uint32 (Salmon et al., SC'11; 10 rounds)."""
    c = np.array(counter, dtype=np.uint32)
    k = np.array(key, dtype=np.uint32)
    lead = np.broadcast_shapes(c.shape[:-1], k.shape[:-1])
    c = np.broadcast_to(c, lead + (4,)).copy()
    k0 = np.broadcast_to(k[..., 0], lead).copy()
    k1 = np.broadcast_to(k[..., 1], lead).copy()
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _M0 * c[..., 0].astype(np.uint64)
            p1 = _M1 * c[..., 2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & _LO).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & _LO).astype(np.uint32)
            c = np.stack([hi1 ^ c[..., 1] ^ k0, lo1, hi0 ^ c[..., 3] ^ k1, lo0], axis=-1)
            k0 = k0 + _W0
            k1 = k1 + _W1
    return c


def normal_noise(env_ids, step, num_actions, sigma=0.05):
    """N(0, sigma^2) noise [len(env_ids), num_actions] (float64) for control step `step`; see the module docstring."""
    env_ids = np.asarray(env_ids, dtype=np.int64)
    n = env_ids.shape[0]
    ctr = np.zeros((n, num_actions, 4), dtype=np.uint32)
    ctr[..., 0] = ((int(step) * num_actions + np.arange(num_actions, dtype=np.int64)) & 0xFFFFFFFF)[None, :]
    key = np.zeros((n, num_actions, 2), dtype=np.uint32)
    key[..., 0] = ((0xD33B + env_ids) & 0xFFFFFFFF)[:, None]
    r = philox4x32_10(ctr, key)
    u1 = (r[..., 0].astype(np.float64) + 0.5) * (1.0 / 4294967296.0)          # (0,1)
    u2 = (r[..., 1].astype(np.float64) + 0.5) * (1.0 / 4294967296.0)
    return sigma * np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def reset_phase(env_ids, duration):
    """Deterministic reset clip time of env i: ((i * 2654435761) mod 2^32) / 2^32 * duration (replaces the global-RNG
    draw of scenes/SceneImitate.cpp:494-500)."""
    i = np.asarray(env_ids, dtype=np.uint64)
    return ((i * np.uint64(2654435761)) & _LO).astype(np.float64) / 4294967296.0 * float(duration)


def stream_a0(n, num_actions):
    return np.zeros((n, num_actions), dtype=np.float64)


def stream_a2(base_actions, env_ids, step, sigma=0.05):
    base_actions = np.asarray(base_actions, dtype=np.float64)
    return base_actions + normal_noise(env_ids, step, base_actions.shape[1], sigma)


def _normalize_angle(th):
    """cMathUtil::NormalizeAngle (util/MathUtil.cpp): wrap to [-pi, pi]."""
    th = np.fmod(th, 2 * np.pi)
    if th > np.pi:
        th -= 2 * np.pi
    elif th < -np.pi:
        th += 2 * np.pi
    return th


def quat_to_exp_map(q):
    """cMathUtil::QuaternionToExpMap via QuaternionToAxisAngle (util/MathUtil.cpp:463-481,607-615); q = (w,x,y,z)."""
    w, x, y, z = [float(c) for c in q]
    if w > 1:
        n = np.sqrt(w * w + x * x + y * y + z * z)
        w, x, y, z = w / n, x / n, y / n, z / n
    st = np.sqrt(max(0.0, 1 - w * w))
    if st > 0.000001:
        th = _normalize_angle(2 * np.arccos(w))
        return np.array([x, y, z]) / st * th
    return np.zeros(3)


def pose_to_action(tables, pose):
    """The action whose PD targets are `pose` (stream A1): spherical joint -> exp map of its quaternion, revolute ->
    angle, fixed -> nothing (action layout of sim/CtCtrlUtil.cpp:10-35: joints in order, root excluded)."""
    from . import model
    pose = np.asarray(pose, dtype=np.float64)
    out = []
    for j in range(1, tables.num_joints):
        off, ty = int(tables.joint_mat[j, model.JD_PARAM_OFFSET]), int(tables.joint_mat[j, model.JD_TYPE])
        if ty == model.JT_SPHERICAL:
            out += list(quat_to_exp_map(pose[off:off + 4]))
        elif ty == model.JT_REVOLUTE:
            out.append(pose[off])
    return np.array(out)


_MASK64 = (1 << 64) - 1


def reset_rand01(seed: int, env: int, episode: int, stream: int) -> float:
    """Host mirror of the device's reset generator `dm_rand01` (dm_device.h): splitmix64 of (seed, global env id, episode
    counter, stream) -> uniform in [0, 1).  stream 0 draws the clip time of a reset (cSceneImitate::CalcRandKinResetTime,
    scenes/SceneImitate.cpp:494-500, which the reference draws from its process-global RNG), stream 1 the episode-timer limit.
    Lets a caller (or the oracle in the parity tests) reproduce every auto-reset of a batched rollout."""
    z = (seed + 0x9E3779B97F4A7C15 * ((env * 0x100000001B3 + episode * 0xD6E8FEB86659FD93 + stream + 1) & _MASK64)) & _MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK64
    z = z ^ (z >> 31)
    return float(z >> 11) * (1.0 / 9007199254740992.0)
