"""One GPU owner for W one-env callers: the worker-coalescing route of the cDeepMimicCore facade (DM_FACADE_SHARED=1, round 4).

The reference's deployment is W processes with one `cDeepMimicCore` each (`mpiexec -n W python3 DeepMimic_Optimizer.py`, mpi_run.py:16-24).  Served
one `dm_ctx` per process, W processes time-slice the GPU and every control step of every worker is a launch of ONE wavefront plus a blocking read-back:
2 539 env-steps/s at W = 16 and 733 at W = 64 in round 3, slower than the host cores.  Here ONE process owns the device -- a context of `max_workers`
envs -- and the workers are thin proxies: a request (reset / control step / query / state get-set) is written into the worker's slot of a POSIX
shared-memory region, the owner collects every request that is pending, runs ALL pending control steps as ONE `dm_step_envs` launch (include/dm_hip.h;
slot = env id), writes each worker's row back and wakes it.  The trajectory of a worker's env is the one it would follow in a context of its own
(one character per wavefront either way; tests/test_broker.py: bit-identical to the per-process facade).

Mechanics: numpy views over `multiprocessing.shared_memory`; per-slot sequence words `req` / `ack` (a request is pending while they differ); Linux
futexes for sleeping and waking -- the owner on one word, all workers on one reply-generation word, one syscall per round -- (no sockets on the data path); x86-TSO store order (arguments first, `req` last; results
first, `ack` last).  The owner process is started by the first worker that finds none (file lock), and leaves when its last worker has detached.

Scope (round 5): every scene kind the context serves -- imitate / imitate_amp, the goal-conditioned task scenes, multi-clip datasets, random perturbations, random
yaw, the dribble ball, DM-physics v1 and v2.  What makes a slot of the shared context draw like a private one-env context is `dm_set_env_keys` (include/dm_hip.h):
at attach the slot gets the worker's seed as its own draw key and is re-initialised the way `dm_create` initialises env 0; the requests of a round are grouped by
(mode, episode-limit parameters), which are per-context settings on the device.  Test infrastructure / learner plumbing only in the sense of SURVEY 8(b): this is
boundary work, the kernels do not know about it (tests/test_broker.py: bit-identical to the per-process facade for one asset of every scene kind).
"""
from __future__ import annotations

import ctypes as C
import fcntl
import hashlib
import os
import pickle
import platform
import stat
import subprocess
import sys
import time
from multiprocessing import shared_memory

import numpy as np

from .core import TAPE_HDR, TAPE_STRIDE

OP_RESET, OP_STEP, OP_QUERY, OP_GET_STATE, OP_SET_STATE, OP_QUERY_AMP, OP_AMP_EXPERT, OP_DETACH, OP_ATTACH = 1, 2, 3, 4, 5, 6, 7, 8, 9
MAGIC = 0x444D4252          # "DMBR"
_libc = C.CDLL(None, use_errno=True)
_SYS_FUTEX, _FUTEX_WAIT, _FUTEX_WAKE = 202, 0, 1          # x86-64


class _Timespec(C.Structure):
    _fields_ = [("tv_sec", C.c_long), ("tv_nsec", C.c_long)]


def futex_wait(addr: int, expected: int, timeout_s: float):
    ts = _Timespec(int(timeout_s), int((timeout_s - int(timeout_s)) * 1e9))
    _libc.syscall(_SYS_FUTEX, C.c_void_p(addr), _FUTEX_WAIT, C.c_int(expected), C.byref(ts), None, 0)


def futex_wake(addr: int, n: int = 1):
    _libc.syscall(_SYS_FUTEX, C.c_void_p(addr), _FUTEX_WAKE, C.c_int(n), None, None, 0)


def _open_private(path: str, mode: str):
    """Open (creating) a file of OURS under the world-writable /dev/shm: never through a symlink, never one another user planted under our name."""
    flags = os.O_CREAT | os.O_NOFOLLOW | os.O_CLOEXEC | (os.O_RDWR if "+" in mode else os.O_WRONLY) | (os.O_APPEND if "a" in mode else 0)
    fd = os.open(path, flags, 0o600)
    st = os.fstat(fd)
    if st.st_uid != os.getuid() or not stat.S_ISREG(st.st_mode):
        os.close(fd)
        raise RuntimeError("DM_FACADE_SHARED: %s exists and is not a regular file of this user; refusing to use it (set DM_FACADE_SHM to another name)" % path)
    return os.fdopen(fd, mode)


def _check_region_owner(name: str):
    st = os.stat("/dev/shm/%s" % name)
    if st.st_uid != os.getuid():
        raise RuntimeError("DM_FACADE_SHARED: shared-memory region %s belongs to uid %d, not to this user; refusing to attach" % (name, st.st_uid))


def region_name(tables, precision: int, device: int, physics: int = 1) -> str:
    """One region per (user, scene tables, precision, device, physics): unrelated runs never meet; DM_FACADE_SHM overrides."""
    if os.environ.get("DM_FACADE_SHM"):
        return os.environ["DM_FACADE_SHM"]
    h = hashlib.sha1(pickle.dumps(tables, protocol=4)).hexdigest()[:12]
    return "dmshare_%d_%s_%d_%d_%d" % (os.getuid(), h, precision, device, physics)


class Region:
    """The shared region: header + per-slot arrays, laid out from (W, S, A, P, J, AMP, G, feature bits: 1 goal row, 2 free body, 4 perturbations, 8 manifolds, NC clips)."""
    HDR = 64          # int64 words

    def __init__(self, name: str, create=False, dims=None):
        if create:
            W, S, A, P, J, AMP, G, FB, NC = dims
            size = self._layout(W, S, A, P, J, AMP, G, FB, NC)
            self.shm = shared_memory.SharedMemory(name=name, create=True, size=size)
            self.hdr = np.ndarray((self.HDR,), dtype=np.int64, buffer=self.shm.buf, offset=0)
            self.hdr[:] = 0
            self.hdr[1:7] = (W, S, A, P, J, AMP); self.hdr[11:14] = (G, FB, NC)
        else:
            _check_region_owner(name)                       # (FileNotFoundError when there is none, like SharedMemory itself)
            self.shm = shared_memory.SharedMemory(name=name)
            # Python 3.10 registers an ATTACHED segment with the process's resource tracker, which unlinks it when this process exits (bpo-38119):
            # a worker that leaves would take the region away from under the others.  The owner alone unlinks it.
            try:
                from multiprocessing import resource_tracker
                resource_tracker.unregister(self.shm._name, "shared_memory")
            except Exception:
                pass
            self.hdr = np.ndarray((self.HDR,), dtype=np.int64, buffer=self.shm.buf, offset=0)
            W, S, A, P, J, AMP = [int(x) for x in self.hdr[1:7]]; G, FB, NC = [int(x) for x in self.hdr[11:14]]
            self._layout(W, S, A, P, J, AMP, G, FB, NC)
        self.W, self.S, self.A, self.P, self.J, self.AMP, self.G, self.FB, self.NC = W, S, A, P, J, AMP, G, FB, NC
        buf = self.shm.buf
        for nm, (off, dt, shape) in self._fields.items():
            setattr(self, nm, np.ndarray(shape, dtype=dt, buffer=buf, offset=off))
        self._base = C.addressof(C.c_char.from_buffer(buf))

    def _layout(self, W, S, A, P, J, AMP, G=0, FB=0, NC=1):
        self._fields = {}
        off = self.HDR * 8

        def add(nm, dt, shape):
            nonlocal off
            off = (off + 63) // 64 * 64
            self._fields[nm] = (off, dt, shape)
            off += int(np.prod(shape)) * np.dtype(dt).itemsize
        add("wake", np.int32, (32,))                       # [0]: the owner's futex word; [1]: the reply generation all workers sleep on
        add("owner", np.int32, (W,))                       # pid of the worker that holds the slot (0 = free)
        add("req", np.int32, (W,)); add("ack", np.int32, (W,)); add("op", np.int32, (W,)); add("status", np.int32, (W,))
        add("iargs", np.int32, (W, 8)); add("dargs", np.float64, (W, 8))
        add("action", np.float32, (W, max(A, 1))); add("state", np.float32, (W, S)); add("reward", np.float32, (W,)); add("flags", np.int32, (W, 4))
        add("clocks", np.float64, (W, 5)); add("amp", np.float32, (W, max(AMP, 1)))
        # get / set state: pose, vel, tar, kin, clocks, flags [| goal state 12, goal aux 8, clip 1][| free body 13][| perturbation row 16][| manifolds J x 25]
        self.off_goal = 3 * P + 16; self.off_obj = self.off_goal + (21 if FB & 1 else 0); self.off_pert = self.off_obj + (13 if FB & 2 else 0)
        self.off_manif = self.off_pert + (16 if FB & 4 else 0); self.big_w = self.off_manif + (J * 25 if FB & 8 else 0)
        add("big", np.float64, (W, self.big_w))
        add("goal", np.float32, (W, max(G, 1)))               # RecordGoal of the last step / query
        add("tape", np.float64, (W, TAPE_STRIDE))             # the worker's draw tape (DM_RNG=reference: include/dm_hip.h DM_TAPE_*); the owner writes the header back after the launch
        add("meta", np.float64, (4 * S + 4 * A + 8 + max(AMP, 1) * 3 + 2 * max(NC, 1),))      # offsets / scales / bounds / norm groups, duration, ..., clip durations | cdf
        return off

    def addr(self, nm: str, i: int) -> int:
        off, dt, _ = self._fields[nm]
        return self._base + off + i * np.dtype(dt).itemsize

    def close(self, unlink=False):
        for nm in list(self._fields) + ["hdr"]:
            if hasattr(self, nm):
                delattr(self, nm)
        try:
            self.shm.close()
            if unlink:
                self.shm.unlink()
        except Exception:
            pass


# ======================================================================================================================= owner side
def serve(name: str, max_workers: int, device: int, precision: int, lib_path: str, physics: int = 1, idle_exit_s: float = 3.0):
    """The owner loop.  The scene tables arrive pickled on stdin, from the worker that started this process (a pipe: nothing another user
    could have planted under /dev/shm is ever unpickled)."""
    from deepmimic_amd.core import BatchEnv
    tables = pickle.load(sys.stdin.buffer)
    env = BatchEnv(tables, max_workers, device_id=device, seed=0, precision=precision, lib_path=lib_path or None, wave_packing=1, physics=physics)
    FB = (1 if env._has_goal_row else 0) | (2 if env.has_obj else 0) | (4 if env.has_perturbs else 0) | (8 if env.physics == 2 else 0)
    NC = max(1, int(env.num_clips))
    R = Region(name, create=True, dims=(max_workers, env.S, env.A, env.P, env.J, env.amp_size, env.G, FB, NC))
    off = env.offsets_scales()
    S, A = env.S, env.A
    m = R.meta
    m[0:S] = off["state_offset"]; m[S:2 * S] = off["state_scale"]; m[2 * S:3 * S] = off["state_norm_groups"]
    b = 3 * S
    for k in ("action_offset", "action_scale", "action_min", "action_max"):
        m[b:b + A] = off[k]; b += A
    m[b] = env.duration; m[b + 1] = env.D; m[b + 2] = env.max_contacts; m[b + 3] = env.physics
    cd, cc = env.clip_table()
    b2 = 4 * S + 4 * A + 8 + max(env.amp_size, 1) * 3
    m[b2:b2 + NC] = cd; m[b2 + NC:b2 + 2 * NC] = cc
    cur_cfg = [None]
    tape_bound = [False]

    def configure(mode, lo, hi, ex):
        """mode and episode-limit parameters are per-CONTEXT settings on the device: the requests of a round are served in groups that agree on them"""
        key = (int(mode), float(lo), float(hi), float(ex))
        if cur_cfg[0] != key:
            env.set_time_limits(key[1], key[2], key[3])
            if env.G:
                env.set_mode(bool(key[0]))
            cur_cfg[0] = key
    R.hdr[8] = os.getpid()
    R.hdr[0] = MAGIC                                        # ready
    W = max_workers
    had_worker, idle_since = False, time.monotonic()
    gather_max = float(os.environ.get("DM_BROKER_GATHER_US", "3000")) * 1e-6        # longest wait for the stragglers of a control step
    gather_quiet = float(os.environ.get("DM_BROKER_QUIET_US", "250")) * 1e-6      # ... or this long without a new arrival
    gen_addr = R.addr("wake", 1)
    wake_addr = R.addr("wake", 0)
    gone = False
    last_sweep = time.monotonic()
    stats = {"launches": 0, "steps": 0, "rounds": 0, "t_gather": 0.0, "t_step": 0.0, "t_other": 0.0, "t_idle": 0.0, "t_call": 0.0, "t_pre": 0.0, "other_ops": 0}
    t_mark = time.perf_counter()
    try:
        while True:
            pend = np.nonzero((R.req != R.ack) & (R.owner != 0))[0]
            if pend.size == 0:
                R.wake[0] = 0
                pend = np.nonzero((R.req != R.ack) & (R.owner != 0))[0]
                if pend.size == 0:
                    n_att = int(np.count_nonzero(R.owner))
                    if n_att:
                        had_worker, idle_since = True, time.monotonic()
                        # a worker that died without detaching frees its slot
                        for i in np.nonzero(R.owner)[0]:
                            if not os.path.exists("/proc/%d" % int(R.owner[i])):
                                R.owner[i] = 0
                    elif (had_worker and time.monotonic() - idle_since > idle_exit_s) or (not had_worker and time.monotonic() - idle_since > 120.0):
                        # leave -- decided and carried out under the region's file lock, the one workers hold while they look for an owner and claim a slot: nobody
                        # can attach to a region that is about to go, and no late unlink can remove the name of a successor's region
                        with _open_private("/dev/shm/%s.lock" % name, "a+") as lk:
                            fcntl.flock(lk, fcntl.LOCK_EX)
                            if np.count_nonzero(R.owner) == 0:
                                R.hdr[0] = 0
                                R.hdr[9] = stats["launches"]; R.hdr[10] = stats["steps"]
                                R.close(unlink=True)
                                gone = True
                        if gone:
                            break
                        had_worker, idle_since = True, time.monotonic()      # a worker claimed a slot while the lock was awaited: stay
                        continue
                    futex_wait(wake_addr, 0, 0.05)
                    continue
            # gather window: the workers of a control step were woken together and come back in a burst.  A launch costs ~1.5 ms whatever it carries, so
            # the owner waits for the burst: until every attached worker has a request pending, or nothing new has arrived for `quiet`, or `max_wait`.
            if time.monotonic() - last_sweep > 1.0:          # a worker that died without detaching must not hold every later round to the full gather window
                last_sweep = time.monotonic()
                for i in np.nonzero(R.owner)[0]:
                    if not os.path.exists("/proc/%d" % int(R.owner[i])):
                        R.owner[i] = 0
                pend = np.nonzero((R.req != R.ack) & (R.owner != 0))[0]
                if pend.size == 0:
                    continue
            n_att = int(np.count_nonzero(R.owner))
            had_worker, idle_since = True, time.monotonic()          # (a served request is a worker seen: an owner that was never idle while its workers were attached must not wait out the 120 s start-up grace afterwards)
            t_g0 = time.perf_counter(); stats["t_idle"] += t_g0 - t_mark
            if pend.size < n_att and gather_max > 0:
                t0 = t_last = time.perf_counter(); last_n = pend.size
                while True:
                    pend = np.nonzero((R.req != R.ack) & (R.owner != 0))[0]
                    if pend.size >= n_att:
                        break
                    now = time.perf_counter()
                    if pend.size != last_n:
                        last_n, t_last = pend.size, now
                    if now - t_last > gather_quiet or now - t0 > gather_max:
                        break
            t_g1 = time.perf_counter(); stats["t_gather"] += t_g1 - t_g0; stats["rounds"] += 1
            t_s1 = t_g1
            ops = R.op[pend]
            P = env.P

            def read_all():
                s_ = env.get_state()
                if FB & 1:
                    s_["goal"] = env.get_goal_state(); s_["aux"] = env.get_goal_aux(); s_["clip"] = env.get_clips().astype(np.float64)[:, None]
                if FB & 2:
                    s_["obj"] = env.get_obj_state()
                if FB & 4:
                    s_["pert"] = env.get_perturb_state()
                if FB & 8:
                    s_["manif"] = env.get_manifolds().reshape(W, -1)
                return s_

            def pack_state(s_, ids):
                cols = [s_["pose"][ids], s_["vel"][ids], s_["tar"][ids], s_["kin"][ids], s_["clocks"][ids], s_["flags"][ids].astype(np.float64)]
                if FB & 1:
                    cols += [s_["goal"][ids], s_["aux"][ids], s_["clip"][ids]]
                if FB & 2:
                    cols.append(s_["obj"][ids])
                if FB & 4:
                    cols.append(s_["pert"][ids])
                if FB & 8:
                    cols.append(s_["manif"][ids])
                R.big[ids] = np.concatenate(cols, axis=1)

            failed = np.zeros(W, dtype=bool)

            def guarded(ids, fn):
                """one request kind (or one group) of the round: a failure marks ITS slots only -- what the other calls of the round already did to the
                device stands, and their workers get their results"""
                if ids.size == 0:
                    return
                try:
                    fn(ids)
                except Exception as ex_:
                    failed[ids] = True
                    sys.stderr.write("deepmimic_amd.broker: %d request(s) of a round failed: %r\n" % (ids.size, ex_))

            def by_config(ids, extra_cols=()):
                """groups of `ids` that agree on (mode, lo, hi, ex) and on the extra key columns"""
                if ids.size == 0:
                    return []
                km = np.column_stack([R.iargs[ids, 5].astype(np.float64), R.dargs[ids, 2], R.dargs[ids, 3], R.dargs[ids, 4]] + [c for c in extra_cols] + [(R.iargs[ids, 6] != 0).astype(np.float64)])
                uniq, inv = np.unique(km, axis=0, return_inverse=True)
                return [(uniq[g], ids[np.ravel(inv) == g]) for g in range(uniq.shape[0])]

            def bind_tape(ids, on):
                """DM_RNG=reference workers hand a draw tape with the request (last key column of by_config): their rows go up before the launch, the headers come back
                after it; a group without tapes runs with none bound"""
                if on:
                    new = ids[R.iargs[ids, 6] == 2]            # rows re-tabulated since the last launch (2); the others (1) are on the device already, untouched
                    if new.size or not tape_bound[0]:
                        env.set_draw_tape_envs(new.astype(np.int32), R.tape[new]); tape_bound[0] = True
                elif tape_bound[0]:
                    env.set_draw_tape(None); tape_bound[0] = False

            def commit_tape(ids, on):
                if on:
                    R.tape[ids, :TAPE_HDR] = env.draw_tape_state_envs(ids.astype(np.int32))

            # every kind of request of the round is served by ONE call per group for all the slots that made it (a slot has one request pending at a time)
            at = pend[ops == OP_ATTACH]

            def attach(ids):
                # a slot becomes what env 0 of a fresh one-env context with the worker's seed is: own draw key, counters at 0, then dm_create's own first reset
                env.set_env_keys(ids.astype(np.int32), R.dargs[ids, 0].astype(np.uint64))
                bind_tape(ids, False)
                for key, grp in by_config(ids):
                    configure(*key[:4])
                    env.reset(env_ids=grp.astype(np.int32))
            guarded(at, attach)
            rs = pend[ops == OP_RESET]
            for key, grp in by_config(rs, (R.iargs[rs, 0].astype(np.float64),)):
                def reset_group(ids, key=key):
                    configure(*key[:4])
                    bind_tape(ids, int(key[-1]))
                    if int(key[4]):                            # explicit clip time / episode limit (the facade drew them on the host, e.g. in the reference's order)
                        env.reset(env_ids=ids.astype(np.int32), kin_times=R.dargs[ids, 0], max_times=R.dargs[ids, 1])
                    else:                                      # the device draws clip, clip time, yaw and limit under the slot's own key (or off the slot's tape)
                        env.reset(env_ids=ids.astype(np.int32))
                    commit_tape(ids, int(key[-1]))
                guarded(grp, reset_group)
            stp = pend[ops == OP_STEP]
            snap = stp[R.iargs[stp, 4] != 0] if stp.size else stp            # control steps that may be rolled back: the state they start from
            gs, ss = pend[ops == OP_GET_STATE], pend[ops == OP_SET_STATE]
            want = np.concatenate([rs, at, snap, gs]) if (rs.size or at.size or snap.size or gs.size) else rs

            def states(_):
                s_ = read_all()
                if ss.size:
                    v = R.big[ss]
                    s_["pose"][ss] = v[:, :P]; s_["vel"][ss] = v[:, P:2 * P]; s_["tar"][ss] = v[:, 2 * P:3 * P]; s_["kin"][ss] = v[:, 3 * P:3 * P + 7]
                    s_["clocks"][ss] = v[:, 3 * P + 7:3 * P + 12]; s_["flags"][ss] = v[:, 3 * P + 12:3 * P + 16].astype(np.int32)
                    env.set_state(pose=s_["pose"], vel=s_["vel"], tar=s_["tar"], kin=s_["kin"], clocks=s_["clocks"], flags=s_["flags"])
                    if FB & 1:
                        og = R.off_goal
                        s_["goal"][ss] = v[:, og:og + 12]; s_["aux"][ss] = v[:, og + 12:og + 20]; s_["clip"][ss] = v[:, og + 20:og + 21]
                        env.set_goal_state(s_["goal"]); env.set_goal_aux(s_["aux"]); env.set_clips(s_["clip"][:, 0].astype(np.int32))
                    if FB & 2:
                        s_["obj"][ss] = v[:, R.off_obj:R.off_obj + 13]; env.set_obj_state(s_["obj"])
                    if FB & 4:
                        s_["pert"][ss] = v[:, R.off_pert:R.off_pert + 16]; env.set_perturb_state(s_["pert"])
                    if FB & 8:
                        s_["manif"][ss] = v[:, R.off_manif:]; env.set_manifolds(s_["manif"].reshape(W, env.J, 25))      # after set_state, which empties them
                if want.size:
                    pack_state(s_, want)
            guarded(np.concatenate([want, ss]) if (want.size or ss.size) else want, states)
            t_s0 = time.perf_counter(); stats["t_pre"] += t_s0 - t_g1
            stp = stp[~failed[stp]]          # a slot whose rollback snapshot (or state write) failed above is answered with an error and NOT stepped: its device state stays where the worker last saw it
            for key, grp in by_config(stp, (R.dargs[stp, 0], R.iargs[stp, 0].astype(np.float64), R.iargs[stp, 1].astype(np.float64), R.iargs[stp, 2].astype(np.float64), R.iargs[stp, 3].astype(np.float64))):
                def step_group(ids, key=key):
                    configure(*key[:4])
                    dt, n_upd, has_act, end_early, want_amp = float(key[4]), int(key[5]), int(key[6]), int(key[7]), int(key[8])
                    t_c0 = time.perf_counter()
                    bind_tape(ids, int(key[-1]))
                    out = env.step_envs(ids.astype(np.int32), R.action[ids, :A] if has_act else None, dt, n_upd, end_early=bool(end_early), amp=bool(want_amp))
                    commit_tape(ids, int(key[-1]))
                    stats["t_call"] += time.perf_counter() - t_c0
                    R.state[ids] = out["state"]; R.reward[ids] = out["reward"]
                    R.flags[ids, 0] = out["terminate"]; R.flags[ids, 1] = out["valid"]; R.flags[ids, 2] = out["episode_end"]
                    R.clocks[ids] = out["clocks"]
                    if want_amp and env.amp_size:
                        R.amp[ids] = out["amp_obs"]
                    if env.G:
                        R.goal[ids, :env.G] = env.last_goals()[ids, :env.G]
                    stats["launches"] += 1; stats["steps"] += len(ids)
                guarded(grp, step_group)
            t_s1 = time.perf_counter(); stats["t_step"] += t_s1 - t_s0
            qs = pend[(ops == OP_QUERY) | (ops == OP_QUERY_AMP)]
            for key, grp in by_config(qs):
                def queries(ids, key=key):
                    configure(*key[:4])
                    q = env.query()
                    R.state[ids] = q["state"][ids]; R.reward[ids] = q["reward"][ids]
                    R.flags[ids, 0] = q["terminate"][ids]; R.flags[ids, 1] = q["valid"][ids]; R.flags[ids, 2] = q["episode_end"][ids]; R.flags[ids, 3] = q["need_new_action"][ids]
                    if env.G:
                        R.goal[ids, :env.G] = env.query_goal()[ids]
                    qa = ids[R.op[ids] == OP_QUERY_AMP]
                    if qa.size and env.amp_size:
                        R.amp[qa] = env.query_amp()[qa]
                guarded(grp, queries)

            def experts(ids):
                if env.num_clips > 1:
                    R.amp[ids] = env.amp_expert_clips(int(ids.size), R.iargs[ids, 6].copy(), R.dargs[ids, 0].copy(), R.dargs[ids, 1].copy())
                else:
                    R.amp[ids] = env.amp_expert(int(ids.size), R.dargs[ids, 0].copy(), R.dargs[ids, 1].copy())
            guarded(pend[ops == OP_AMP_EXPERT], experts)
            dt_ = pend[ops == OP_DETACH]
            if dt_.size:
                R.owner[dt_] = 0
            R.status[pend] = np.where(failed[pend], -1, 0)
            stats["other_ops"] += int(pend.size - stp.size)
            done = pend
            done = np.array(done, dtype=np.int64)
            R.ack[done] = R.req[done]
            R.wake[1] = (int(R.wake[1]) + 1) & 0x7FFFFFFF      # one generation word for all workers: one syscall wakes the round
            futex_wake(gen_addr, 0x7FFFFFFF)
            t_mark = time.perf_counter(); stats["t_other"] += t_mark - t_s1
    finally:
        if not gone:                             # (an exception: tear down under the same lock)
            try:
                with _open_private("/dev/shm/%s.lock" % name, "a+") as lk:
                    fcntl.flock(lk, fcntl.LOCK_EX)
                    R.hdr[0] = 0
                    R.hdr[9] = stats["launches"]; R.hdr[10] = stats["steps"]
                    R.close(unlink=True)
            except Exception:
                pass
        if os.environ.get("DM_BROKER_STATS"):
            import json
            with open(os.environ["DM_BROKER_STATS"], "a") as f:
                f.write(json.dumps(dict(stats, max_workers=W)) + "\n")
        env.close()
        try:                                     # (the lock file stays: a worker may be blocked on it right now, and a second inode under the same name would let two owners start)
            if os.path.getsize("/dev/shm/%s.log" % name) == 0:
                os.unlink("/dev/shm/%s.log" % name)
        except OSError:
            pass


# ======================================================================================================================= worker side
class SharedEnv:
    """What the cDeepMimicCore facade needs of a `BatchEnv` with one env, served by the owner process through the shared region."""

    def __init__(self, tables, seed: int = 0, device_id: int = 0, precision: int = 32, lib_path=None, max_workers=None, physics: int = 1):
        self.tables = tables
        c = tables.cfg
        if platform.machine() != "x86_64" or not sys.platform.startswith("linux"):
            raise NotImplementedError("DM_FACADE_SHARED: the broker uses Linux futexes by x86-64 syscall number and relies on x86 store order")
        self._seed, self._ep, self._expert_calls = int(seed) & (2 ** 64 - 1), 1, 0          # (a fresh one-env ctx has consumed episode 0 in dm_create's own reset)
        if self._seed >= (1 << 53):
            raise NotImplementedError("DM_FACADE_SHARED: seeds ride in a double of the env's goal row (dm_set_env_keys): below 2^53")
        W = int(max_workers or os.environ.get("DM_FACADE_SHARED_MAX", "256"))
        name = region_name(tables, precision, device_id, physics)
        # open the region (starting an owner if there is none) AND claim a slot under ONE acquisition of the region's file lock: the owner decides to leave under the
        # same lock and only while no slot is claimed, so a region found live here cannot be unlinked between the look and the claim
        self.R, self.slot = self._attach(name, tables, W, device_id, precision, lib_path or os.environ.get("DM_HIP_LIB") or "", physics)
        R = self.R
        self.N, self.S, self.A, self.P, self.J, self.amp_size, self.G = 1, R.S, R.A, R.P, R.J, R.AMP, R.G
        m, S, A = R.meta, R.S, R.A
        b = 3 * S + 4 * A
        self.duration, self.D, self.max_contacts, self.physics = float(m[b]), int(m[b + 1]), int(m[b + 2]), int(m[b + 3])
        b2 = 4 * S + 4 * A + 8 + max(R.AMP, 1) * 3
        self._clip_dur, self._clip_cdf = m[b2:b2 + R.NC].copy(), m[b2 + R.NC:b2 + 2 * R.NC].copy()
        self.num_clips = int(tables.num_clips); self._has_goal_row = bool(R.FB & 1); self.has_obj = bool(R.FB & 2); self.has_perturbs = bool(R.FB & 4)
        self.precision = precision
        self._timer = (c.timer_type, float(c.time_lim_min), float(c.time_lim_max), float(c.time_lim_exp))
        self._mode = 0
        self._tape_on = self._tape_new = False
        self._gen_addr, self._wake_addr = R.addr("wake", 1), R.addr("wake", 0)
        self._state, self._snap = None, None
        # the slot becomes env 0 of a one-env context of this worker's seed (own draw key, counters at 0, dm_create's own first reset)
        R.dargs[self.slot, 0] = float(self._seed)
        self._put_config()
        self._call(OP_ATTACH)
        self._state = self._unpack()

    def _put_config(self, launch=False):
        """mode and episode-limit parameters travel with every request (per-context settings on the device: the owner groups by them).  launch: a reset or a step -- the
        requests the owner binds this worker's draw tape for (rows tabulated since the last such request go up with it)"""
        ty, lo, hi, ex = self._timer
        self.R.iargs[self.slot, 5] = self._mode
        self.R.iargs[self.slot, 6] = (2 if self._tape_new else 1) if self._tape_on else 0
        if launch:
            self._tape_new = False
        self.R.dargs[self.slot, 2] = lo; self.R.dargs[self.slot, 3] = hi; self.R.dargs[self.slot, 4] = ex if ty == "exp" else 0.0

    @staticmethod
    def _attach(name, tables, W, device, precision, lib_path, physics=1, timeout=180.0):
        """Open the region, starting the owner process if there is none (first worker), and claim a slot -- all under one hold of the file lock.  Returns (region, slot)."""
        lock_path = "/dev/shm/%s.lock" % name
        t_end = time.monotonic() + timeout
        with _open_private(lock_path, "a+") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            alive = False
            try:
                R = Region(name)
                pid = int(R.hdr[8])
                alive = int(R.hdr[0]) == MAGIC and pid > 0 and os.path.exists("/proc/%d" % pid)
                if not alive:
                    R.close(unlink=True)
            except FileNotFoundError:
                pass
            if not alive:
                env = dict(os.environ)
                root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
                env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
                with _open_private("/dev/shm/%s.log" % name, "ab") as log:
                    owner = subprocess.Popen([sys.executable, "-m", "deepmimic_amd.broker", "--serve", name, str(W), str(device), str(precision), lib_path, str(physics)],
                                             env=env, stdin=subprocess.PIPE, stdout=log, stderr=log, start_new_session=True, close_fds=True)
                pickle.dump(tables, owner.stdin, protocol=4)          # the scene tables go down a pipe
                owner.stdin.close()
                while True:
                    try:
                        R = Region(name)
                        if int(R.hdr[0]) == MAGIC:
                            break
                        R.close()
                    except (FileNotFoundError, ValueError):
                        pass
                    if time.monotonic() > t_end:
                        raise RuntimeError("DM_FACADE_SHARED: the owner process did not come up (see /dev/shm/%s.log)" % name)
                    time.sleep(0.05)
            # still under the lock: claim a slot of the region that was just seen live
            free = np.nonzero(R.owner == 0)[0]
            if free.size == 0:
                raise RuntimeError("DM_FACADE_SHARED: all %d slots of %s are taken (DM_FACADE_SHARED_MAX)" % (R.W, name))
            slot = int(free[0])
            R.ack[slot] = R.req[slot]
            R.owner[slot] = os.getpid()
        return R, slot

    # ---- one request / reply
    def _call(self, op, timeout=120.0):
        R, i = self.R, self.slot
        R.op[i] = op
        old = int(R.ack[i])
        R.req[i] = (old + 1) & 0x7FFFFFFF
        R.wake[0] = 1
        futex_wake(self._wake_addr, 1)
        t_end = time.monotonic() + timeout
        while True:
            gen = int(R.wake[1])                             # read BEFORE the check: a reply in between changes the word and the wait returns at once
            if int(R.ack[i]) != old:
                break
            futex_wait(self._gen_addr, gen, 0.5)
            if int(R.ack[i]) == old and time.monotonic() > t_end:
                raise RuntimeError("DM_FACADE_SHARED: no reply from the owner process within %.0f s" % timeout)
        if int(R.status[i]) != 0:
            raise RuntimeError("DM_FACADE_SHARED: the owner process failed op %d (see its log in /dev/shm)" % op)

    # ---- the BatchEnv surface the facade uses
    def reset(self, env_ids=None, kin_times=None, max_times=None):
        from . import model, streams
        R, i = self.R, self.slot
        self._put_config(launch=True)
        if kin_times is None and self._has_goal_row:
            # goal scenes / multi-clip datasets / random yaw: the DEVICE draws clip, clip time, yaw and episode limit, under this slot's own key and episode counter
            # -- exactly the draws of a one-env context of its own
            R.iargs[i, 0] = 0
        else:
            if kin_times is None:                             # counter mode: the draws a one-env ctx of its own would make on the device (stream 0 / 1)
                kt = self.duration * streams.reset_rand01(self._seed, 0, self._ep, 0)
            else:
                kt = float(np.ravel(kin_times)[0])
            if max_times is None:
                ty, lo, hi, ex = self._timer
                mt = hi if not hi > lo else model.draw_time_limit(ty, lo, hi, ex, streams.reset_rand01(self._seed, 0, self._ep, 1))
            else:
                mt = float(np.ravel(max_times)[0])
            R.iargs[i, 0] = 1; R.dargs[i, 0] = kt; R.dargs[i, 1] = mt
        self._ep += 1
        self._call(OP_RESET)
        self._state = self._unpack()                          # the owner hands the reset state back with the reply: get_state() right after costs no round trip

    # ---- the draw tape of this worker's generators (DM_RNG=reference): travels with every reset / step request, the header comes back with the reply
    def set_draw_tape(self, tape):
        if tape is None:
            self._tape_on = False
            return
        self.R.tape[self.slot] = np.asarray(tape, dtype=np.float64).reshape(-1)
        self._tape_on = self._tape_new = True

    def draw_tape_state(self):
        return self.R.tape[self.slot:self.slot + 1, :TAPE_HDR].copy()

    def set_clips(self, clips):
        cur = {k: a.copy() for k, a in self._full().items()}
        cur["clip"] = np.asarray(clips, dtype=np.float64).reshape(1, 1)
        self._put_state(cur)

    def clip_table(self):
        return self._clip_dur.copy(), self._clip_cdf.copy()

    def set_time_limits(self, lo, hi, ex=None):
        self._timer = (self._timer[0], float(lo), float(hi), self._timer[3] if ex is None else float(ex))

    def set_mode(self, test_mode):
        self._mode = int(bool(test_mode))

    def _out(self, amp=False):
        R, i = self.R, self.slot
        out = dict(state=R.state[i:i + 1].copy(), reward=R.reward[i:i + 1].copy(), terminate=R.flags[i:i + 1, 0].copy(), valid=R.flags[i:i + 1, 1].copy(),
                   episode_end=R.flags[i:i + 1, 2].copy())
        if amp and self.amp_size:
            out["amp_obs"] = R.amp[i:i + 1].copy()
        if self.G:
            out["goal"] = R.goal[i:i + 1, :self.G].copy()
        return out

    def step(self, actions=None, timestep=1.0 / 600, n_updates=20, auto_reset=False, open_loop=False, amp=False, end_early=None):
        if auto_reset or open_loop:
            raise NotImplementedError("SharedEnv.step: the facade resets explicitly and always hands an action (or none)")
        R, i = self.R, self.slot
        if actions is not None:
            R.action[i, :self.A] = np.asarray(actions, dtype=np.float32).reshape(self.A)
        self._put_config(launch=True)
        R.dargs[i, 0] = float(timestep)
        snap, self._snap = self._snap, None
        R.iargs[i, :5] = (int(n_updates), 0 if actions is None else 1, int(bool(end_early)), int(bool(amp)), 0 if snap is None else 1)
        self._state = None
        self._call(OP_STEP)
        if snap is not None:
            snap.update(self._unpack())                       # the state this step started from, captured by the owner for all workers of the round at once
        out = self._out(amp)
        out["clocks"] = R.clocks[i:i + 1].copy()             # kin_time, ctrl_time, init_time_offset, timer_time, timer_max after the step
        return out

    def query(self):
        self._put_config()
        self._call(OP_QUERY)
        out = self._out()
        out["need_new_action"] = self.R.flags[self.slot:self.slot + 1, 3].copy()
        return out

    def query_amp(self):
        self._put_config()
        self._call(OP_QUERY_AMP)
        return self.R.amp[self.slot:self.slot + 1].copy()

    def amp_expert(self, n, times=None, ground_h=None):
        assert n == 1
        if times is None:            # counter mode: the draw dm_amp_expert makes for a one-env ctx of its own (dm_host.cpp: key (seed, 0x414D50, call, sample 0))
            from . import streams
            times = [self.duration * streams.reset_rand01(self._seed, 0x414D50, self._expert_calls, 0)]
            self._expert_calls += 1
        self.R.iargs[self.slot, 6] = 0
        self.R.dargs[self.slot, 0] = float(np.ravel(times)[0]); self.R.dargs[self.slot, 1] = 0.0 if ground_h is None else float(np.ravel(ground_h)[0])
        self._call(OP_AMP_EXPERT)
        return self.R.amp[self.slot:self.slot + 1].copy()

    def amp_expert_clips(self, n, clips=None, times=None, ground_h=None):
        """one expert sample of a multi-clip dataset; clip and time drawn as dm_amp_expert_clips draws them for a one-env ctx of its own (dm_host.cpp: clip by the
        cdf from key (seed, 0x434C50, call, 0), time = that clip's duration x key (seed, 0x414D50, call, 0))"""
        assert n == 1
        from . import streams
        if clips is None:
            u = streams.reset_rand01(self._seed, 0x434C50, self._expert_calls, 0)
            k = 0
            while k < len(self._clip_cdf) - 1 and not (u < self._clip_cdf[k]):
                k += 1
        else:
            k = int(np.ravel(clips)[0])
        t = float(np.ravel(times)[0]) if times is not None else float(self._clip_dur[k]) * streams.reset_rand01(self._seed, 0x414D50, self._expert_calls, 0)
        if clips is None or times is None:
            self._expert_calls += 1
        self.R.iargs[self.slot, 6] = k
        self.R.dargs[self.slot, 0] = t; self.R.dargs[self.slot, 1] = 0.0 if ground_h is None else float(np.ravel(ground_h)[0])
        self._call(OP_AMP_EXPERT)
        return self.R.amp[self.slot:self.slot + 1].copy()

    def _unpack(self):
        R = self.R
        v, P = R.big[self.slot].copy(), self.P
        out = dict(pose=v[None, :P], vel=v[None, P:2 * P], tar=v[None, 2 * P:3 * P], kin=v[None, 3 * P:3 * P + 7], clocks=v[None, 3 * P + 7:3 * P + 12],
                   flags=v[None, 3 * P + 12:3 * P + 16].astype(np.int32))
        if R.FB & 1:
            og = R.off_goal
            out["goal"] = v[None, og:og + 12]; out["aux"] = v[None, og + 12:og + 20]; out["clip"] = v[None, og + 20:og + 21]
        if R.FB & 2:
            out["obj"] = v[None, R.off_obj:R.off_obj + 13]
        if R.FB & 4:
            out["pert"] = v[None, R.off_pert:R.off_pert + 16]
        if R.FB & 8:
            out["manif"] = v[R.off_manif:].reshape(1, self.J, 25)
        return out

    def _full(self):
        if self._state is None:
            self._call(OP_GET_STATE)
            self._state = self._unpack()
        return self._state

    def get_state(self):
        return {k: a.copy() for k, a in self._full().items() if k in ("pose", "vel", "tar", "kin", "clocks", "flags")}

    def get_goal_state(self):
        return self._full()["goal"].copy()

    def get_goal_aux(self):
        return self._full()["aux"].copy()

    def get_clips(self):
        st = self._full()
        if "clip" not in st:               # a scene without a goal row runs its one clip (BatchEnv.get_clips says 0 there as well)
            return np.zeros(1, np.int32)
        return st["clip"][:, 0].astype(np.int32)

    def get_obj_state(self):
        return self._full()["obj"].copy()

    def get_perturb_state(self):
        return self._full()["pert"].copy()

    def get_manifolds(self):
        return self._full()["manif"].copy()

    def _put_state(self, cur):
        R = self.R
        row = [cur["pose"][0], cur["vel"][0], cur["tar"][0], cur["kin"][0], cur["clocks"][0], np.asarray(cur["flags"][0], dtype=np.float64)]
        if R.FB & 1:
            row += [np.ravel(cur["goal"]), np.ravel(cur["aux"]), np.ravel(cur["clip"])]
        if R.FB & 2:
            row.append(np.ravel(cur["obj"]))
        if R.FB & 4:
            row.append(np.ravel(cur["pert"]))
        if R.FB & 8:
            row.append(np.ravel(cur["manif"]))
        R.big[self.slot] = np.concatenate(row)
        self._state = None
        self._call(OP_SET_STATE)

    def set_state(self, pose=None, vel=None, tar=None, kin=None, clocks=None, flags=None):
        cur = {k: a.copy() for k, a in self._full().items()}
        for k, a in (("pose", pose), ("vel", vel), ("tar", tar), ("kin", kin), ("clocks", clocks), ("flags", flags)):
            if a is not None:
                cur[k] = np.asarray(a, dtype=np.float64).reshape(1, -1)
        if self.R.FB & 8:
            cur["manif"] = np.zeros_like(cur["manif"])        # like dm_set_state: a state set from outside starts without cached contact points
        self._put_state(cur)

    def snapshot(self):
        """the rollback point of the control step that follows: filled in by that step() (the owner reads the state of every stepping worker with one
        copy before the launch), so taking it costs no round trip; a snapshot that is not followed by a step is fetched on first use"""
        self._snap = _LazySnap(self)
        return self._snap

    def restore(self, snap):
        cur = {k: np.asarray(snap[k], dtype=np.float64).copy() for k in ("pose", "vel", "tar", "kin", "clocks", "flags")}
        cur["flags"] = np.asarray(snap["flags"]).astype(np.int32)
        for k in ("goal", "aux", "clip", "obj", "pert", "manif"):
            if k in snap:
                cur[k] = np.asarray(snap[k], dtype=np.float64).copy()
        self._put_state(cur)

    def offsets_scales(self):
        m, S, A = self.R.meta, self.S, self.A
        out = {"state_offset": m[0:S].copy(), "state_scale": m[S:2 * S].copy(), "state_norm_groups": m[2 * S:3 * S].astype(np.int32)}
        b = 3 * S
        for k in ("action_offset", "action_scale", "action_min", "action_max"):
            out[k] = m[b:b + A].copy(); b += A
        return out

    def close(self):
        if getattr(self, "R", None) is not None:
            try:
                self._call(OP_DETACH, timeout=5.0)
            except Exception:
                pass
            self.R.close()
            self.R = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _LazySnap(dict):
    def __init__(self, env):
        super().__init__(); self._env = env

    def __missing__(self, k):                                 # used before the step that would have filled it
        if self._env._snap is self:
            self._env._snap = None
        self.update({kk: a.copy() for kk, a in self._env._full().items()})
        return dict.__getitem__(self, k)


if __name__ == "__main__":
    if len(sys.argv) >= 7 and sys.argv[1] == "--serve":
        serve(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6], int(sys.argv[7]) if len(sys.argv) > 7 else 1)
    else:
        sys.exit("usage: python -m deepmimic_amd.broker --serve <region> <max workers> <device> <precision> <lib path> [physics]   (pickled scene tables on stdin)")
