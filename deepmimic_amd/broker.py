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

Scope: scenes whose episode draws the facade makes on the host -- `imitate` / `imitate_amp`, one clip, no random perturbations, DM-physics v1; for the
others the facade falls back to a context of its own.  Test infrastructure / learner plumbing only in the sense of SURVEY 8(b): this is boundary work,
the kernels do not know about it.
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

OP_RESET, OP_STEP, OP_QUERY, OP_GET_STATE, OP_SET_STATE, OP_QUERY_AMP, OP_AMP_EXPERT, OP_DETACH = 1, 2, 3, 4, 5, 6, 7, 8
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


def region_name(tables, precision: int, device: int) -> str:
    """One region per (user, scene tables, precision, device): unrelated runs never meet; DM_FACADE_SHM overrides."""
    if os.environ.get("DM_FACADE_SHM"):
        return os.environ["DM_FACADE_SHM"]
    h = hashlib.sha1(pickle.dumps(tables, protocol=4)).hexdigest()[:12]
    return "dmshare_%d_%s_%d_%d" % (os.getuid(), h, precision, device)


class Region:
    """The shared region: header + per-slot arrays, laid out from (W, S, A, P, J, AMP)."""
    HDR = 64          # int64 words

    def __init__(self, name: str, create=False, dims=None):
        if create:
            W, S, A, P, J, AMP = dims
            size = self._layout(W, S, A, P, J, AMP)
            self.shm = shared_memory.SharedMemory(name=name, create=True, size=size)
            self.hdr = np.ndarray((self.HDR,), dtype=np.int64, buffer=self.shm.buf, offset=0)
            self.hdr[:] = 0
            self.hdr[1:7] = (W, S, A, P, J, AMP)
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
            W, S, A, P, J, AMP = [int(x) for x in self.hdr[1:7]]
            self._layout(W, S, A, P, J, AMP)
        self.W, self.S, self.A, self.P, self.J, self.AMP = W, S, A, P, J, AMP
        buf = self.shm.buf
        for nm, (off, dt, shape) in self._fields.items():
            setattr(self, nm, np.ndarray(shape, dtype=dt, buffer=buf, offset=off))
        self._base = C.addressof(C.c_char.from_buffer(buf))

    def _layout(self, W, S, A, P, J, AMP):
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
        add("big", np.float64, (W, 3 * P + 7 + 5 + 4))      # get / set state: pose, vel, tar, kin, clocks, flags
        add("meta", np.float64, (4 * S + 4 * A + 8 + max(AMP, 1) * 3,))      # offsets / scales / bounds / norm groups, duration, ...
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
def serve(name: str, max_workers: int, device: int, precision: int, lib_path: str, idle_exit_s: float = 3.0):
    """The owner loop.  The scene tables arrive pickled on stdin, from the worker that started this process (a pipe: nothing another user
    could have planted under /dev/shm is ever unpickled)."""
    from deepmimic_amd.core import BatchEnv
    tables = pickle.load(sys.stdin.buffer)
    env = BatchEnv(tables, max_workers, device_id=device, seed=0, precision=precision, lib_path=lib_path or None, wave_packing=1)
    R = Region(name, create=True, dims=(max_workers, env.S, env.A, env.P, env.J, env.amp_size))
    off = env.offsets_scales()
    S, A = env.S, env.A
    m = R.meta
    m[0:S] = off["state_offset"]; m[S:2 * S] = off["state_scale"]; m[2 * S:3 * S] = off["state_norm_groups"]
    b = 3 * S
    for k in ("action_offset", "action_scale", "action_min", "action_max"):
        m[b:b + A] = off[k]; b += A
    m[b] = env.duration; m[b + 1] = env.D; m[b + 2] = env.max_contacts
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

            def pack_state(s_, ids):
                R.big[ids] = np.concatenate([s_["pose"][ids], s_["vel"][ids], s_["tar"][ids], s_["kin"][ids], s_["clocks"][ids], s_["flags"][ids].astype(np.float64)], axis=1)

            failed = np.zeros(W, dtype=bool)

            def guarded(ids, fn):
                """one request kind (or one step group) of the round: a failure marks ITS slots only -- what the other calls of the round already did to the
                device stands, and their workers get their results"""
                if ids.size == 0:
                    return
                try:
                    fn(ids)
                except Exception as ex_:
                    failed[ids] = True
                    sys.stderr.write("deepmimic_amd.broker: %d request(s) of a round failed: %r\n" % (ids.size, ex_))

            # every kind of request of the round is served by ONE call for all the slots that made it (a slot has one request pending at a time)
            rs = pend[ops == OP_RESET]
            guarded(rs, lambda ids: env.reset(env_ids=ids.astype(np.int32), kin_times=R.dargs[ids, 0], max_times=R.dargs[ids, 1]))
            stp = pend[ops == OP_STEP]
            snap = stp[R.iargs[stp, 4] != 0] if stp.size else stp            # control steps that may be rolled back: the state they start from
            gs, ss = pend[ops == OP_GET_STATE], pend[ops == OP_SET_STATE]
            want = np.concatenate([rs, snap, gs]) if (rs.size or snap.size or gs.size) else rs

            def states(_):
                s_ = env.get_state()
                if ss.size:
                    v = R.big[ss]
                    s_["pose"][ss] = v[:, :P]; s_["vel"][ss] = v[:, P:2 * P]; s_["tar"][ss] = v[:, 2 * P:3 * P]; s_["kin"][ss] = v[:, 3 * P:3 * P + 7]
                    s_["clocks"][ss] = v[:, 3 * P + 7:3 * P + 12]; s_["flags"][ss] = v[:, 3 * P + 12:3 * P + 16].astype(np.int32)
                    env.set_state(pose=s_["pose"], vel=s_["vel"], tar=s_["tar"], kin=s_["kin"], clocks=s_["clocks"], flags=s_["flags"])
                if want.size:
                    pack_state(s_, want)
            guarded(np.concatenate([want, ss]) if (want.size or ss.size) else want, states)
            t_s0 = time.perf_counter(); stats["t_pre"] += t_s0 - t_g1
            if stp.size:
                kmat = np.column_stack([R.dargs[stp, 0], R.iargs[stp, :4].astype(np.float64)])
                uniq, inv = np.unique(kmat, axis=0, return_inverse=True)
                for g in range(uniq.shape[0]):
                    dt, n_upd, has_act, end_early, want_amp = float(uniq[g, 0]), int(uniq[g, 1]), int(uniq[g, 2]), int(uniq[g, 3]), int(uniq[g, 4])

                    def step_group(ids, dt=dt, n_upd=n_upd, has_act=has_act, end_early=end_early, want_amp=want_amp):
                        t_c0 = time.perf_counter()
                        out = env.step_envs(ids.astype(np.int32), R.action[ids, :A] if has_act else None, dt, n_upd, end_early=bool(end_early), amp=bool(want_amp))
                        stats["t_call"] += time.perf_counter() - t_c0
                        R.state[ids] = out["state"]; R.reward[ids] = out["reward"]
                        R.flags[ids, 0] = out["terminate"]; R.flags[ids, 1] = out["valid"]; R.flags[ids, 2] = out["episode_end"]
                        R.clocks[ids] = out["clocks"]
                        if want_amp and env.amp_size:
                            R.amp[ids] = out["amp_obs"]
                        stats["launches"] += 1; stats["steps"] += len(ids)
                    guarded(stp[np.ravel(inv) == g], step_group)
            t_s1 = time.perf_counter(); stats["t_step"] += t_s1 - t_s0
            qs = pend[(ops == OP_QUERY) | (ops == OP_QUERY_AMP)]

            def queries(ids):
                q = env.query()
                R.state[ids] = q["state"][ids]; R.reward[ids] = q["reward"][ids]
                R.flags[ids, 0] = q["terminate"][ids]; R.flags[ids, 1] = q["valid"][ids]; R.flags[ids, 2] = q["episode_end"][ids]; R.flags[ids, 3] = q["need_new_action"][ids]
                qa = pend[ops == OP_QUERY_AMP]
                if qa.size and env.amp_size:
                    R.amp[qa] = env.query_amp()[qa]
            guarded(qs, queries)

            def experts(ids):
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

    def __init__(self, tables, seed: int = 0, device_id: int = 0, precision: int = 32, lib_path=None, max_workers=None):
        self.tables = tables
        c = tables.cfg
        if platform.machine() != "x86_64" or not sys.platform.startswith("linux"):
            raise NotImplementedError("DM_FACADE_SHARED: the broker uses Linux futexes by x86-64 syscall number and relies on x86 store order")
        if tables.goal_kind != 0 or tables.num_clips != 1 or (c.enable_rand_perturbs and np.isfinite(c.perturb_time_min)) or c.enable_rand_rot_reset:
            raise NotImplementedError("DM_FACADE_SHARED serves imitate / imitate_amp scenes with one clip, no perturbations, no random yaw")
        self._seed, self._ep, self._expert_calls = int(seed) & (2 ** 64 - 1), 1, 0          # (a fresh one-env ctx has consumed episode 0 in dm_create's own reset)
        W = int(max_workers or os.environ.get("DM_FACADE_SHARED_MAX", "256"))
        name = region_name(tables, precision, device_id)
        self.R = self._attach(name, tables, W, device_id, precision, lib_path or os.environ.get("DM_HIP_LIB") or "")
        R = self.R
        # claim a slot under the region's file lock
        with _open_private("/dev/shm/%s.lock" % name, "a+") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            free = np.nonzero(R.owner == 0)[0]
            if free.size == 0:
                raise RuntimeError("DM_FACADE_SHARED: all %d slots of %s are taken (DM_FACADE_SHARED_MAX)" % (R.W, name))
            self.slot = int(free[0])
            R.ack[self.slot] = R.req[self.slot]
            R.owner[self.slot] = os.getpid()
        self.N, self.S, self.A, self.P, self.J, self.amp_size, self.G = 1, R.S, R.A, R.P, R.J, R.AMP, 0
        m, S, A = R.meta, R.S, R.A
        b = 3 * S + 4 * A
        self.duration, self.D, self.max_contacts = float(m[b]), int(m[b + 1]), int(m[b + 2])
        self.physics, self.num_clips, self.has_obj, self.has_perturbs, self.precision = 1, 1, False, False, precision
        self._timer = (c.timer_type, float(c.time_lim_min), float(c.time_lim_max), float(c.time_lim_exp))
        self._gen_addr, self._wake_addr = R.addr("wake", 1), R.addr("wake", 0)
        self._state, self._snap = None, None

    @staticmethod
    def _attach(name, tables, W, device, precision, lib_path, timeout=180.0):
        """Open the region; start the owner process if there is none (first worker, under the file lock)."""
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
                    owner = subprocess.Popen([sys.executable, "-m", "deepmimic_amd.broker", "--serve", name, str(W), str(device), str(precision), lib_path],
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
        return R

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
        if kin_times is None:                                 # counter mode: the draws a one-env ctx of its own would make on the device (stream 0 / 1)
            kt = self.duration * streams.reset_rand01(self._seed, 0, self._ep, 0)
        else:
            kt = float(np.ravel(kin_times)[0])
        if max_times is None:
            ty, lo, hi, ex = self._timer
            mt = hi if not hi > lo else model.draw_time_limit(ty, lo, hi, ex, streams.reset_rand01(self._seed, 0, self._ep, 1))
        else:
            mt = float(np.ravel(max_times)[0])
        self._ep += 1
        self.R.dargs[self.slot, 0] = kt; self.R.dargs[self.slot, 1] = mt
        self._call(OP_RESET)
        self._state = self._unpack()                          # the owner hands the reset state back with the reply: get_state() right after costs no round trip

    def set_time_limits(self, lo, hi, ex=None):
        self._timer = (self._timer[0], float(lo), float(hi), self._timer[3] if ex is None else float(ex))

    def set_mode(self, test_mode):
        pass

    def _out(self, amp=False):
        R, i = self.R, self.slot
        out = dict(state=R.state[i:i + 1].copy(), reward=R.reward[i:i + 1].copy(), terminate=R.flags[i:i + 1, 0].copy(), valid=R.flags[i:i + 1, 1].copy(),
                   episode_end=R.flags[i:i + 1, 2].copy())
        if amp and self.amp_size:
            out["amp_obs"] = R.amp[i:i + 1].copy()
        return out

    def step(self, actions=None, timestep=1.0 / 600, n_updates=20, auto_reset=False, open_loop=False, amp=False, end_early=None):
        if auto_reset or open_loop:
            raise NotImplementedError("SharedEnv.step: the facade resets explicitly and always hands an action (or none)")
        R, i = self.R, self.slot
        if actions is not None:
            R.action[i, :self.A] = np.asarray(actions, dtype=np.float32).reshape(self.A)
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
        self._call(OP_QUERY)
        out = self._out()
        out["need_new_action"] = self.R.flags[self.slot:self.slot + 1, 3].copy()
        return out

    def query_amp(self):
        self._call(OP_QUERY_AMP)
        return self.R.amp[self.slot:self.slot + 1].copy()

    def amp_expert(self, n, times=None, ground_h=None):
        assert n == 1
        if times is None:            # counter mode: the draw dm_amp_expert makes for a one-env ctx of its own (dm_host.cpp: key (seed, 0x414D50, call, sample 0))
            from . import streams
            times = [self.duration * streams.reset_rand01(self._seed, 0x414D50, self._expert_calls, 0)]
            self._expert_calls += 1
        self.R.dargs[self.slot, 0] = float(np.ravel(times)[0]); self.R.dargs[self.slot, 1] = 0.0 if ground_h is None else float(np.ravel(ground_h)[0])
        self._call(OP_AMP_EXPERT)
        return self.R.amp[self.slot:self.slot + 1].copy()

    def _unpack(self):
        v, P = self.R.big[self.slot].copy(), self.P
        return dict(pose=v[None, :P], vel=v[None, P:2 * P], tar=v[None, 2 * P:3 * P], kin=v[None, 3 * P:3 * P + 7], clocks=v[None, 3 * P + 7:3 * P + 12],
                    flags=v[None, 3 * P + 12:3 * P + 16].astype(np.int32))

    def get_state(self):
        if self._state is not None:
            return {k: a.copy() for k, a in self._state.items()}
        self._call(OP_GET_STATE)
        return self._unpack()

    def set_state(self, pose=None, vel=None, tar=None, kin=None, clocks=None, flags=None):
        cur = self.get_state()
        for k, a in (("pose", pose), ("vel", vel), ("tar", tar), ("kin", kin), ("clocks", clocks), ("flags", flags)):
            if a is not None:
                cur[k] = np.asarray(a, dtype=np.float64).reshape(1, -1)
        self.R.big[self.slot] = np.concatenate([cur["pose"][0], cur["vel"][0], cur["tar"][0], cur["kin"][0], cur["clocks"][0], np.asarray(cur["flags"][0], dtype=np.float64)])
        self._state = None
        self._call(OP_SET_STATE)

    def snapshot(self):
        """the rollback point of the control step that follows: filled in by that step() (the owner reads the state of every stepping worker with one
        copy before the launch), so taking it costs no round trip; a snapshot that is not followed by a step is fetched on first use"""
        self._snap = _LazySnap(self)
        return self._snap

    def restore(self, snap):
        self.set_state(**{k: snap[k] for k in ("pose", "vel", "tar", "kin", "clocks", "flags")})

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
        self.update(self._env.get_state())
        return dict.__getitem__(self, k)


if __name__ == "__main__":
    if len(sys.argv) >= 7 and sys.argv[1] == "--serve":
        serve(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6])
    else:
        sys.exit("usage: python -m deepmimic_amd.broker --serve <region> <max workers> <device> <precision> <lib path>   (pickled scene tables on stdin)")
