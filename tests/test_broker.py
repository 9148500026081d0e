"""DM_FACADE_SHARED=1 (deepmimic_amd/broker.py): W cDeepMimicCore worker processes behind ONE context -- one dm_step_envs launch for
all control steps that are pending -- must see exactly what they see with a context each: same observations, rewards, flags, clock, through episode
ends, resets (reference-order draws and counter draws), mid-step peeks (rollback / replay) and the AMP observation path.  CPU: the emulator build."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMPAT = os.path.join(ROOT, "deepmimic_amd", "compat")


def _tables(scene):
    """`scene`: "imitate" / "imitate_amp" (humanoid3d_walk under that scene), an asset name (goal scenes, multi-clip datasets), "perturbs" (imitate + random
    perturbations)"""
    from deepmimic_amd import model
    if scene in ("imitate", "imitate_amp", "perturbs"):
        t = model.load_asset("humanoid3d_walk")
        t.cfg.scene = "imitate" if scene == "perturbs" else scene
        if scene == "perturbs":
            t.cfg.enable_rand_perturbs = True; t.cfg.perturb_time_min = 0.05; t.cfg.perturb_time_max = 0.15; t.cfg.min_pertrub_duration = 0.02; t.cfg.max_perturb_duration = 0.1
    else:
        t = model.load_asset(scene)
    t.cfg.time_lim_min = t.cfg.time_lim_max = t.cfg.time_end_lim_min = t.cfg.time_end_lim_max = 0.175     # the episode timer ends inside a control step (update 105)
    return t


def _worker(rank, shared, lib, scene, n_updates, rng_mode, shm, q, barrier, physics=1):
    os.environ.update(DM_HIP_LIB=lib, DM_ALLOW_EMULATOR="1", DM_PRECISION="64", DM_FACADE_SHARED="1" if shared else "0", DM_RNG=rng_mode,
                      DM_FACADE_SHM=shm, DM_FACADE_SHARED_MAX="8", DM_PHYSICS=str(physics))
    sys.path.insert(0, ROOT); sys.path.insert(0, COMPAT)
    import warnings
    warnings.simplefilter("ignore", RuntimeWarning)       # (DM_RNG=reference warns for multi-clip datasets)
    from DeepMimicCore import DeepMimicCore
    t = _tables(scene)
    scene_kind = t.cfg.scene
    core = DeepMimicCore.cDeepMimicCore(False)
    core.SeedRand(100 + rank); core.LoadTables(t, 10); core.Init()
    if rank == 1:
        core.SetMode(1)                                  # one worker in test mode: mode and episode limits are per-context settings on the device, the owner groups by them
    if barrier is not None:
        barrier.wait()                                   # all workers attached: their control steps meet in the owner
    rng = np.random.default_rng(rank)
    seen = []
    for u in range(n_updates):
        if core.NeedNewAction(0):
            seen.append(("s", np.array(core.RecordState(0)), core.CalcReward(0)))
            if core.GetGoalSize(0):
                seen.append(("g", np.array(core.RecordGoal(0)), 0.0))
            if scene_kind != "imitate":
                seen.append(("a", np.array(core.RecordAMPObsAgent(0)), float(np.sum(core.RecordAMPObsExpert(0)))))
            core.SetAction(0, [float(x) for x in (0.15 * rng.normal(size=core.GetActionSize(0))).astype(np.float32)])
        core.Update(1.0 / 600)
        if u in (33, 77):                                # a look inside a control step: rollback + replay on the batched route
            seen.append(("p", np.array(core.RecordState(0)), core.CalcReward(0)))
        end, ok = core.IsEpisodeEnd(), core.CheckValidEpisode()
        seen.append(("f", np.array([float(end), float(ok), core.CheckTerminate(0), core.GetTime()]), 0.0))
        if end or not ok:
            core.Reset()
    st = dict(core.stats)
    core.Shutdown()
    q.put((rank, seen, st))


def _run(lib, shared, W, scene, n_updates, rng_mode, shm, physics=1):
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); barrier = ctx.Barrier(W) if shared else None
    ps = [ctx.Process(target=_worker, args=(r, shared, lib, scene, n_updates, rng_mode, shm, q, barrier, physics)) for r in range(W)]
    for p in ps:
        p.start()
    res = {}
    try:
        import queue
        import time
        t_end = time.monotonic() + 600
        while len(res) < W:
            try:
                rank, seen, st = q.get(timeout=2)
                res[rank] = (seen, st)
            except queue.Empty:
                dead = [p.exitcode for p in ps if p.exitcode not in (None, 0)]
                assert not dead, "a worker died: exit codes %s" % dead
                assert time.monotonic() < t_end, "workers did not finish"
        for p in ps:
            p.join(timeout=60)
            assert p.exitcode == 0
    finally:
        for p in ps:
            if p.is_alive():
                p.kill()
    return res


# one asset of every scene kind the context serves: single-clip imitate (host draws in the reference's order) and imitate_amp, the four kinds of goal
# scene + the dribble ball, a four-clip dataset, random perturbations, DM-physics v2
CASES = [("imitate", "reference", 1), ("imitate_amp", "counter", 1), ("amp_heading_zombie", "counter", 1), ("amp_target_zombie", "counter", 1), ("amp_heading_getup", "counter", 1),
         ("amp_strike_punch", "counter", 1), ("amp_dribble_zombie", "reference", 1), ("amp_heading_clips4", "reference", 1), ("perturbs", "reference", 1), ("imitate", "counter", 2)]
# ("reference": the worker's generators in the reference's order -- the draws the device makes come off the worker's draw tape, which travels with its requests;
#  "counter": the counter-based streams of the worker's seed)


@pytest.mark.parametrize("scene,rng_mode,physics", CASES, ids=["%s-%s-v%d" % c for c in CASES])
def test_shared_workers_equal_private_contexts(emu_lib, scene, rng_mode, physics):
    W, n = 3, 130
    shm = "dmtest_%d_%s_%d" % (os.getpid(), scene, physics)
    shared = _run(emu_lib, True, W, scene, n, rng_mode, shm, physics)
    private = _run(emu_lib, False, W, scene, n, rng_mode, shm, physics)
    for r in range(W):
        a, b = shared[r][0], private[r][0]
        assert len(a) == len(b)
        for k, (x, y) in enumerate(zip(a, b)):
            assert x[0] == y[0] and np.array_equal(x[1], y[1]) and x[2] == y[2], (r, k, x[0])
        assert any(x[0] == "f" and x[1][0] == 1.0 for x in a)                       # an episode ended inside the window
        assert shared[r][1]["rollbacks"] == private[r][1]["rollbacks"] == 2
        assert shared[r][1]["launches"] > 0
    # the owner leaves a few seconds after its last worker and takes the region with it; the (empty) lock file is the test's to remove
    import glob
    import time
    t_end = time.monotonic() + 45
    while os.path.exists("/dev/shm/" + shm) and time.monotonic() < t_end:
        time.sleep(0.2)
    if os.path.exists("/dev/shm/" + shm):               # say what the owner was doing
        from deepmimic_amd.broker import Region
        diag = {}
        try:
            R = Region(shm)
            pid = int(R.hdr[8]); diag = {"owner_pid": pid, "alive": os.path.exists("/proc/%d" % pid), "slots": R.owner.tolist(), "req": R.req.tolist(), "ack": R.ack.tolist(), "hdr0": int(R.hdr[0])}
            if diag["alive"]:
                diag["wchan"] = open("/proc/%d/wchan" % pid).read(); diag["stat"] = open("/proc/%d/stat" % pid).read().split()[2]
            R.close()
        except Exception as ex:
            diag["error"] = repr(ex)
        log = "/dev/shm/%s.log" % shm
        diag["log"] = open(log, "rb").read()[-2000:] if os.path.exists(log) else None
        raise AssertionError("the owner process did not leave: %r" % (diag,))
    for f in glob.glob("/dev/shm/%s.*" % shm):
        os.unlink(f)


def test_region_layout_round_trip():
    from deepmimic_amd.broker import Region
    name = "dmtest_layout_%d" % os.getpid()
    a = Region(name, create=True, dims=(4, 227, 28, 43, 15, 226, 3, 1 | 4 | 8, 4))
    try:
        b = Region(name)
        assert (b.W, b.S, b.A, b.P, b.J, b.AMP, b.G, b.FB, b.NC) == (4, 227, 28, 43, 15, 226, 3, 13, 4)
        assert b.big.shape == (4, 3 * 43 + 16 + 21 + 16 + 15 * 25) and b.goal.shape == (4, 3)
        a.state[2, 5] = 1.5; a.req[3] = 7
        assert b.state[2, 5] == 1.5 and b.req[3] == 7 and b.addr("ack", 1) - b.addr("ack", 0) == 4
        b.close()
    finally:
        a.close(unlink=True)


def test_broker_refuses_files_it_does_not_own(tmp_path, monkeypatch):
    """/dev/shm is world-writable: the lock / log files are opened without following symlinks and only when they are regular files of this user; a region
    of another uid is not attached; the scene tables travel to the owner process down a pipe (no pickle file under /dev/shm)."""
    import inspect
    from deepmimic_amd import broker
    victim = tmp_path / "victim.txt"; victim.write_text("keep")
    link = "/dev/shm/dmtest_symlink_%d.lock" % os.getpid()
    os.symlink(str(victim), link)
    try:
        with pytest.raises(OSError):                     # O_NOFOLLOW: ELOOP
            broker._open_private(link, "a+")
        assert victim.read_text() == "keep"
    finally:
        os.unlink(link)
    path = "/dev/shm/dmtest_private_%d.lock" % os.getpid()
    try:
        with broker._open_private(path, "a+") as f:
            assert (os.fstat(f.fileno()).st_mode & 0o777) == 0o600
        monkeypatch.setattr(os, "getuid", lambda: 54321)      # "another user's file"
        with pytest.raises(RuntimeError, match="not a regular file of this user"):
            broker._open_private(path, "a+")
        with pytest.raises(RuntimeError, match="belongs to uid"):
            broker._check_region_owner(os.path.basename(path))
    finally:
        monkeypatch.undo()
        os.unlink(path)
    src = inspect.getsource(broker)
    assert "pickle.load(sys.stdin.buffer)" in src and '.tables"' not in src and "tables_path" not in src
