#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the vectorised imitate env (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--envs 4096] [--scene humanoid3d_walk]

One "step" = one 30 Hz control step of EVERY env on the rank = 20 scene updates (20 SPD solves + 40 rigid-body substeps) + reward + observation
+ auto-reset per env.  By default (`--groups 0` = auto) the rank's envs run as two env groups of n / 2 on their own streams (deepmimic_amd/groups.py:
two `k_env_step_duo` launches of 2048 envs per control step, drifting apart in phase and filling each other's wave-time tail); `--groups 1` is one
launch per control step.  Workload (configs[1] of BASELINE.json): humanoid3d_walk, 4096 envs per GPU, fixed-action stream A1 (open-loop mocap tracking,
generated on device), inputs resident in HBM.  The line carries `roofline` (per launch; `valu` = the binding figures from the committed counter files,
stamped with the kernel sources they were taken on), `sustained` (a >= 8 s window behind the timed steps: longer than a 5 s outside GPU-busy sampler's period), `closed_loop` (N = 1: the same envs driven by the on-device policy, an extra -- never `value`) and `cpu_baseline`.
N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), env shards are independent (weak scaling); the only collective is the per-step
all-gather of (state, reward, terminate) for the learner, one exchange per env group on the group's stream.
`python bench.py --gpus N` with no WORLD_SIZE in the environment launches its own N ranks (one per GPU, rank r on GPU r) under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` -- the reference's precedent for a
self-launching driver is mpi_run.py:16-24 (`mpiexec -n W python3 DeepMimic_Optimizer.py`); under an external launcher
(WORLD_SIZE set) it is one of the ranks.  Rank 0 prints ONE JSON line and refuses to print one whose n_gpus is not --gpus.
`--backend gloo` is the CPU test harness of this N > 1 path (emulator build of the kernels, tests/test_bench_launcher.py).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); kernels of two streams that share a queue do not overlap.  Eight keeps the
# env groups' streams, torch's and RCCL's apart whatever the creation order (no effect on a one-stream run: measured).  Read at HIP initialisation.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes_per_env_step(env):
    """HBM bytes one env-step must move: env record in + out, observation/reward/flags out."""
    rec = 4 * (3 * env.P + env.D + 8) + 8 * 6 + 4 * 4            # pose, vel, tar, tau, kin | clocks | flags
    out = 4 * env.S + 4 + 3 * 4                                   # state, reward, terminate/valid/episode_end
    extra = 0
    if getattr(env, "amp_size", 0):                               # imitate_amp and the task scenes: pose | vel history written at the action latch
        extra += 4 * 2 * env.P                                    # (this bench asks for no AMP observation: its read-back of the row and the 4 amp_size output are not moved)
    if getattr(env, "G", 0):                                      # task scenes: the goal row (16 doubles) in + out, RecordGoal kept for dm_last_goals
        extra += 2 * 16 * 8 + 4 * env.G
    if getattr(env, "has_obj", False):                            # dribble_amp: the free body's record in + out
        extra += 2 * 16 * 4
    return 2 * rec + out + extra


def kernel_source_sha1():
    """hash of the device sources this run's library was built from (tools/collect_profiles.py stamps the same hash into every counter file)"""
    import hashlib
    h = hashlib.sha1()
    try:
        for f in ("dm_device.h", "dm_device_duo.h", "dm_types.h", "dm_math.h"):
            with open(os.path.join(ROOT, "deepmimic_amd", "csrc", f), "rb") as fh:
                h.update(fh.read())
        with open(os.path.join(ROOT, "deepmimic_amd", "csrc", "Makefile")) as fh:          # the code-generation flags of the kernel families, not the host-side rules
            h.update("".join(l for l in fh if l.startswith(("HIPFLAGS", "NOLICM_IDS", "SCHED_IDS", "licmflag", "ARCH"))).encode())
        return h.hexdigest()
    except OSError:
        return None


def measured_traffic(scene, n, kernel=None, physics=1):
    """(HBM bytes per step-kernel launch, source file) from the committed rocprofv3 PMC passes (profiles/r0N_traffic*.json,
    collected as MI355X_MICROARCH.md prescribes: separate --pmc passes for FETCH_SIZE and WRITE_SIZE; tools/collect_profiles.py),
    newest round first; (None, None) when no profile of this workload is committed.  It is a committed counter measurement of
    the same workload and kernel, NOT a counter read of the run that prints it (PMC passes serialise the kernels): the bench
    line names the file in `roofline.traffic_source`.  The counter files are of the DM-physics v1 kernels unless they say `"physics": 2`: a v2 run
    does not borrow them (its `traffic` is null)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic*.json")), reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
            if t.get("scene") == scene and int(t.get("envs", -1)) == n and (kernel is None or t.get("kernel", kernel) == kernel) and int(t.get("physics", 1)) == physics:
                measured_traffic.current = (t.get("kernel_source_sha1") == kernel_source_sha1()) if t.get("kernel_source_sha1") else None
                return float(t["hbm_bytes_per_launch"]), os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def measured_valu(scene, n, kernel, env_steps_per_s, physics=1):
    """The figures that actually bind this kernel (VALU issue + dependent-chain latency; HBM is ~0 by construction), from the committed
    rocprofv3 PMC passes of the same workload and kernel (profiles/r0N_pmc_sq*.json, r0N_flops*.json; newest round first): VALU busy
    fraction of SIMD time, the share of wave cycles spent in s_waitcnt, VALU instructions per env-step, and -- issued lane-flops per
    env-step x the rate of THIS run -- issued TFLOP/s against the 157.3 TFLOP/s fp32 vector peak.  Committed counters, not counters of
    this run (PMC passes serialise the kernels); `source` names the files."""
    import glob
    out = {"binding": "VALU issue + dependent-chain latency (fp32 vector)", "fp32_vector_peak_tflops": 157.3, "source": []}
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_sq*.json")), reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
            if t.get("scene") == scene and int(t.get("envs", -1)) == n and t.get("kernel", kernel) == kernel and int(t.get("physics", 1)) == physics:
                out["valu_busy"] = t["derived"]["valu_busy_fraction_of_simd_time"]
                out["source_current"] = (t.get("kernel_source_sha1") == kernel_source_sha1()) if t.get("kernel_source_sha1") else None
                out["wait_fraction"] = t["SQ_WAIT_ANY"] / t["SQ_WAVE_CYCLES"]
                out["valu_instructions_per_env_step"] = t["derived"]["valu_instructions_per_env_step"]
                out["source"].append(os.path.relpath(path, ROOT))
                break
        except Exception:
            continue
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_flops*.json")), reverse=True):
        try:
            with open(path) as f:
                t = json.load(f)
            if t.get("scene") == scene and int(t.get("envs", -1)) == n and t.get("kernel", kernel) == kernel and int(t.get("physics", 1)) == physics:
                out["issued_flops_per_env_step"] = t["issued_flops_per_env_step"]
                out["issued_tflops"] = t["issued_flops_per_env_step"] * env_steps_per_s / 1e12
                out["frac_of_fp32_peak"] = out["issued_tflops"] / 157.3
                out["source"].append(os.path.relpath(path, ROOT))
                break
        except Exception:
            continue
    return out if out["source"] else None


def self_launch(argv, n):
    """--gpus N > 1 without an external launcher: start N ranks (one per GPU) of this script and hand their output through."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def timer_limits_of(tables, test_mode=True):
    c = tables.cfg
    tmin, tmax = float(c.time_lim_min), float(c.time_lim_max)
    if test_mode and c.time_end_lim_max is not None:
        tmin = tmax = float(c.time_end_lim_max)
    return tmin, tmax


def cpu_baseline_worker(scene, env_id, budget_s):
    """One process of the CPU baseline (no torch, no HIP): the oracle on bench.py's own workload -- open-loop tracking with
    auto-reset, start phase and reset draws keyed like the GPU run's env `env_id`.  Prints one JSON line."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from deepmimic_amd import model, streams
    tables = model.load_asset(scene)
    variant = "native" if os.environ.get("DM_ORACLE_NATIVE") == "1" else ""      # built once by the parent (cpu_baseline)
    o = oracle_lib.Oracle(tables, variant=variant)
    tmin, tmax = timer_limits_of(tables)
    t0 = float(streams.reset_phase(np.array([env_id]), o.duration)[0])
    o.rollout_auto_reset(30, 1234, env_id, t0, tmin, tmax)      # warm-up
    steps, secs, resets, live = 0, 0.0, 0, 0
    while secs < budget_s:
        s, r, _, lv = o.rollout_auto_reset(300, 1234, env_id, t0, tmin, tmax)
        steps += 300; secs += s; resets += r; live += lv
    print(json.dumps({"steps": steps, "secs": secs, "resets": resets, "live": live, "native": bool(variant)}))


def cpu_oracle_rate(scene, nproc, budget_s=6.0):
    """aggregate env-steps/s of `nproc` independent oracle processes (the CPU restatement, one env each: what W reference-style workers would do on this host's cores)"""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    env = dict(os.environ)
    try:
        oracle_lib.build("native"); env["DM_ORACLE_NATIVE"] = "1"
    except Exception:
        oracle_lib.build("all"); env["DM_ORACLE_NATIVE"] = "0"
    procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(i), "--scene", scene, "--cpu-budget", str(budget_s)],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for i in range(nproc)]
    total = 0.0
    for p_ in procs:
        so, se = p_.communicate(timeout=20 * budget_s + 120)
        lines = [l for l in so.splitlines() if l.startswith("{")]
        if p_.returncode != 0 or not lines:
            raise RuntimeError("worker rc=%s: %s" % (p_.returncode, se.strip()[-300:]))
        r = json.loads(lines[-1]); total += r["steps"] / r["secs"]
    return total


def cpu_baseline(scene, budget_s=12.0):
    """The CPU path timed beside the GPU line on this host: the oracle restatement (kind "port"; DeepMimicCore + Bullet is
    not buildable here) on the SAME workload -- open-loop tracking with auto-reset -- as (a) one process and (b) one process per
    host core, each an independent env (the reference scales by processes: mpi_run.py).  `value`/`cores` are the all-core figures;
    the single-process rate is reported next to them."""
    import subprocess

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    env = dict(os.environ)
    try:        # host-tuned build, made HERE (a -march=native .so from another machine must not be reused) and once (the workers only load it)
        oracle_lib.build("native"); env["DM_ORACLE_NATIVE"] = "1"
    except Exception:
        oracle_lib.build("all"); env["DM_ORACLE_NATIVE"] = "0"

    def run(nproc):
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(i), "--scene", scene,
                                   "--cpu-budget", str(budget_s)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
                 for i in range(nproc)]
        res = []
        for p in procs:
            so, se = p.communicate(timeout=20 * budget_s + 120)
            lines = [l for l in so.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not lines:
                raise RuntimeError("worker rc=%s: %s" % (p.returncode, se.strip()[-300:]))
            res.append(json.loads(lines[-1]))
        return sum(r["steps"] / r["secs"] for r in res), res

    try:
        cores = os.cpu_count() or 1
        one, r1 = run(1)
        allc, rn = run(cores) if cores > 1 else (one, r1)
        live = sum(r["live"] for r in rn) / max(1, sum(r["steps"] for r in rn))
        return {"value": allc, "unit": "env-steps/s", "cores": cores, "kind": "port", "single_core_value": one,
                "sample": "%d independent processes x %.0f s of 300-step open-loop rollouts WITH auto-reset (same episode mixture as the "
                          "GPU line: %.0f %% of the steps live, %d resets), oracle built -O3%s; single process: %.1f env-steps/s"
                          % (cores, budget_s, 100 * live, sum(r["resets"] for r in rn), " -march=native" if rn[0]["native"] else "", one)}
    except Exception as ex:  # the baseline is reported, never required
        return {"value": None, "unit": "env-steps/s", "cores": 1, "kind": "port", "sample": "failed: %r" % (ex,)}


def _facade_worker(scene, steps, batch, barrier, q, shared=False):
    """one worker of the drop-in route: ONE env per cDeepMimicCore, the reference's update_world loop"""
    os.environ["DM_FACADE_BATCH"] = batch
    os.environ["DM_FACADE_SHARED"] = "1" if shared else "0"      # shared: every worker behind ONE context / one launch per control step (deepmimic_amd/broker.py)
    sys.path.insert(0, os.path.join(ROOT, "deepmimic_amd", "compat"))
    from DeepMimicCore import DeepMimicCore
    from deepmimic_amd import model
    t = model.load_asset(scene)
    core = DeepMimicCore.cDeepMimicCore(False)
    core.SeedRand(1 + os.getpid() % 1000); core.LoadTables(t, 10); core.Init()
    rng = np.random.default_rng(os.getpid())
    A, dt = core.GetActionSize(0), 1.0 / 600
    lo, hi = np.array(core.BuildActionBoundMin(0)), np.array(core.BuildActionBoundMax(0))

    def run(n_steps):
        n = 0
        while n < n_steps:
            if core.NeedNewAction(0):
                core.RecordState(0); core.RecordGoal(0); core.CalcReward(0)
                core.SetAction(0, [float(x) for x in np.clip(0.1 * rng.normal(size=A), lo, hi)]); n += 1
            core.Update(dt)
            if (not core.CheckValidEpisode()) or core.IsEpisodeEnd():
                core.RecordState(0); core.CalcReward(0); core.CheckTerminate(0)
                core.Reset()
    run(20)
    core.stats.update(launches=0, updates=0, rollbacks=0)
    if barrier is not None:
        barrier.wait()
    t0 = time.perf_counter(); run(steps); el = time.perf_counter() - t0
    res = {"elapsed": el, "steps": steps, "launches": core.stats["launches"], "updates": core.stats["updates"]}
    core.Shutdown()
    if q is not None:
        q.put(res)
    return res


def facade_bench(scene, steps, workers=(1,), private_too=True):
    """The drop-in path the reference's trainer uses: ONE env per cDeepMimicCore, driven by the reference's update_world loop
    (DeepMimic.py:62-80: NeedNewAction / RecordState / CalcReward / SetAction once per 1/30 s; Update, CheckValidEpisode,
    IsEpisodeEnd once per 1/600 s).  Reports env-steps/s of one worker with the control step batched into one launch
    (DM_FACADE_BATCH=1, default) and update by update, and -- `--workers W ...` -- the AGGREGATE of W such worker processes sharing
    one GPU (the reference scales by processes: mpi_run.py:16-24), all timed from a common barrier to the last finisher."""
    import multiprocessing as mp
    out = {}
    for tag, batch in (("batched", "1"), ("per_update", "0")):
        r = _facade_worker(scene, steps, batch, None, None)
        out[tag] = {"env_steps_per_s": steps / r["elapsed"], "ms_per_control_step": 1e3 * r["elapsed"] / steps,
                    "launches_per_control_step": r["launches"] / steps, "updates_per_control_step": r["updates"] / steps}
    agg, agg_shared = {}, {}
    ctx = mp.get_context("spawn")
    for shared, dst in ((False, agg), (True, agg_shared)):
        os.environ["DM_FACADE_SHARED_MAX"] = str(max(2, max(workers)))
        for w in workers:
            if w <= 1 and not shared:
                dst["1"] = out["batched"]["env_steps_per_s"]; continue
            if not shared and w > 1 and not private_too:
                continue
            barrier, q = ctx.Barrier(w), ctx.Queue()
            procs = [ctx.Process(target=_facade_worker, args=(scene, steps, "1", barrier, q, shared)) for _ in range(w)]
            for p_ in procs:
                p_.start()
            res = [q.get(timeout=1800) for _ in procs]
            for p_ in procs:
                p_.join()
            dst[str(w)] = w * steps / max(r["elapsed"] for r in res)
        time.sleep(4.0)          # the owner process of the shared route leaves a few seconds after its last worker
    # the CPU column: W independent processes of the oracle on this host's cores, beside each W of the GPU routes (the reference scales by processes too)
    cpu = {}
    for w in workers:
        try:
            cpu[str(w)] = cpu_oracle_rate(scene, w)
        except Exception as ex:                                      # noqa: BLE001
            cpu[str(w)] = "failed: %r" % (ex,)
    print(json.dumps({"metric": "facade env-steps/s, one env per cDeepMimicCore (%s)" % scene, "value": out["batched"]["env_steps_per_s"],
                      "aggregate_env_steps_per_s_by_workers_cpu_oracle": cpu,
                      "unit": "env-steps/s", "n_gpus": 1, "steps": steps, "higher_is_better": True, "dtype": "f32", "data": "synthetic",
                      "config": {"workload": "%s, 1 env per worker, reference driver protocol, random actions N(0, 0.1^2), auto reset by the driver" % scene},
                      "aggregate_env_steps_per_s_by_workers": agg,
                      "aggregate_env_steps_per_s_by_workers_shared": agg_shared,      # DM_FACADE_SHARED=1: the workers behind one context (deepmimic_amd/broker.py)
                      "host_cores": os.cpu_count(), **out}))


def latency_ceiling(tables, args, device_id, kname, family_id, n_envs):
    """The ceiling of THIS kernel shape: a control step is a chain of dependent phases per wavefront, and the chip holds `wave_slots` wavefronts at the
    kernel's occupancy.  Measured live on 64 envs (32 or 64 wavefronts, each alone on its SIMD), same workload, same episode mixture after the same warm-up:
    * one round of waves (the batch's wavefronts fit the slots: the two-per-wave humanoid at 4096 envs): a launch lasts as long as its SLOWEST wave, so the bound is
      envs / (control step of the 64-env batch, which is the slowest of its waves);
    * more than one round (one character per wave at 4096 envs: dog3d, dribble_amp, --wave-packing 1): finished slots are refilled, the bound is
      slots x envs_per_wave / MEAN wave time -- the 64-env step time scaled by mean / max of the per-wave cycle totals of one profiled step of the same batch
      (dm_probe 3, the tap build of the kernel: a ratio of two cycle counts of one run, not a time).  `frac` is against the bound that applies."""
    from deepmimic_amd.core import BatchEnv
    from deepmimic_amd import streams
    n0 = 64
    e = BatchEnv(tables, n0, device_id=device_id, seed=1234, precision=args.precision, test_mode=True, wave_packing=args.wave_packing, physics=args.physics)
    e.reset(kin_times=streams.reset_phase(np.arange(n0), e.duration))
    ms = e.bench_rollout(60, 200) / 200
    epw = 2 if kname == "k_env_step_duo" else 1
    ratio = None
    try:
        e.probe(3, 1.0 / 600)
        tot = e.debug("prof").sum(1)[::epw]
        tot = tot[tot > 0]
        ratio = float(tot.mean() / tot.max()) if tot.size else None
    except Exception:
        pass
    e.close()
    occ = None
    try:        # waves per SIMD of the shipped kernel from the compiler's resource remarks (deepmimic_amd/csrc/build/k_f32_<family>.o.res)
        import re
        txt = open(os.path.join(ROOT, "deepmimic_amd", "csrc", "build", "k_f%d_%d.o.res" % (args.precision, family_id))).read()
        occ = min(int(x) for x in re.findall(r"Occupancy \[waves/SIMD\]: (\d+)", txt))
    except Exception:
        pass
    waves_per_simd = occ if occ else 2
    slots = 256 * 4 * waves_per_simd
    rounds = -(-(n_envs // epw) // slots)
    one_round = min(n_envs, slots * epw) / (ms * 1e-3)
    out = {"lone_wave_ms_per_step": ms, "waves_per_simd": waves_per_simd, "waves_per_simd_source": "compiler resource remarks" if occ else "assumed",
           "wave_slots": slots, "envs_per_wave": epw, "rounds_of_waves": rounds, "mean_over_max_wave_cycles_64_envs": ratio}
    if rounds <= 1:
        out["env_steps_per_s"] = one_round; out["model"] = "one round: envs / slowest wave"
    elif ratio:
        out["env_steps_per_s"] = slots * epw / (ms * ratio * 1e-3); out["model"] = "%d rounds, slots refilled: slots x envs_per_wave / mean wave time" % rounds
    else:
        out["env_steps_per_s"] = None; out["model"] = "more than one round and no per-wave profile: no bound stated"
    return out


def parity_check(tables, envs, step_and_read, n_sample=64, steps=5, physics=1, actions=None, prepared=None):
    """`checks.parity` of the bench line: `n_sample` strided envs of THIS run's contexts -- where the timed rollout left them, the same step function, the same
    launches (every wave slot occupied, both groups in flight) -- compared with the CPU oracle over `steps` control steps.  Before each step the oracles are put
    where the device envs are (get_state -> Oracle.set_full_state; DM-physics v2: the persistent ground manifolds too, get_manifolds -> Oracle.set_manifolds, and the
    oracles run the v2 restatement), so every sample is one control step (20 updates, 40 substeps) from identical inputs; nothing is written to the device.
    actions (the closed-loop leg): a callable (k, st0) -> [N, A] float32 actions of step k; the step function then takes them (cDeepMimicCore::SetAction at the action boundary).
    The oracle is the CHECKER here (tests/parity_common.sampled_compare), after and outside every timed region.
    Rows: reward scenes/SceneImitate.cpp:7-127, flags scenes/RLSceneSimChar.cpp, via the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity_common as pc
    n = envs.N
    ids = np.unique(np.round(np.linspace(0, n - 1, n_sample)).astype(np.int64))
    mc = envs.envs[0].max_contacts if physics == 2 else None
    beyond32 = lambda: np.stack([envs.debug("fallback"), envs.debug("borrowed")])      # per env: substeps on the 64-lane fallback | on borrowed lanes (two-per-wave kernel)
    fb = [beyond32(), np.zeros((2, n))]

    def on_step(k, st0, out):
        f = beyond32(); fb[1] = fb[1] + (f - fb[0]); fb[0] = f

    dr, ds, alive, ok, ends = pc.sampled_compare(envs.get_state, step_and_read, tables, ids, steps, physics=physics, get_manifolds=envs.get_manifolds if physics == 2 else None,
                                                 max_contacts=mc, actions=actions, on_step=on_step)
    live = dr[alive]
    sl = ds[alive & np.isfinite(ds)]
    q = lambda a, p_: (float(np.quantile(a, p_)) if a.size else None)
    return {"envs": int(ids.size), "steps": int(steps), "samples": int(dr.size), "live": int(alive.sum()), "episode_ends": int(ends),
            "reward_mae": (float(live.mean()) if live.size else None), "reward_p99": q(live, 0.99), "reward_max": (float(live.max()) if live.size else None),
            "reward_max_not_live": float(dr[~alive].max(initial=0.0)), "state_rel_mean": (float(sl.mean()) if sl.size else None), "state_rel_max": (float(sl.max()) if sl.size else None),
            "flags_equal": bool(ok),
            # pair-substeps of the WHOLE batch in which a character had more than 32 constraint rows during these steps: on the 64-lane fallback of the two-per-wave kernel (dm_get_debug
            # "fallback"; both characters of a pair count one) or on borrowed lanes ("borrowed": one heavy character, the pair within 64 rows), and how many of the sampled envs were among them
            "fallback_share": float(fb[1][0].sum() / 2 / max(1, (n // 2) * 40 * steps)), "borrowed_lanes_share": float(fb[1][1].sum() / 2 / max(1, (n // 2) * 40 * steps)),
            "sampled_envs_with_substeps_beyond_32_rows": int((fb[1].sum(0)[ids] > 0).sum()),
            "against": "oracle (fp64%s), re-synchronised from the device state before every control step; live = oracle reward != 0" % (", DM-physics v2 with the device's manifolds" if physics == 2 else "")}


def closed_loop(envs, stream_handles, dev, steps, tables=None, parity_envs=0, parity_steps=5, physics=1):
    """policy(g) -> control step(g) on each group's stream, `steps` times; one Policy object per group (its activation buffers are per object).
    parity_envs > 0: behind the timed loop, `parity_steps` more steps of the same loop with sampled envs checked against the oracle fed the policy's actions
    (`closed_loop.parity`: the learner's entry -- explicit actions on the state distribution the policy makes, fallback pairs included)."""
    import torch
    from deepmimic_amd.policy import Policy, random_weights
    env, n = envs.envs[0], envs.N
    offs = env.offsets_scales()
    w = random_weights(env.S, env.A, seed=0)
    w["s_mean"] = -offs["state_offset"].astype(np.float32); w["s_std"] = (1.0 / offs["state_scale"]).astype(np.float32)
    w["a_mean"] = -offs["action_offset"].astype(np.float32); w["a_std"] = (1.0 / offs["action_scale"]).astype(np.float32)
    pols = [Policy(w, device_id=dev.index or 0) for _ in range(envs.G)]
    f32, i32 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.int32, device=dev)
    st = torch.zeros((n, env.S), **f32); ac = torch.zeros((n, env.A), **f32); rw = torch.zeros(n, **f32)
    tm = torch.zeros(n, **i32); vd = torch.zeros(n, **i32); en = torch.zeros(n, **i32)
    ptrs = (st.data_ptr(), rw.data_ptr(), tm.data_ptr(), vd.data_ptr(), en.data_ptr())
    torch.cuda.synchronize()
    for g in range(envs.G):
        envs.step_group_device(g, 0, *ptrs, n_updates=0)            # RecordState of wherever the rollout left the envs

    def act(k, g):
        o = envs.start[g]
        pols[g].forward_device(st.data_ptr() + 4 * o * env.S, envs.count[g], ac.data_ptr() + 4 * o * env.A, 0, sample=True, seed=1, step=k,
                               env_id_offset=envs.first[g], stream=stream_handles[g])

    def loop(k0, k1):
        for k in range(k0, k1):
            for g in range(envs.G):
                act(k, g)
                envs.step_group_device(g, ac.data_ptr(), *ptrs, timestep=1.0 / 600, n_updates=20, auto_reset=True)

    warm = 60                  # two policy-driven episode lengths: the mixture of standing / tumbling / freshly reset characters the policy produces, not the transient behind the open-loop rollout
    loop(0, warm)
    envs.synchronize(); torch.cuda.synchronize()
    fb0 = np.array([envs.debug("fallback").sum(), envs.debug("borrowed").sum()])
    t0 = time.perf_counter()
    loop(warm, warm + steps)
    envs.synchronize(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    fb1 = np.array([envs.debug("fallback").sum(), envs.debug("borrowed").sum()])
    res = {"value": n * steps / dt, "unit": "env-steps/s", "steps": steps, "ms_per_step": 1e3 * dt / steps, "groups": envs.G,
           "policy": "%d -> 1024 -> 512 -> %d, random init, sampled (dm_policy_forward: one launch per group and step, k_policy_fused, bf16 MFMA)" % (env.S, env.A), "warmup_steps": warm,
           "mean_reward": float(rw.mean().item()), "finite": bool(torch.isfinite(st).all().item()),
           # share of the pair-substeps of the timed loop with a character beyond 32 constraint rows: on the 64-lane fallback of the two-per-wave kernel | on borrowed lanes
           "fallback_share": float((fb1 - fb0)[0] / 2 / max(1, (n // 2) * 40 * steps)), "borrowed_lanes_share": float((fb1 - fb0)[1] / 2 / max(1, (n // 2) * 40 * steps))}
    if parity_envs > 0 and tables is not None:
        try:
            def actions_of(k, st0):
                for g in range(envs.G):
                    act(warm + steps + k, g)
                envs.synchronize(); torch.cuda.synchronize()
                return ac.cpu().numpy()

            def step_and_read(acts):                                  # (`acts` are already in `ac` on the device)
                for g in range(envs.G):
                    envs.step_group_device(g, ac.data_ptr(), *ptrs, timestep=1.0 / 600, n_updates=20, auto_reset=True)
                envs.synchronize(); torch.cuda.synchronize()
                return {"state": st.cpu().numpy(), "reward": rw.cpu().numpy(), "terminate": tm.cpu().numpy(), "valid": vd.cpu().numpy(), "episode_end": en.cpu().numpy()}
            res["parity"] = parity_check(tables, envs, step_and_read, parity_envs, parity_steps, physics=physics, actions=actions_of)
        except Exception as ex:                                       # noqa: BLE001
            res["parity"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    for p in pols:
        p.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--scene", default="humanoid3d_walk")
    ap.add_argument("--precision", type=int, default=32)
    ap.add_argument("--wave-packing", type=int, default=0, help="characters per wavefront of the step kernel: 0 = the library's default (2 for the biped class incl. dribble_amp under DM-physics v1, 1 for the dog), 1 or 2")
    ap.add_argument("--physics", type=int, default=1, choices=[1, 2], help="contact model: 1 = DM-physics v1 (default, the headline), 2 = v2 (DESIGN.md 4.6; two characters per wavefront too since round 4)")
    ap.add_argument("--groups", type=int, default=0,
                    help="env groups per GPU: the rank's envs as G independent contexts on their own HIP streams (deepmimic_amd/groups.py; "
                         "0 = auto (2 from 4096 envs per GPU on: the half-batches drift apart in phase and fill each other's wave-time tail, humanoid +2.5..7 %%, dog3d +15 %%; 1 below); 1 = one launch per control step")
    ap.add_argument("--sustain-seconds", type=float, default=8.0,
                    help="after the timed --steps region, run back-to-back control steps for at least this long and report that rate too (`sustained`); 0 = skip")
    ap.add_argument("--solver-iters", type=int, default=0, help="measurement only: Gauss-Seidel iterations of the contact solver (0 = the library's 10); counter passes of two settings give the sweep's instruction mix by difference")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--force-gather", action="store_true", help="exercise the record exchange even with one rank")
    ap.add_argument("--gather", choices=["torch", "cabi"], default="torch",
                    help="record exchange through torch.distributed (default) or through the C-ABI (dm_comm_* / dm_gather_records: RCCL driven by libdm_hip.so)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl (= RCCL, the product) or gloo: CPU test harness of the N > 1 path (needs DM_HIP_LIB = the emulator build and DM_ALLOW_EMULATOR=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true", help="skip `checks.parity` (sampled envs of this run's contexts against the CPU oracle, after the timed regions)")
    ap.add_argument("--parity-envs", type=int, default=64)
    ap.add_argument("--parity-steps", type=int, default=5)
    ap.add_argument("--no-closed-loop", action="store_true", help="skip the `closed_loop` extra (policy on the matrix cores -> control step, N = 1 only)")
    ap.add_argument("--facade", action="store_true", help="measure the single-env cDeepMimicCore facade path instead of the batched env")
    ap.add_argument("--workers", type=int, nargs="*", default=[1], help="with --facade: aggregate rate of W worker processes (one cDeepMimicCore each) sharing the GPU, e.g. --workers 1 16 64")
    ap.add_argument("--shared-only", action="store_true", help="with --facade --workers: skip the one-context-per-worker runs for W > 1 (they take minutes at W = 256)")
    ap.add_argument("--cpu-baseline-worker", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-budget", type=float, default=12.0, help=argparse.SUPPRESS)
    ap.add_argument("--min-warmup", type=int, default=60,
                    help="control steps run before the timed region regardless of --warmup: the first episodes of a fresh batch are a "
                         "lighter mixture than steady state (nobody has fallen yet); two episode lengths reach it")
    args = ap.parse_args()
    if args.cpu_baseline_worker is not None:
        cpu_baseline_worker(args.scene, args.cpu_baseline_worker, args.cpu_budget)
        return
    if args.facade:
        facade_bench(args.scene, min(args.steps, 300), args.workers, private_too=not args.shared_only)
        return

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))

    import torch
    import torch.distributed as dist
    from deepmimic_amd import model
    from deepmimic_amd.core import BatchEnv

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d launched with WORLD_SIZE=%d: the launcher's rank count must equal --gpus" % (args.gpus, world))
    on_gpu = args.backend == "nccl"
    if on_gpu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit("rank %d has no GPU of its own (%d visible): one process per GPU" % (local_rank, torch.cuda.device_count()))
        torch.cuda.set_device(local_rank)          # rank r <-> GPU r
    elif os.environ.get("DM_ALLOW_EMULATOR") != "1":
        raise SystemExit("--backend gloo is the CPU test harness (emulator build): set DM_ALLOW_EMULATOR=1 and DM_HIP_LIB")
    dev = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")

    def dev_sync():
        if on_gpu:
            torch.cuda.synchronize()

    if world > 1 or (args.force_gather and "MASTER_ADDR" in os.environ):
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if on_gpu:
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    tables = model.load_asset(args.scene)
    n = args.envs
    from deepmimic_amd.groups import EnvGroups
    gather = (world > 1 or args.force_gather) and not args.no_gather
    # --groups 0 (default): two env groups on their own streams once the batch fills the chip's 2048 wave slots (>= 4096 envs): group A's next control step
    # starts when A's own slowest wave is done and backfills the slots B's finished waves left (humanoid +2.5 .. 7 %, dog3d -- two rounds of waves per
    # launch -- +15 %: profiles/r04_bench_env_sweep*.jsonl); below that size a second group only loses 1-4 %
    auto_groups = 2 if n >= 4096 else 1
    want_groups = auto_groups if args.groups <= 0 else args.groups
    envs = EnvGroups(tables, n, groups=want_groups, device_id=local_rank if on_gpu else 0, seed=1234, precision=args.precision, env_id_offset=rank * n,
                     test_mode=True, wave_packing=args.wave_packing, physics=args.physics, solver_iters=args.solver_iters)
    G = envs.G
    env = envs.envs[0]
    main_stream = torch.cuda.current_stream() if on_gpu else None
    # groups run on their contexts' OWN streams (created back to back by dm_create: they land on different hardware queues, measured; two
    # streams out of torch's pool shared one queue under the default GPU_MAX_HW_QUEUES = 4 and serialised: 1.23 M instead of 2.24 M,
    # tools/gpu_groups_modes.py), wrapped as torch external streams so that events order them against the collective's stream
    if on_gpu and G > 1:
        gstreams = [torch.cuda.ExternalStream(e.own_stream(), device=dev) for e in envs.envs]
    elif on_gpu:
        gstreams = [main_stream]; envs.set_streams([main_stream.cuda_stream])
    else:
        gstreams = [None] * G
    # deterministic per-env start phase keyed by the global env id; later episodes draw from the device's
    # counter-based generator, keyed by (seed, global env id, episode)
    from deepmimic_amd import streams
    envs.reset(kin_times=streams.reset_phase(rank * n + np.arange(n), env.duration))
    valid = torch.empty((n,), dtype=torch.int32, device=dev)
    ends = torch.empty((n,), dtype=torch.int32, device=dev)
    # per-env learner record {state[S], reward, terminate}: the step kernel of a group writes it straight into that group's flat exchange
    # buffer of slot k % 2; one RCCL all-gather per group and control step, issued asynchronously behind the group's kernel ON THE GROUP'S
    # STREAM, so that it overlaps the group's step k + 1 and nothing ever orders one group behind another (a joint gather of both groups' records
    # made group A's step k + 1 wait for B's step k and pulled the groups back into phase: measured 2.08 M instead of 2.21 M with one rank)
    from deepmimic_amd.dist import CabiRecordExchange, RecordExchange
    if args.gather == "cabi" and gather:
        # one dm_comm (its own RCCL communicator and stream) per env group: a group's gather is ordered against THAT group's ctx stream only (HIP events), so the
        # groups never wait for each other; the communicators are created in group order on every rank
        exs = [CabiRecordExchange(envs.envs[g], world, rank, dev, depth=2, force_rccl=True) for g in range(G)]
    else:
        exs = [RecordExchange(envs.count[g], env.S, world, dev, depth=2, env=env if G == 1 else None) for g in range(G)]
    tick = [0]
    import contextlib

    def on_stream(g):
        return torch.cuda.stream(gstreams[g]) if (on_gpu and G > 1) else contextlib.nullcontext()

    def one_step():
        slot = tick[0] & 1; tick[0] += 1
        for g in range(G):
            with on_stream(g):
                states, rewards, term = exs[g].begin(slot) if gather else exs[g].views(slot)
                r = envs.rows(g)
                envs.envs[g].step_device(0, states.data_ptr(), rewards.data_ptr(), term.data_ptr(), valid[r].data_ptr(), ends[r].data_ptr(),
                                         timestep=1.0 / 600, n_updates=20, auto_reset=True, open_loop=True)
                if gather:
                    exs[g].launch(slot)

    def drain():
        if gather:
            for g in range(G):
                with on_stream(g):
                    exs[g].wait(0); exs[g].wait(1)

    def timed(k):
        """k control steps between device-wide syncs: (wall seconds, [per-stream mean ms per launch from HIP events on the launch stream])"""
        dev_sync()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(G)] if on_gpu else []
        t0_ = time.perf_counter()
        for g in range(len(ev)):
            ev[g][0].record(gstreams[g])
        for _ in range(k):
            one_step()
        for g in range(len(ev)):
            ev[g][1].record(gstreams[g])
        drain()
        dev_sync()
        return time.perf_counter() - t0_, [a.elapsed_time(b) / k for a, b in ev]

    warm = max(args.warmup, args.min_warmup)
    for _ in range(warm):
        one_step()
    drain()
    dev_sync()
    if world > 1:
        dist.barrier()
    elapsed, launch_ms = timed(args.steps)
    if world > 1:
        dist.barrier()
    elapsed_local = elapsed
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # per-env parity of THIS run's contexts against the oracle, right behind the timed region: the envs are where warm-up + K timed steps left them, which is the same
    # state on every box for the same flags (deterministic resets, bit-reproducible kernels), so the sample -- and its result -- is reproducible.  Rank 0 compares; with
    # a record exchange every rank takes the same steps
    parity = None
    if not args.no_parity_check:
        taken = [0]

        def step_and_read():
            one_step(); taken[0] += 1; drain(); dev_sync()
            vs = [exs[g].views((tick[0] - 1) & 1) for g in range(len(exs))]
            return {"state": torch.cat([v[0] for v in vs]).cpu().numpy(), "reward": torch.cat([v[1] for v in vs]).cpu().numpy(),
                    "terminate": torch.cat([v[2] for v in vs]).cpu().numpy(), "valid": valid.cpu().numpy(), "episode_end": ends.cpu().numpy()}
        # every rank takes EXACTLY parity_steps steps (each is a collective with a record exchange), whatever happens to the comparison on rank 0: rank 0 builds the
        # oracle first and tells the others whether the check runs at all; if its comparison fails midway it takes the remaining steps before reporting
        go = 1
        if rank == 0:
            try:
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import oracle_lib
                oracle_lib.build("all")
            except Exception as ex:                                 # noqa: BLE001
                go = 0; parity = {"error": "oracle build failed: %s: %s" % (type(ex).__name__, ex)}
        if world > 1:
            gt = torch.tensor([go], dtype=torch.int32, device=dev)
            dist.broadcast(gt, src=0); go = int(gt.item())
        if go:
            try:
                if rank == 0:
                    parity = parity_check(tables, envs, step_and_read, args.parity_envs, args.parity_steps, physics=args.physics)
            except Exception as ex:                                 # noqa: BLE001  (reported, never silently dropped)
                parity = {"error": "%s: %s" % (type(ex).__name__, ex)}
            while taken[0] < args.parity_steps:                     # ranks > 0: all of them; rank 0: what an exception left
                step_and_read()

    # a sustained window behind the --steps region: >= sustain-seconds of back-to-back control steps (same step function, same exchange), so
    # that the line also carries a rate measured over seconds (DVFS settled, visible to an outside GPU-busy sampler) next to the short one
    sustained = None
    if on_gpu and args.sustain_seconds > 0:
        k2 = int(np.ceil(args.sustain_seconds / (elapsed / args.steps)))       # `elapsed` is the max over ranks: the same count on every rank
        el2, _ = timed(k2)
        if world > 1:
            tt = torch.tensor([el2], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el2 = float(tt.item())
        sustained = {"seconds": el2, "steps": k2, "value": world * n * k2 / el2, "unit": "env-steps/s", "ms_per_step": 1e3 * el2 / k2}

    # what the exchange costs on the critical path: the same K steps once more without it (per rank), so that a multi-GPU run is
    # interpretable -- value / N vs per_rank_no_gather tells exposed collective time from slow ranks
    local_rate = n * args.steps / elapsed_local
    exposed_ms = None
    if gather:
        gather_saved, gather = gather, False
        el_ng, _ = timed(args.steps)
        gather = gather_saved
        exposed_ms = 1e3 * (elapsed_local - el_ng) / args.steps
    per_rank = [local_rate]
    if world > 1:
        tl = torch.tensor([local_rate], dtype=torch.float64, device=dev)
        allr = [torch.zeros_like(tl) for _ in range(world)]
        dist.all_gather(allr, tl)
        per_rank = [float(x.item()) for x in allr]

    # average launch duration of the dominant kernel over the TIMED region: HIP events on each launch stream around its back-to-back
    # launches (no exchange: nothing else is on those streams); with G groups, G launches of n / G envs are in flight at once
    kernel_ms = float(np.mean(launch_ms)) if launch_ms else 0.0
    last = [exs[g].views((tick[0] - 1) & 1) for g in range(G)]
    mean_reward = float(torch.cat([v[1] for v in last]).mean().item())
    finite = all(bool(torch.isfinite(v[0]).all().item()) for v in last)

    # extra, N = 1: the same envs driven by the on-device policy (dm_policy.h: S -> 1024 -> 512 -> A on the MFMA units -- 227 / 28 for the humanoid --, random init, sampled
    # actions) instead of fixed actions -- what a sampler sees per control step.  Never `value`; a failure here is reported, not fatal.
    closed = None
    if on_gpu and world == 1 and not args.no_closed_loop and not gather:
        try:
            closed = closed_loop(envs, [s.cuda_stream for s in gstreams] if G > 1 else [0], dev, min(max(args.steps, 200), 1000),      # >= 200 steps whatever --steps: a 20-step window is a transient
                                 tables=tables, parity_envs=(0 if args.no_parity_check else args.parity_envs), parity_steps=args.parity_steps, physics=args.physics)
        except Exception as ex:                                     # noqa: BLE001
            closed = {"error": "%s: %s" % (type(ex).__name__, ex)}

    if rank == 0:
        # the line must describe the job that was asked for: N ranks, N per-rank rates
        if world != args.gpus or len(per_rank) != args.gpus:
            raise SystemExit("bench.py: ran %d rank(s) with %d per-rank rate(s) under --gpus %d" % (world, len(per_rank), args.gpus))
        envs_per_launch = n // G
        bytes_per_launch = algorithmic_bytes_per_env_step(env) * envs_per_launch
        kname = "k_env_step_duo" if (args.wave_packing != 1 and env.J <= 15 and env.D == 34 and envs_per_launch % 2 == 0 and not (tables.goal_kind == 5 and args.physics == 2)) else "k_env_step"      # (dribble_amp: two per wave under DM-physics v1 since round 6)
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else None      # (the emulator has no HIP events)
        # committed counter passes are taken with --groups 1 (a PMC pass serialises the kernels: a half-batch launch alone on the chip would be
        # another regime); HBM bytes are per env, so the per-launch figure of a group is the whole-batch one scaled by its share of the envs
        traffic, traffic_source = measured_traffic(args.scene, n, kname, physics=args.physics)
        if traffic is not None:
            traffic = traffic * envs_per_launch / n
        value = world * n * args.steps / elapsed
        valu_obj = measured_valu(args.scene, n, kname, value, physics=args.physics) or {"binding": "VALU issue + dependent-chain latency (fp32 vector)", "source": []}
        if on_gpu:
            try:
                # kernel family of dm_kernels.cpp this workload launches (deepmimic_amd/csrc/Makefile KIDS): plain / AMP-or-perturbation / v2 instantiation of the
                # two-per-wave kernel, or of the one-per-wave kernel on the compiled humanoid3d / dog3d topology, or biped + free body
                v = 2 if args.physics == 2 else (1 if (env.amp_size > 0 or env.has_perturbs) else 0)
                if kname == "k_env_step_duo":
                    fam = 24 if tables.goal_kind == 5 else (0, 1, 22)[v]
                elif tables.goal_kind == 5:
                    fam = 9
                else:
                    fam = ((12, 13, 20) if env.J > 15 else (15, 16, 21))[v]
                lc = latency_ceiling(tables, args, local_rank, kname, fam, n)
                lc["frac"] = ((value / world) / lc["env_steps_per_s"]) if lc["env_steps_per_s"] else None
                valu_obj["latency_ceiling"] = lc
            except Exception as ex:                                 # noqa: BLE001
                valu_obj["latency_ceiling"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        if world > 1:
            workload = "%s, %d envs sharded %d x %d (one shard per GPU), fixed-action (open-loop mocap tracking) rollout, auto-reset, " \
                       "20 updates of 1/600 s x 2 substeps per step" % (args.scene, world * n, world, n)
        else:
            workload = "%s, %d envs per GPU%s, fixed-action (open-loop mocap tracking) rollout, auto-reset, " \
                       "20 updates of 1/600 s x 2 substeps per step" % (args.scene, n, (" as %d groups of %d on their own streams" % (G, envs_per_launch)) if G > 1 else "")
        out = {
            "metric": "env-steps/sec at N parallel envs (%s)" % args.scene,
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32" if args.precision == 32 else "f64", "data": "synthetic",
            "config": {"workload": workload, "envs_per_gpu": n, "envs_total": world * n, "backend": "rccl" if on_gpu else "gloo + CPU emulator (test harness, not a measurement)", "wave_packing": args.wave_packing, "physics": args.physics, "solver_iters": (args.solver_iters or 10), "warmup_steps_run": warm, "groups": G, "envs_per_launch": envs_per_launch, "parallelism": "env-shards x%d%s" % (world, " + async RCCL all-gather of the record, overlapped with the next step" if gather else "")},
            "sim_updates_per_s": value * 20,
            "per_rank_env_steps_per_s": per_rank, "record_exchange": {"backend": (args.gather if gather else None), "exposed_ms_per_step_rank0": exposed_ms},
            "roofline": {"bound": "valu-issue/latency", "bound_of_the_figures_below": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                         "traffic": traffic, "traffic_source": traffic_source,
                         # true: the committed counter file was taken on exactly the device sources this library was built from; false: a kernel has changed since
                         # (re-profile: tools/gpu_round_profile.sh); null: the file predates the stamp
                         "traffic_source_current": getattr(measured_traffic, "current", None),
                         "kernel": kname, "kernel_ms": kernel_ms,
                         "concurrent_launches": G, "achieved_all_streams": (achieved * G) if achieved else None,
                         "valu": valu_obj,
                         "algorithmic_bytes_per_env_step": algorithmic_bytes_per_env_step(env),
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "note": "`bound` names what binds (VALU issue + dependent-chain latency; `valu` holds those figures, `valu.latency_ceiling` the live lone-wave ceiling); achieved / peak / frac / traffic are the HBM figures of the bench contract.  VALU-issue / latency bound by construction (SURVEY 8d, DESIGN.md 6): the env record stays in LDS/VGPRs for the 20 updates "
                                 "of a control step, HBM sees 2.4 KB per env-step; `valu` holds the binding figures.  achieved / kernel_ms are PER LAUNCH "
                                 "(HIP events on each launch stream over the timed region); with `concurrent_launches` groups in flight the chip moves "
                                 "`achieved_all_streams`"},
            "sustained": (dict(sustained, ratio_to_value=sustained["value"] / value) if sustained else None),
            "checks": {"mean_reward": mean_reward, "finite": finite, "parity": parity},
            "closed_loop": closed,
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.scene, args.cpu_budget)
        else:
            out["cpu_baseline"] = {"value": None, "unit": "env-steps/s", "cores": 1, "kind": "port", "sample": "skipped (N>1 or --no-cpu-baseline)"}
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
