"""bench.py's own N > 1 path on CPU: `python bench.py --gpus 2` must launch 2 ranks itself and print a line that says so.

Runs bench.py as the driver would (no WORLD_SIZE in the environment), on the test harness backend: gloo + the emulator build of
the kernels.  The reference's precedent for a self-launching driver is mpi_run.py:16-24."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(emu_lib, extra, env_extra=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DM_HIP_LIB=emu_lib, DM_ALLOW_EMULATOR="1")
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--backend", "gloo", "--envs", "4", "--steps", "2", "--warmup", "1",
                        "--min-warmup", "1", "--no-cpu-baseline", "--precision", "64"] + extra,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def test_bench_gpus_2_launches_two_ranks(emu_lib):
    p, line = run_bench(emu_lib, ["--gpus", "2", "--groups", "2"])      # (auto picks two groups from 4096 envs per rank on: asked for here, at 4 envs)
    assert p.returncode == 0, p.stderr[-2000:]
    assert line is not None, p.stdout[-2000:]
    assert line["n_gpus"] == 2
    assert len(line["per_rank_env_steps_per_s"]) == 2 and all(r > 0 for r in line["per_rank_env_steps_per_s"])
    assert line["scaling"] == "weak" and line["config"]["envs_total"] == 8 and line["config"]["envs_per_gpu"] == 4
    assert "8 envs sharded 2 x 4" in line["config"]["workload"]
    assert line["record_exchange"]["backend"] == "torch"           # the all-gather of the learner record ran on both ranks
    assert line["value"] > 0 and line["checks"]["finite"] and 0.0 < line["checks"]["mean_reward"] <= 1.0
    assert len([l for l in p.stdout.splitlines() if l.startswith("{")]) == 1     # rank 0 prints ONE line
    assert line["config"]["groups"] == 2 and line["config"]["envs_per_launch"] == 2      # two env groups per rank (deepmimic_amd/groups.py)


def test_bench_gpus_8_launches_eight_ranks(emu_lib):
    """the launch path the driver takes on an 8-GPU node: 8 ranks, 8 per-rank rates, one line, the exchange on every rank"""
    p, line = run_bench(emu_lib, ["--gpus", "8"], timeout=1500)
    assert p.returncode == 0, p.stderr[-2000:]
    assert line is not None and line["n_gpus"] == 8 and len(line["per_rank_env_steps_per_s"]) == 8
    assert line["config"]["envs_total"] == 32 and "32 envs sharded 8 x 4" in line["config"]["workload"]
    assert line["record_exchange"]["backend"] == "torch" and line["checks"]["finite"]
    assert len([l for l in p.stdout.splitlines() if l.startswith("{")]) == 1


def test_bench_groups_1_is_one_launch_per_step(emu_lib):
    p, line = run_bench(emu_lib, ["--gpus", "1", "--groups", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert line["config"]["groups"] == 1 and line["config"]["envs_per_launch"] == 4 and line["roofline"]["concurrent_launches"] == 1


def test_bench_gpus_1_single_process(emu_lib):
    p, line = run_bench(emu_lib, ["--gpus", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert line["n_gpus"] == 1 and len(line["per_rank_env_steps_per_s"]) == 1
    assert "4 envs per GPU" in line["config"]["workload"]
    assert "traffic_source" in line["roofline"]


def test_bench_refuses_rank_count_mismatch(emu_lib):
    # an external launcher that started ONE rank for --gpus 2 must not yield an N = 1 number under an N = 2 command
    p, line = run_bench(emu_lib, ["--gpus", "2"], env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and line is None
    assert "must equal --gpus" in p.stderr


def test_bench_physics_2_and_dog(emu_lib):
    # the measurement switches of round 3: --physics 2 (DM-physics v2) and the dog on its compiled topology
    p, line = run_bench(emu_lib, ["--gpus", "1", "--physics", "2"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert line["config"]["physics"] == 2 and line["roofline"]["kernel"] == "k_env_step_duo" and line["checks"]["finite"]      # (two per wavefront since round 4)
    p, line = run_bench(emu_lib, ["--gpus", "1", "--physics", "2", "--wave-packing", "1"])
    assert p.returncode == 0 and line["roofline"]["kernel"] == "k_env_step"
    p, line = run_bench(emu_lib, ["--gpus", "1", "--scene", "dog3d_pace"])
    assert p.returncode == 0, p.stderr[-2000:]
    assert line["roofline"]["kernel"] == "k_env_step" and line["roofline"]["algorithmic_bytes_per_env_step"] > 2420 and line["checks"]["finite"]


import pytest  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("gather", ["cabi", "torch"])
def test_bench_record_exchange_is_hidden_gpu(hip_lib, gather):
    """`bench.py --gpus 1 --force-gather --gather cabi|torch` drives the record exchange of the multi-GPU path with one rank through real
    RCCL -- the C-ABI route (dm_comm_* / dm_gather_records, one dm_comm per env group) and the torch.distributed route (one all_gather_into_tensor per group and
    step on one process group) -- double-buffered behind the step kernels of the two env groups; what it leaves exposed on the critical path must stay below 50 us
    per control step.  With this green, `bench.py --gpus N` on an 8-GPU node is the same code with N ranks."""
    import socket
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "DM_HIP_LIB", "DM_ALLOW_EMULATOR")}
    if gather == "torch":         # (a process group of one rank: bench.py initialises it when the rendezvous variables are there)
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "200", "--warmup", "10", "--force-gather", "--gather", gather,
                        "--no-cpu-baseline", "--sustain-seconds", "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["record_exchange"]["backend"] == gather and line["config"]["groups"] == 2      # (round 5: one exchange per env group on either route)
    assert line["record_exchange"]["exposed_ms_per_step_rank0"] < 0.05, line["record_exchange"]
    assert line["value"] > 1.0e6 and line["checks"]["finite"]
    assert line["checks"]["parity"]["flags_equal"] and line["checks"]["parity"]["reward_mae"] < 1e-5      # the envs behind the exchange buffers are the envs of the oracle


@pytest.mark.gpu
def test_bench_default_line_gpu(hip_lib):
    """the driver's command: the line carries the two env groups, the binding bound (roofline.valu) and the sustained window"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "DM_HIP_LIB", "DM_ALLOW_EMULATOR")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["config"]["groups"] == 2 and line["roofline"]["concurrent_launches"] == 2 and line["roofline"]["kernel"] == "k_env_step_duo"
    assert line["value"] > 1.0e6 and line["sustained"]["seconds"] >= 7.5 and 0.85 < line["sustained"]["ratio_to_value"] < 1.25      # (BASELINE's target; boxes of the pool differ by up to 40 %)
    v = line["roofline"]["valu"]
    assert v and 0.3 < v["valu_busy"] < 1.0 and v["source"] and 0.01 < v["frac_of_fp32_peak"] < 1.0
    assert abs(line["ms_per_step"] * line["value"] / 1e3 - 4096) < 1.0
    # round 6: the closed-loop leg carries its own parity figures (explicit actions from the on-device policy, oracle fed the same actions); the ceiling is a ceiling
    cp = line["closed_loop"]["parity"]
    assert cp["flags_equal"] and cp["reward_mae"] < 3e-5 and cp["live"] > 100, cp
    assert line["closed_loop"]["value"] > 0.9 * line["value"]
    assert 0.5 < v["latency_ceiling"]["frac"] <= 1.0 and v["latency_ceiling"]["rounds_of_waves"] == 1
    assert line["roofline"]["bound"] == "valu-issue/latency" and line["roofline"]["bound_of_the_figures_below"] == "hbm"
    lc = v["latency_ceiling"]                                   # live: a control step of waves that are alone on their SIMDs -> the ceiling of this kernel shape
    assert "error" not in lc and lc["wave_slots"] == 2048 and lc["envs_per_wave"] == 2 and 0.5 < lc["frac"] < 1.0 and 1.0 < lc["lone_wave_ms_per_step"] < 2.5, lc
    par = line["checks"]["parity"]                              # sampled envs of THIS run's contexts against the oracle, right behind the timed region
    assert par and "error" not in par and par["envs"] == 64 and par["steps"] == 5 and par["flags_equal"] is True and par["live"] > 200, par
    assert par["reward_mae"] < 1e-5 and par["reward_max_not_live"] < 1e-6, par
    c = line["closed_loop"]                                     # extra: the same envs driven by the on-device policy
    assert c and "error" not in c, c
    assert c["groups"] == 2 and c["finite"] and c["value"] > 0.8e6 and 0.5 < c["value"] / line["value"] < 1.1, c
