"""N > 1 path on CPU: world_size-2 gloo run of the env shards (emulator build of the kernels as the device)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["DM_ROOT"])
import torch, torch.distributed as dist
from deepmimic_amd import model
from deepmimic_amd.dist import ShardedEnv, shard_range
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
total = int(os.environ["DM_TOTAL"])
t = model.load_asset("humanoid3d_walk")
sh = ShardedEnv(t, total, rank=rank, world=world, device_id=0, seed=11, precision=64, lib_path=os.environ["DM_HIP_LIB"], wave_packing=1)
sh.env.reset()
recs = []
for k in range(2):
    out = sh.env.step(None, 1.0 / 600, 20, open_loop=True, auto_reset=True)
    recs.append(sh.gather(sh.pack_record(out)).numpy().copy())      # the gathered tensor is reused by the next gather (uneven shards)
if rank == 0:
    np.save(os.environ["DM_OUT"], np.stack(recs))
# double-buffered flat record exchange (equal shards): what bench.py runs per control step
from deepmimic_amd.dist import RecordExchange
n = 2
eq = ShardedEnv(t, 2 * n, rank=rank, world=world, device_id=0, seed=11, precision=64, lib_path=os.environ["DM_HIP_LIB"], wave_packing=1)
eq.env.reset()
ex = RecordExchange(n, eq.S, world, "cpu", depth=2)
got = []
for k in range(3):
    slot = k & 1
    st, rw, tm = ex.begin(slot)
    out = eq.env.step(None, 1.0 / 600, 20, open_loop=True, auto_reset=True)
    st.copy_(torch.from_numpy(out["state"])); rw.copy_(torch.from_numpy(out["reward"])); tm.copy_(torch.from_numpy(out["terminate"]))
    ex.launch(slot)
    if k >= 1:                                   # consume step k-1 while step k's gather is in flight
        S_, R_, T_ = ex.result((k - 1) & 1)
        got.append(np.concatenate([S_.reshape(world * n, -1).numpy(), R_.reshape(-1, 1).numpy(), T_.reshape(-1, 1).numpy().astype(np.float32)], 1))
if rank == 1:
    np.save(os.environ["DM_OUT"] + ".ex.npy", np.stack(got))
dist.barrier()
dist.destroy_process_group()
'''


def test_shard_range_partitions():
    from deepmimic_amd.dist import shard_range
    for total in (1, 7, 8, 4096, 32768 + 3):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == total
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1


def test_two_rank_gloo_matches_single_process(emu_lib, tmp_path):
    total = 5                                  # uneven split: rank 0 gets 3 envs, rank 1 gets 2
    out = str(tmp_path / "rec.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, DM_ROOT=ROOT, DM_HIP_LIB=emu_lib, DM_TOTAL=str(total), DM_OUT=out, MASTER_ADDR="127.0.0.1")
    port = 29500 + (os.getpid() % 2000)
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)], env=env, timeout=600)
    got = np.load(out)
    # single process, all envs in one shard: trajectories must not depend on the partition (one character per wave on both
    # sides: the two-per-wave kernel sums in a different order, so shards of different parity would differ in the last bits)
    from deepmimic_amd import model
    from deepmimic_amd.dist import ShardedEnv
    sh = ShardedEnv(model.load_asset("humanoid3d_walk"), total, rank=0, world=1, device_id=0, seed=11, precision=64, lib_path=emu_lib, wave_packing=1)
    sh.env.reset()
    for k in range(2):
        o = sh.env.step(None, 1.0 / 600, 20, open_loop=True, auto_reset=True)
        assert np.array_equal(got[k], sh.pack_record(o))
    # RecordExchange: 2 ranks x 2 envs, steps 0 and 1 consumed one step late, in global env order
    ex = np.load(out + ".ex.npy")
    sh = ShardedEnv(model.load_asset("humanoid3d_walk"), 4, rank=0, world=1, device_id=0, seed=11, precision=64, lib_path=emu_lib, wave_packing=1)
    sh.env.reset()
    for k in range(2):
        o = sh.env.step(None, 1.0 / 600, 20, open_loop=True, auto_reset=True)
        assert np.array_equal(ex[k], sh.pack_record(o))


def test_cabi_record_exchange_single_rank(emu_lib):
    """dm_comm_create / dm_gather_records / dm_gather_wait (include/dm_hip.h) behind dist.CabiRecordExchange, one rank, emulator
    build: the exchange must hand back exactly what the step wrote, double-buffered."""
    import torch
    from deepmimic_amd import model
    from deepmimic_amd.core import BatchEnv
    from deepmimic_amd.dist import CabiRecordExchange
    os.environ["DM_HIP_LIB"] = emu_lib
    try:
        t = model.load_asset("humanoid3d_walk")
        env = BatchEnv(t, 2, precision=64, lib_path=emu_lib, wave_packing=1, seed=3)
        env.reset()
        ex = CabiRecordExchange(env, 1, 0, "cpu", depth=2)
        valid = torch.zeros(2, dtype=torch.int32); ends = torch.zeros(2, dtype=torch.int32)
        for k in range(3):
            slot = k & 1
            st, rw, tm = ex.begin(slot)
            env.step_device(0, st.data_ptr(), rw.data_ptr(), tm.data_ptr(), valid.data_ptr(), ends.data_ptr(), auto_reset=True, open_loop=True)
            ex.launch(slot)
            S_, R_, T_ = ex.result(slot)
            assert torch.equal(S_[0], st) and torch.equal(R_[0], rw) and torch.equal(T_[0], tm)
            assert float(rw.min()) > 0
    finally:
        del os.environ["DM_HIP_LIB"]


NORM_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["DM_ROOT"])
import torch, torch.distributed as dist
from deepmimic_amd.normalizer import DeviceNormalizer
from deepmimic_amd.dist import all_reduce_normalizer
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
S = 11
gids = np.array([0, 0, 1, 1, 1, -1, 0, 2, 2, 0, 0], np.int32)
nrm = DeviceNormalizer(S, gids, eps=0.02, lib_path=os.environ["DM_HIP_LIB"])
nrm.set_mean_std(np.linspace(-1, 1, S), np.linspace(0.5, 2, S))
for it in range(3):
    rng = np.random.default_rng(100 * it + rank)               # every rank records its OWN rollouts (different sizes too)
    x = (rng.normal(size=(5 + 3 * rank + it, S)) * 2 + 0.3).astype(np.float32)
    nrm.record(x)
    all_reduce_normalizer(nrm, "cpu")
    nrm.update()
np.savez(os.environ["DM_OUT"] + ".norm%d.npz" % rank, mean=nrm.mean, std=nrm.std, mean_sq=nrm.mean_sq, count=np.array([nrm.count]))
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_normalizer_reduce_sum(emu_lib, tmp_path):
    """learning/normalizer.py:47-73 over two workers: pending sums all-reduced (gloo here, RCCL on GPUs), then the same update on both ranks --
    identical statistics on both (check_synced), equal to the oracle fed with both ranks' records"""
    import sys as _sys
    _sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from normalizer_oracle import NormalizerOracle
    out = str(tmp_path / "n")
    script = tmp_path / "norm_worker.py"
    script.write_text(NORM_WORKER)
    env = dict(os.environ, DM_ROOT=ROOT, DM_HIP_LIB=emu_lib, DM_OUT=out, MASTER_ADDR="127.0.0.1", DM_ALLOW_EMULATOR="1")
    port = 31500 + (os.getpid() % 2000)
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)], env=env, timeout=600)
    a, b = np.load(out + ".norm0.npz"), np.load(out + ".norm1.npz")
    for k in ("mean", "std", "mean_sq", "count"):
        assert np.array_equal(a[k], b[k]), k                       # bit-identical on both ranks
    S = 11
    ora = NormalizerOracle(S, np.array([0, 0, 1, 1, 1, -1, 0, 2, 2, 0, 0]), 0.02)
    ora.set_mean_std(np.linspace(-1, 1, S), np.linspace(0.5, 2, S))
    for it in range(3):
        for rank in range(2):
            rng = np.random.default_rng(100 * it + rank)
            ora.record((rng.normal(size=(5 + 3 * rank + it, S)) * 2 + 0.3).astype(np.float32))
        ora.update()
    assert int(a["count"][0]) == ora.count
    assert np.allclose(a["mean"], ora.mean, rtol=1e-12, atol=1e-13) and np.allclose(a["std"], ora.std, rtol=1e-12, atol=1e-13)


GROUPS_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.environ["DM_ROOT"])
import torch, torch.distributed as dist
from deepmimic_amd import model
from deepmimic_amd.groups import EnvGroups
from deepmimic_amd.dist import RecordExchange
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
t = model.load_asset("humanoid3d_walk")
n, G = 4, 2
envs = EnvGroups(t, n, groups=G, env_id_offset=rank * n, seed=11, precision=64, lib_path=os.environ["DM_HIP_LIB"], wave_packing=1)
assert envs.G == G
envs.reset()
exs = [RecordExchange(envs.count[g], envs.S, world, "cpu", depth=2) for g in range(G)]
got = []
for k in range(4):
    slot = k & 1
    for g in range(G):                       # bench.py's order: per group begin -> step -> launch; two gathers per control step on ONE process group
        st, rw, tm = exs[g].begin(slot)
        out = envs.envs[g].step(None, 1.0 / 600, 20, open_loop=True, auto_reset=True)
        st.copy_(torch.from_numpy(out["state"])); rw.copy_(torch.from_numpy(out["reward"])); tm.copy_(torch.from_numpy(out["terminate"]))
        exs[g].launch(slot)
    if k >= 1:
        rows = np.zeros((world * n, envs.S + 2), np.float32)
        for g in range(G):
            S_, R_, T_ = exs[g].result((k - 1) & 1)
            for r in range(world):           # group g of rank r holds the global envs [r n + start_g, r n + start_g + count_g)
                lo = r * n + envs.start[g]
                rows[lo:lo + envs.count[g], :envs.S] = S_[r].numpy(); rows[lo:lo + envs.count[g], envs.S] = R_[r].numpy(); rows[lo:lo + envs.count[g], envs.S + 1] = T_[r].numpy()
        got.append(rows)
np.save(os.environ["DM_OUT"] + ".g%d.npy" % rank, np.stack(got))
dist.barrier()
dist.destroy_process_group()
'''


def test_two_ranks_two_groups_share_one_process_group(emu_lib, tmp_path):
    """bench.py with env groups issues TWO all-gathers per control step (one per group) on ONE communicator.  That is correct because every rank issues them in
    the same order (group 0, then group 1) and a process group runs its collectives in issue order (nccl: on its own stream; gloo: synchronously) -- the
    assumption DESIGN.md section 8 states.  Here: 2 ranks x 2 groups, records consumed one step late; every rank must see every env's record, in global env
    order, equal to the single-process rollout."""
    out = str(tmp_path / "grp")
    script = tmp_path / "groups_worker.py"
    script.write_text(GROUPS_WORKER)
    env = dict(os.environ, DM_ROOT=ROOT, DM_HIP_LIB=emu_lib, DM_OUT=out, MASTER_ADDR="127.0.0.1", DM_ALLOW_EMULATOR="1")
    port = 33500 + (os.getpid() % 2000)
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)], env=env, timeout=600)
    a, b = np.load(out + ".g0.npy"), np.load(out + ".g1.npy")
    assert np.array_equal(a, b)
    from deepmimic_amd import model
    from deepmimic_amd.dist import ShardedEnv
    sh = ShardedEnv(model.load_asset("humanoid3d_walk"), 8, rank=0, world=1, device_id=0, seed=11, precision=64, lib_path=emu_lib, wave_packing=1)
    sh.env.reset()
    for k in range(3):
        o = sh.env.step(None, 1.0 / 600, 20, open_loop=True, auto_reset=True)
        assert np.array_equal(a[k], sh.pack_record(o)), k
