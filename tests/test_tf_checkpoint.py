"""deepmimic_amd/tf_checkpoint.py: the reference's policy checkpoints (tf.train.Saver V2: learning/tf_agent.py:26-48) without TensorFlow.  The index reader runs
on the reference's own shipped `.index` files (all of them; their `.data` blobs are not in the repository); tensors and the mapping into the device actor are
checked on checkpoints this file writes in the same format."""
import glob
import os
import struct

import numpy as np
import pytest

from deepmimic_amd import tf_checkpoint as tfc

REF_POLICIES = "/root/reference/data/policies"


def _pv(n):
    out = b""
    while True:
        c = n & 0x7f; n >>= 7
        if n:
            out += bytes([c | 0x80])
        else:
            return out + bytes([c])


def _entry_proto(arr, offset, crc):
    dt = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.int64): 9}[arr.dtype]
    shape = b"".join(b"\x12" + _pv(len(_pv(d)) + 1) + b"\x08" + _pv(d) for d in arr.shape)
    v = b"\x08" + _pv(dt) + b"\x12" + _pv(len(shape)) + shape
    if offset:
        v += b"\x20" + _pv(offset)
    return v + b"\x28" + _pv(arr.nbytes) + b"\x35" + struct.pack("<I", crc)


def _table_block(entries, restart_interval=16):
    out, restarts, prev = b"", [], b""
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(k), len(prev)) and k[shared] == prev[shared]:
                shared += 1
        out += _pv(shared) + _pv(len(k) - shared) + _pv(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    out += b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))
    return out


def write_checkpoint(prefix, tensors):
    """a one-shard V2 checkpoint in tf.train.Saver's layout (test infrastructure: the writer side of tensorflow/core/util/tensor_bundle)"""
    names = sorted(tensors)
    data, entries = b"", [(b"", b"\x08\x01\x1a\x02\x08\x01")]          # BundleHeaderProto: one shard, little endian, version {producer 1}
    for n in names:
        a = np.ascontiguousarray(tensors[n])
        raw = a.tobytes()
        entries.append((n.encode(), _entry_proto(a, len(data), tfc.masked_crc32c(raw))))
        data += raw
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(data)
    blocks, pos = b"", 0

    def put(block):
        nonlocal blocks, pos
        h = _pv(pos) + _pv(len(block))
        body = block + b"\x00"
        blocks += body + struct.pack("<I", tfc.masked_crc32c(body)); pos += len(body) + 4
        return h
    hd = put(_table_block(entries))
    hm = put(_table_block([]))
    hi = put(_table_block([(names[-1].encode() + b"\x00", hd)], 1))
    foot = hm + hi
    foot += b"\x00" * (40 - len(foot)) + struct.pack("<Q", tfc.TABLE_MAGIC)
    with open(prefix + ".index", "wb") as f:
        f.write(blocks + foot)


def _agent_tensors(S, A, G=0, seed=0):
    from deepmimic_amd.policy import random_weights
    w = random_weights(S + G, A, seed=seed)
    rng = np.random.default_rng(seed + 1)
    t = {"agent/main/actor/0/dense/kernel": w["w1"], "agent/main/actor/0/dense/bias": rng.normal(size=1024).astype(np.float32) * 0.1,
         "agent/main/actor/1/dense/kernel": w["w2"], "agent/main/actor/1/dense/bias": rng.normal(size=512).astype(np.float32) * 0.1,
         "agent/main/actor/dist_gauss_diag/mean/kernel": w["w3"], "agent/main/actor/dist_gauss_diag/mean/bias": rng.normal(size=A).astype(np.float32) * 0.01,
         "agent/main/actor/dist_gauss_diag/logstd/bias": w["logstd"],
         "agent/main/critic/0/dense/kernel": rng.normal(size=(S + G, 1024)).astype(np.float32),
         "agent/resource/s_norm/mean": rng.normal(size=S).astype(np.float32), "agent/resource/s_norm/std": (0.5 + rng.random(S)).astype(np.float32),
         "agent/resource/s_norm/count": np.array([12345], np.int32),
         "agent/resource/a_norm/mean": rng.normal(size=A).astype(np.float32) * 0.1, "agent/resource/a_norm/std": (0.5 + rng.random(A)).astype(np.float32),
         "agent/resource/g_norm/mean": rng.normal(size=G).astype(np.float32), "agent/resource/g_norm/std": (0.5 + rng.random(G)).astype(np.float32)}
    return t


@pytest.mark.skipif(not os.path.isdir(REF_POLICIES), reason="needs the reference checkout (data/policies/*.ckpt.index)")
def test_every_shipped_index_file_parses():
    files = sorted(glob.glob(os.path.join(REF_POLICIES, "*", "*.ckpt.index")))
    assert len(files) >= 40
    plain = gated = 0
    for f in files:
        idx = tfc.read_index(f)                       # (table block checksums are verified on the way)
        assert idx[""]["num_shards"] == 1
        ents = [(n, e) for n, e in idx.items() if n]
        # the Saver lays the tensors out back to back in name order: offsets chain, sizes are shape x itemsize
        pos = 0
        for n, e in ents:
            assert e["offset"] == pos and e["size"] == int(np.prod(e["shape"], dtype=np.int64)) * np.dtype(tfc.DTYPES[e["dtype"]]).itemsize, (f, n, e)
            pos += e["size"]
        k0, k1, km = (idx["agent/main/actor/%s" % s] for s in ("0/dense/kernel", "1/dense/kernel", "dist_gauss_diag/mean/kernel"))
        assert k0["shape"][1] == k1["shape"][0] == 1024 and k1["shape"][1] == km["shape"][0] == 512
        s_dim = idx["agent/resource/s_norm/mean"]["shape"][0]; g = idx["agent/resource/g_norm/mean"]["shape"]
        assert len(g) == 1 and k0["shape"][0] == s_dim + g[0]
        assert idx["agent/resource/a_norm/mean"]["shape"] == [km["shape"][1]] == idx["agent/main/actor/dist_gauss_diag/logstd/bias"]["shape"]
        if any(n.startswith("agent/main/actor/gate") for n, _ in ents):
            gated += 1
            with pytest.raises(NotImplementedError):
                tfc.actor_weights(f[:-len(".index")])
        else:
            plain += 1
            with pytest.raises(FileNotFoundError):      # the .data blob is not in the repository: said so, not a crash
                tfc.actor_weights(f[:-len(".index")])
    assert plain >= 20 and gated >= 5


def test_humanoid_and_dog_state_sizes_match_the_shipped_policies(emu_lib, monkeypatch):
    """the observation and action this repo's context records / takes have the sizes the reference's pretrained actors were trained on"""
    if not os.path.isdir(REF_POLICIES):
        pytest.skip("needs the reference checkout")
    monkeypatch.setenv("DM_ALLOW_EMULATOR", "1")
    from deepmimic_amd import model
    from deepmimic_amd.core import BatchEnv
    for asset, ck in (("humanoid3d_walk", "humanoid3d/humanoid3d_walk"), ("dog3d_pace", "dog3d/dog3d_pace")):
        idx = tfc.read_index(os.path.join(REF_POLICIES, ck + ".ckpt.index"))
        env = BatchEnv(model.load_asset(asset), 1, lib_path=emu_lib, precision=64)
        assert idx["agent/resource/s_norm/mean"]["shape"] == [env.S] and idx["agent/resource/a_norm/mean"]["shape"] == [env.A]
        env.close()


def test_round_trip_and_crc(tmp_path):
    t = _agent_tensors(50, 12, G=3)
    prefix = str(tmp_path / "agent0_model.ckpt")
    write_checkpoint(prefix, t)
    back = tfc.read_tensors(prefix, verify_below=None)
    assert sorted(back) == sorted(t)
    for n in t:
        assert back[n].dtype == t[n].dtype and np.array_equal(back[n], t[n]), n
    assert tfc.crc32c(b"123456789") == 0xE3069283                       # the CRC-32C check value
    raw = bytearray(open(prefix + ".data-00000-of-00001", "rb").read()); raw[100] ^= 1
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        tfc.read_tensors(prefix, verify_below=None)


def test_actor_weights_drive_the_device_actor(emu_lib, tmp_path, monkeypatch):
    """checkpoint -> deepmimic_amd.policy.Policy: the mode action of the kernels equals the numpy statement of learning/pg_agent.py's actor on the tensors written"""
    monkeypatch.setenv("DM_ALLOW_EMULATOR", "1")
    from deepmimic_amd.policy import Policy, reference_forward
    S, A = 197, 36
    t = _agent_tensors(S, A, seed=3)
    prefix = str(tmp_path / "policy.ckpt")
    write_checkpoint(prefix, t)
    w = tfc.actor_weights(prefix, state_dim=S)
    assert np.array_equal(w["w1"], t["agent/main/actor/0/dense/kernel"]) and np.array_equal(w["s_std"], t["agent/resource/s_norm/std"]) and "g_mean" not in w
    pol = Policy(w, lib_path=emu_lib)
    s = np.random.default_rng(5).normal(size=(32, S)).astype(np.float32)
    a, _ = pol.forward_host(s)
    ref, _ = reference_forward(w, s, bf16=True)
    assert np.abs(a - ref).max() < 2e-3 * max(1.0, np.abs(ref).max())
    with pytest.raises(ValueError, match="state features"):
        tfc.actor_weights(prefix, state_dim=S + 1)
