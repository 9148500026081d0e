"""Reader for the reference's policy checkpoints (`--model_files data/policies/...ckpt`: learning/rl_world.py:67-85 -> learning/tf_agent.py:36-48, a
`tf.train.Saver` V2 checkpoint) into the on-device actor of `deepmimic_amd.policy` -- without TensorFlow.

A V2 checkpoint is two files.  `<prefix>.index` is an uncompressed leveldb-format sorted table (blocks of prefix-compressed keys with restart arrays, a 5-byte
block trailer, a 48-byte footer ending in the magic 0xdb4775248b80fb57) whose keys are variable names and whose values are `BundleEntryProto` messages
(dtype, shape, shard, offset, size, masked crc32c); the empty key holds the `BundleHeaderProto`.  `<prefix>.data-0000k-of-0000n` holds the tensors raw,
little-endian, at those offsets.  The reference checkout ships the 52 `.index` files of its pretrained policies (the `.data` blobs are not in the repository):
`read_index` is tested on all of them, `read_tensors` / `actor_weights` on checkpoints written by the test suite's own writer in the same format.

Variables of an agent (learning/pg_agent.py:141-188, learning/nets/fc_2layers_1024units.py, learning/rl_agent.py normalizers), scope `agent`:
    agent/main/actor/0/dense/{kernel,bias}  [S + G, 1024]      agent/main/actor/1/dense/{kernel,bias}  [1024, 512]
    agent/main/actor/dist_gauss_diag/mean/{kernel,bias}  [512, A]      agent/main/actor/dist_gauss_diag/logstd/bias  [A]
    agent/resource/{s_norm,g_norm,a_norm}/{mean,std}
The AMP task policies use the gated net (learning/nets/fc_2layers_gated_1024units.py: `gate0`, `gate1`, `gate_common` variables); the device actor is the
plain two-layer net, so those are refused with the list of what was found."""
import os
import struct
from typing import Dict, Optional

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64}       # tensorflow/core/framework/types.proto: DT_FLOAT, DT_DOUBLE, DT_INT32, DT_INT64


def _varint(b: bytes, i: int):
    r = s = 0
    while True:
        c = b[i]; i += 1
        r |= (c & 0x7f) << s; s += 7
        if c < 0x80:
            return r, i


def _block(b: bytes, off: int, size: int):
    """entries of one table block: shared-prefix length, unshared length, value length (varints), key suffix, value; restart array + count at the end"""
    if b[off + size] != 0:
        raise ValueError("compressed table block (type %d): tf.train.Saver writes the index uncompressed" % b[off + size])
    if masked_crc32c(b[off:off + size + 1]) != struct.unpack("<I", b[off + size + 1:off + size + 5])[0]:
        raise ValueError("table block at %d: checksum mismatch" % off)
    blk = b[off:off + size]
    n_restarts = struct.unpack("<I", blk[-4:])[0]
    end = len(blk) - 4 - 4 * n_restarts
    i, key, out = 0, b"", []
    while i < end:
        shared, i = _varint(blk, i); unshared, i = _varint(blk, i); vlen, i = _varint(blk, i)
        key = key[:shared] + blk[i:i + unshared]; i += unshared
        out.append((key, blk[i:i + vlen])); i += vlen
    return out


def _handle(b: bytes, i: int):
    off, i = _varint(b, i); size, i = _varint(b, i)
    return (off, size), i


def _entry(v: bytes) -> dict:
    """BundleEntryProto (tensorflow/core/protobuf/tensor_bundle.proto): 1 dtype, 2 shape {2 dim {1 size}}, 3 shard_id, 4 offset, 5 size, 6 crc32c (fixed32)"""
    e = {"dtype": 0, "shape": [], "shard": 0, "offset": 0, "size": 0, "crc32c": None}
    i = 0
    while i < len(v):
        tag, i = _varint(v, i)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            x, i = _varint(v, i)
        elif wire == 5:
            x = struct.unpack("<I", v[i:i + 4])[0]; i += 4
        elif wire == 2:
            n, i = _varint(v, i); x = v[i:i + n]; i += n
        elif wire == 1:
            x = v[i:i + 8]; i += 8
        else:
            raise ValueError("unsupported protobuf wire type %d in a bundle entry" % wire)
        if field == 1:
            e["dtype"] = x
        elif field == 2:
            j = 0
            while j < len(x):
                t2, j = _varint(x, j)
                if t2 & 7 != 2:
                    _, j = _varint(x, j); continue          # unknown_rank and the like
                n2, j = _varint(x, j); dim = x[j:j + n2]; j += n2
                if t2 >> 3 == 2:
                    k, dsize = 0, 0                             # (proto3 omits a zero: an empty dim message is a dimension of size 0 -- g_norm of an agent without a goal)
                    while k < len(dim):
                        t3, k = _varint(dim, k)
                        if t3 & 7 == 0:
                            val, k = _varint(dim, k)
                            if t3 >> 3 == 1:
                                dsize = val
                        else:
                            n3, k = _varint(dim, k); k += n3    # dim name
                    e["shape"].append(dsize)
        elif field == 3:
            e["shard"] = x
        elif field == 4:
            e["offset"] = x
        elif field == 5:
            e["size"] = x
        elif field == 6:
            e["crc32c"] = x
    return e


def read_index(path: str) -> Dict[str, dict]:
    """`<prefix>.index` -> {variable name: {dtype, shape, shard, offset, size, crc32c}}; key "" holds {"num_shards": n}"""
    b = open(path, "rb").read()
    if len(b) < 48 or struct.unpack("<Q", b[-8:])[0] != TABLE_MAGIC:
        raise ValueError("%s is not a tensor-bundle index (table magic missing)" % path)
    foot = b[-48:]
    _, i = _handle(foot, 0)                    # metaindex block (empty)
    (ioff, isize), _ = _handle(foot, i)
    out = {}
    for _, hv in _block(b, ioff, isize):       # index block: one handle per data block
        (doff, dsize), _ = _handle(hv, 0)
        for key, val in _block(b, doff, dsize):
            if key == b"":                      # BundleHeaderProto: 1 num_shards, 2 endianness (0 little), 3 version
                hdr, i2 = {"num_shards": 1, "endianness": 0}, 0
                while i2 < len(val):
                    tag, i2 = _varint(val, i2)
                    if tag & 7 == 0:
                        x, i2 = _varint(val, i2)
                        if tag >> 3 == 1:
                            hdr["num_shards"] = x
                        elif tag >> 3 == 2:
                            hdr["endianness"] = x
                    else:
                        n, i2 = _varint(val, i2); i2 += n
                if hdr["endianness"] != 0:
                    raise ValueError("big-endian tensor bundle")
                out[""] = hdr
            else:
                out[key.decode()] = _entry(val)
    return out


_CRC_TABLE = None


def crc32c(data: bytes) -> int:
    """CRC-32C (Castagnoli), the checksum of the table blocks and of every tensor"""
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = []
        for n in range(256):
            c = n
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t.append(c)
        _CRC_TABLE = t
    t, c = _CRC_TABLE, 0xFFFFFFFF
    for byte in data:
        c = t[(c ^ byte) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data: bytes) -> int:
    """how the bundle stores it (tensorflow/core/lib/hash/crc32c.h Mask): rotate right by 15, add a constant"""
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


def read_tensors(prefix: str, names=None, verify_below: int = 1 << 16) -> Dict[str, np.ndarray]:
    """tensors of `<prefix>.index` + `<prefix>.data-*`; checksums are verified for tensors up to `verify_below` bytes (all of them with verify_below=None:
    a pure-Python CRC, seconds per megabyte)"""
    idx = read_index(prefix + ".index")
    shards = idx[""]["num_shards"]
    files = {}
    out = {}
    for name, e in idx.items():
        if name == "" or (names is not None and name not in names):
            continue
        if e["dtype"] not in DTYPES:
            raise ValueError("%s: unsupported dtype %d" % (name, e["dtype"]))
        if e["shard"] not in files:
            p = "%s.data-%05d-of-%05d" % (prefix, e["shard"], shards)
            if not os.path.exists(p):
                raise FileNotFoundError("%s (the tensor data of the checkpoint; the reference repository ships the .index files only)" % p)
            files[e["shard"]] = open(p, "rb")
        f = files[e["shard"]]
        f.seek(e["offset"]); raw = f.read(e["size"])
        dt = np.dtype(DTYPES[e["dtype"]])
        if len(raw) != e["size"] or e["size"] != int(np.prod(e["shape"], dtype=np.int64)) * dt.itemsize:
            raise ValueError("%s: %d bytes for shape %s" % (name, len(raw), e["shape"]))
        if e["crc32c"] is not None and (verify_below is None or e["size"] <= verify_below) and masked_crc32c(raw) != e["crc32c"]:
            raise ValueError("%s: checksum mismatch" % name)
        out[name] = np.frombuffer(raw, dtype=dt.newbyteorder("<")).reshape(e["shape"]).astype(dt)
    for f in files.values():
        f.close()
    return out


def actor_weights(prefix: str, scope: str = "agent", state_dim: Optional[int] = None, verify_below: int = 1 << 16) -> dict:
    """The actor of a reference checkpoint as the weights dict of `deepmimic_amd.policy.Policy`: w1 b1 w2 b2 w3 b3 logstd s_mean s_std a_mean a_std (+ g_mean,
    g_std when the agent has a goal: the first layer then takes [state, goal], learning/pg_agent.py:141-146).  `state_dim`: checked against s_norm when given."""
    idx = read_index(prefix + ".index")
    a = scope + "/main/actor/"
    gated = sorted(n for n in idx if n.startswith(a + "gate"))
    if gated:
        raise NotImplementedError("%s holds a gated actor (learning/nets/fc_2layers_gated_1024units.py: %s, ...); the device actor is the plain fc_2layers_1024units net" % (prefix, gated[0]))
    need = [a + "0/dense/kernel", a + "0/dense/bias", a + "1/dense/kernel", a + "1/dense/bias", a + "dist_gauss_diag/mean/kernel", a + "dist_gauss_diag/mean/bias",
            a + "dist_gauss_diag/logstd/bias"]
    missing = [n for n in need if n not in idx]
    if missing:
        raise ValueError("%s: no %s (not a pg / ppo agent checkpoint of scope %r)" % (prefix, missing[0], scope))
    norms = [scope + "/resource/%s/%s" % (g, k) for g in ("s_norm", "g_norm", "a_norm") for k in ("mean", "std")]
    t = read_tensors(prefix, names=set(need + [n for n in norms if n in idx]), verify_below=verify_below)
    w = dict(w1=t[need[0]], b1=t[need[1]], w2=t[need[2]], b2=t[need[3]], w3=t[need[4]], b3=t[need[5]], logstd=t[need[6]])
    for g, key in (("s_norm", "s"), ("g_norm", "g"), ("a_norm", "a")):
        for k in ("mean", "std"):
            n = scope + "/resource/%s/%s" % (g, k)
            if n in t and t[n].size:
                w["%s_%s" % (key, k)] = t[n].reshape(-1)
    S = w["s_mean"].size if "s_mean" in w else w["w1"].shape[0]
    G = w["g_mean"].size if "g_mean" in w else 0
    if state_dim is not None and S != state_dim:
        raise ValueError("%s was trained on %d state features, the scene records %d" % (prefix, S, state_dim))
    if w["w1"].shape[0] != S + G or w["w2"].shape[0] != w["w1"].shape[1] or w["w3"].shape[0] != w["w2"].shape[1] or w["logstd"].size != w["w3"].shape[1]:
        raise ValueError("%s: layer shapes %s %s %s do not chain from %d + %d inputs" % (prefix, w["w1"].shape, w["w2"].shape, w["w3"].shape, S, G))
    return w
