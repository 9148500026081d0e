#!/usr/bin/env python3
"""Serialise the `dm_scene_tables` of a scene (include/dm_hip.h) into one binary blob that a native caller can mmap / read and hand to
dm_create without any Python in the process (tests/native/smoke.c is that caller; the reference's native entry point is DeepMimicCore/Main.cpp:
38-75 -- it builds its scene from an arg file through cArgParser + the JSON loaders, which SURVEY 2 row 13 leaves on the host side; the flat
tables are what crosses the C-ABI, so a native host either parses the files itself or ships this blob).

    python tools/dump_tables.py humanoid3d_walk out.dmtbl          # an asset name of deepmimic_amd/assets, or --args <arg file> [--data-root DIR]

Layout (little endian): char magic[8] "DMTBL\\0\\0\\1" | int32 DM_ABI_VERSION | int32 sizeof(dm_scene_tables) | uint64 n | n x {uint64 offset of a
pointer member inside the struct, uint64 offset of its array inside the blob, uint64 bytes} | the struct (pointer members zero) | the arrays
(16-byte aligned).  The reader patches member = blob base + array offset."""
import argparse
import ctypes as C
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepmimic_amd import core, model  # noqa: E402

MAGIC = b"DMTBL\0\0\1"


def dump(tables, path, test_mode=False):
    st, keep = core.fill_scene_tables(tables, test_mode=test_mode)
    raw = bytearray(bytes(st))
    by_addr = {a.ctypes.data: a for a in keep}
    ptrs = []
    for name, typ in core._SceneTables._fields_:
        if not (isinstance(typ, type) and issubclass(typ, C._Pointer)):
            continue
        off = getattr(core._SceneTables, name).offset
        addr = C.cast(getattr(st, name), C.c_void_p).value
        raw[off:off + 8] = b"\0" * 8
        if addr:
            ptrs.append((off, by_addr[addr]))
    head = 8 + 4 + 4 + 8 + 24 * len(ptrs)
    pos = (head + len(raw) + 15) // 16 * 16
    table, blobs = [], []
    for off, a in ptrs:
        b = a.tobytes()
        table.append((off, pos, len(b))); blobs.append((pos, b))
        pos = (pos + len(b) + 15) // 16 * 16
    out = bytearray(pos)
    out[0:8] = MAGIC
    struct.pack_into("<iiQ", out, 8, core.ABI_VERSION, len(raw), len(ptrs))
    for i, (off, bo, nb) in enumerate(table):
        struct.pack_into("<QQQ", out, 24 + 24 * i, off, bo, nb)
    out[head:head + len(raw)] = raw
    for bo, b in blobs:
        out[bo:bo + len(b)] = b
    with open(path, "wb") as f:
        f.write(out)
    return len(out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("scene", nargs="?", default="humanoid3d_walk")
    ap.add_argument("out")
    ap.add_argument("--args", default=None, help="a reference arg file instead of an asset name")
    ap.add_argument("--data-root", default=".")
    ap.add_argument("--test-mode", action="store_true")
    a = ap.parse_args()
    t = model.load_scene_from_args(["--arg_file", a.args], data_root=a.data_root) if a.args else model.load_asset(a.scene)
    n = dump(t, a.out, a.test_mode)
    print("%s: %d bytes" % (a.out, n))
