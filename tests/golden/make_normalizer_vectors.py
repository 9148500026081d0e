#!/usr/bin/env python
"""Golden vectors for the device normalizer (deepmimic_amd/csrc/dm_norm.h) from the REFERENCE's own learning/normalizer.py, imported from
/root/reference in the build container (mpi4py is absent: util.mpi_util is stubbed with a settable worker count -- reduce_sum over W workers
that recorded the same kind of data is a sum of W pending blocks, which the script forms explicitly).
Writes tests/golden/normalizer_vectors.npz.  Usage: python tests/golden/make_normalizer_vectors.py [/root/reference]"""
import os
import sys
import types

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
# stubs for the two modules normalizer.py imports besides numpy
util = types.ModuleType("util"); mpi = types.ModuleType("util.mpi_util"); logger = types.ModuleType("util.logger")
mpi.EXTRA = None                                    # what the OTHER workers contribute to the next reduce_sum calls (count, sum, sum_sq)


def reduce_sum(x):
    if mpi.EXTRA:
        x = x + mpi.EXTRA.pop(0)
    return x


mpi.reduce_sum = reduce_sum; mpi.is_root_proc = lambda: True; mpi.bcast = lambda x: None


class Logger:
    @staticmethod
    def print(*a):
        return " ".join(str(x) for x in a)


logger.Logger = Logger; util.mpi_util = mpi; util.logger = logger
sys.modules.update({"util": util, "util.mpi_util": mpi, "util.logger": logger})
sys.path.insert(0, os.path.join(REF, "learning"))
import normalizer as refnorm                         # noqa: E402  (the reference's file, unmodified)

rng = np.random.default_rng(20240917)
out = {}
cases = {
    # name: (group ids or None, eps, clip)
    "single": (None, 0.02, np.inf),
    "groups": (np.array([0, 0, 1, 1, 1, 2, 2, -1, -1, 0, 3, 3, 3, 0, -1, 5], np.int32), 0.02, 5.0),
    "wide": (np.concatenate([np.zeros(1, np.int32), np.repeat(np.arange(1, 76, dtype=np.int32), 3), -np.ones(1, np.int32)]), 0.05, 10.0),   # 227 columns like the humanoid's state
}
for name, (gids, eps, clip) in cases.items():
    size = 7 if gids is None else gids.size
    nrm = refnorm.Normalizer(size, groups_ids=gids, eps=eps, clip=clip)
    mean0 = rng.normal(size=size); std0 = 0.5 + rng.random(size)
    nrm.set_mean_std(mean0.copy(), std0.copy())
    out[name + "/gids"] = -2 * np.ones(1, np.int32) if gids is None else gids
    out[name + "/eps_clip"] = np.array([eps, clip]); out[name + "/mean0"] = mean0; out[name + "/std0"] = std0
    rounds = []
    for it in range(4):
        batches = []
        for b in range(1 + it % 3):
            n = int(rng.integers(1, 70))
            x = (rng.normal(size=(n, size)) * (0.1 + 3 * rng.random(size)) + rng.normal(size=size)).astype(np.float32)
            if it == 2:
                x[:, : size // 2] = x[0, : size // 2]                # constant columns: std falls to eps
            batches.append(x)
            nrm.record(x.astype(np.float64))
        other = None
        if it == 3:                                                   # a second worker's pending block joins the reduce_sum
            y = (rng.normal(size=(33, size)) * 2).astype(np.float32)
            other = y
            y64 = y.astype(np.float64)
            mpi.EXTRA = [y64.shape[0], y64.sum(0), np.square(y64).sum(0)]
        nrm.update()
        for k, x in enumerate(batches):
            out["%s/it%d/x%d" % (name, it, k)] = x
        if other is not None:
            out["%s/it%d/other" % (name, it)] = other
        out["%s/it%d/nbatches" % (name, it)] = np.array([len(batches)])
        out["%s/it%d/count" % (name, it)] = np.array([nrm.count], np.int64)
        out["%s/it%d/mean" % (name, it)] = nrm.mean.copy(); out["%s/it%d/std" % (name, it)] = nrm.std.copy(); out["%s/it%d/mean_sq" % (name, it)] = nrm.mean_sq.copy()
        q = (rng.normal(size=(5, size)) * 4).astype(np.float32)
        out["%s/it%d/q" % (name, it)] = q; out["%s/it%d/q_norm" % (name, it)] = nrm.normalize(q.astype(np.float64))
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "normalizer_vectors.npz"), **out)
print("wrote", len(out), "arrays")
