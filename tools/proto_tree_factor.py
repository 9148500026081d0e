"""Design study (numpy, CPU) for the next kernel step: a LEVEL-SCHEDULED branch-induced-sparse factorisation H = L^T L (Featherstone, "Efficient
factorization of the joint-space inertia matrix for branched kinematic trees", IJRR 2005) of the characters' mass matrices, in the lane layout
the step kernel could use ("next levers"):

  * lane k = dof k; lane k keeps row k of L in DEPTH-INDEXED slots: slot d = L[k, anc_d(k)], d = 0 .. depth(k) (the ancestors of a dof have
    distinct depths, so register indices are static for ANY topology);
  * the pivots of one tree depth are eliminated together, deepest level first: each pivot lane scales its own row (lane-local), publishes it,
    and every ancestor lane i of pivot k does slot[d] -= Lk[depth(i)] * Lk[d], d <= depth(i) -- slot d of lane i and slot d of lane k name the
    SAME ancestor; a lane with several pivots below it at this level (only the trunk) takes them one after the other;
  * triangular solves run over the same levels.

It checks the algebra against dense numpy (L^T L = H, both solves, Y = L^-T J^T and A = Y^T Y) on random SPD matrices with the tree's sparsity,
and prints what the kernel design needs: levels (= dependent steps) against columns of the dense code, nonzeros, multiply-adds, the longest
per-lane chain of a level, slots per lane.  usage: python tools/proto_tree_factor.py [asset ...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepmimic_amd import model  # noqa: E402


def dof_tree(t):
    """parent dof of every dof (-1 for the first root dof): the dofs of a joint form a chain, a joint's first dof hangs off its parent joint's last"""
    jm = np.asarray(t.joint_mat)
    types = jm[:, model.JD_TYPE].astype(int); parents = jm[:, model.JD_PARENT].astype(int)
    nd = []
    for j, ty in enumerate(types):
        if parents[j] < 0:
            nd.append(6)
        else:
            nd.append({model.JT_SPHERICAL: 3, model.JT_REVOLUTE: 1}.get(ty, 0))
    first = np.cumsum([0] + nd[:-1])
    lam = []
    for j in range(len(types)):
        # last dof of the nearest ancestor joint that has dofs
        a = parents[j]
        while a >= 0 and nd[a] == 0:
            a = parents[a]
        up = -1 if a < 0 else first[a] + nd[a] - 1
        for k in range(nd[j]):
            lam.append(up if k == 0 else first[j] + k - 1)
    return np.array(lam, dtype=int)


def depths(lam):
    d = np.zeros(len(lam), dtype=int)
    for k in range(len(lam)):
        d[k] = 0 if lam[k] < 0 else d[lam[k]] + 1
    return d


def ancestors(lam, k):
    out = []
    while k >= 0:
        out.append(k); k = lam[k]
    return out[::-1]                                    # root .. k


def random_tree_spd(lam, rng):
    """SPD with the mass matrix's sparsity: H[i, j] != 0 only when one of i, j is an ancestor of the other"""
    n = len(lam)
    H = np.zeros((n, n))
    for k in range(n):
        path = ancestors(lam, k)
        for _ in range(2):
            v = np.zeros(n); v[path] = rng.normal(size=len(path))
            H += np.outer(v, v)
    return H + 0.5 * np.eye(n)


def level_factor(H, lam):
    """the level-scheduled LTL in the depth-indexed lane layout; returns (slots [n x (maxdepth+1)], counters)"""
    n = len(lam); dep = depths(lam); D = dep.max() + 1
    anc = [ancestors(lam, k) for k in range(n)]
    slot = np.zeros((n, D))
    for k in range(n):
        for d, a in enumerate(anc[k]):
            slot[k, d] = H[k, a]
    fma = 0; chain = []; publishes = 0
    for lev in range(D - 1, -1, -1):
        piv = [k for k in range(n) if dep[k] == lev]
        for k in piv:                                   # lane-local: sqrt of the diagonal, scale the row
            slot[k, lev] = np.sqrt(slot[k, lev]); slot[k, :lev] /= slot[k, lev]
        publishes += 1                                  # one LDS publish + read round trip for the whole level
        per_lane = np.zeros(n, dtype=int)
        for k in piv:
            for i in anc[k][:-1]:                       # every proper ancestor lane i of the pivot
                di = dep[i]
                slot[i, :di + 1] -= slot[k, di] * slot[k, :di + 1]
                fma += di + 1; per_lane[i] += di + 1
        chain.append(int(per_lane.max()))
    return slot, dict(levels=int(D), fma=int(fma), longest_lane_chain_per_level=chain, round_trips=publishes)


def dense_L(slot, lam):
    n = len(lam); L = np.zeros((n, n))
    for k in range(n):
        for d, a in enumerate(ancestors(lam, k)):
            L[k, a] = slot[k, d]
    return L


def level_solves(slot, lam, b):
    """x = H^-1 b over the levels: L^T z = b is solved leaves-first (a dof's z needs its DESCENDANTS' contributions: each finished lane pushes
    z_k L[k, a] to its ancestors -- the same publish / ancestor-update pattern as the factorisation), then L x = z root-first (lane k needs its
    ANCESTORS' x: a gather over its own slots)"""
    n = len(lam); dep = depths(lam); D = dep.max() + 1
    anc = [ancestors(lam, k) for k in range(n)]
    acc = b.astype(float).copy(); z = np.zeros(n)
    for lev in range(D - 1, -1, -1):
        for k in [k for k in range(n) if dep[k] == lev]:
            z[k] = acc[k] / slot[k, lev]
            for d, a in enumerate(anc[k][:-1]):
                acc[a] -= slot[k, d] * z[k]
    x = np.zeros(n)
    for lev in range(D):
        for k in [k for k in range(n) if dep[k] == lev]:
            s = z[k]
            for d, a in enumerate(anc[k][:-1]):
                s -= slot[k, d] * x[a]
            x[k] = s / slot[k, lev]
    return x


def study(name, rng):
    t = model.load_asset(name)
    lam = dof_tree(t); n = len(lam); dep = depths(lam)
    H = random_tree_spd(lam, rng)
    slot, c = level_factor(H, lam)
    L = dense_L(slot, lam)
    err_f = np.abs(L.T @ L - H).max() / np.abs(H).max()
    b = rng.normal(size=n)
    err_s = np.abs(level_solves(slot, lam, b) - np.linalg.solve(H, b)).max()
    J = rng.normal(size=(12, n))
    Y = np.linalg.solve(L.T, J.T)                       # Y = L^-T J^T ; A = J H^-1 J^T = Y^T Y
    err_a = np.abs(Y.T @ Y - J @ np.linalg.solve(H, J.T)).max()
    nnz = int(sum(dep + 1))
    dense_fma = sum((n - k - 1) * (n - k) // 2 for k in range(n))          # rank-1 updates of a dense right-looking factorisation (lower triangle)
    per_level = np.bincount(dep)
    return dict(asset=name, dofs=n, nnz_L=nnz, dense_entries=n * (n + 1) // 2, elimination_levels=c["levels"], dense_column_steps=n,
                multiply_adds_sparse=c["fma"], multiply_adds_dense=int(dense_fma), pivots_per_level=per_level.tolist(),
                longest_lane_chain_per_level_fma=c["longest_lane_chain_per_level"], slots_per_lane=int(dep.max() + 1),
                lds_round_trips_factor=c["round_trips"], rel_err_factor=float(err_f), err_solve=float(err_s), err_A=float(err_a))


def main():
    names = sys.argv[1:] or ["humanoid3d_walk", "dog3d_pace"]
    rng = np.random.default_rng(0)
    out = [study(nm, rng) for nm in names]
    for o in out:
        assert o["rel_err_factor"] < 1e-12 and o["err_solve"] < 1e-9 and o["err_A"] < 1e-9, o
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
