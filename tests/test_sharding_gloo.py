"""world_size-2 (gloo, CPU) check of the sharding scheme the multi-GPU path uses: every rank owns
a contiguous slice of all n-vectors and of every pair row; each dot product of the recursion is a
local partial plus one all-reduce(sum) of 1-3 scalars; the guard uses the GLOBAL n and the global
sum of squares.  The per-rank arithmetic here is numpy (a model of the sweeps in kernels.hip, in
the same fused order); the unsharded oracle is the reference result.

This file checks the ALGEBRA on the CPU (no GPU here, and the product has no CPU path).  The library itself
runs sharded in the -m gpu tests: world-size-3 over gloo through stochqn_hip_comm_init_custom
(tests/test_gpu_parity.py::test_bench_starts_its_own_ranks_and_shards_one_problem), P host threads with the
loop-back reducer (::test_sharded_library_equals_unsharded_oracle) and the single-process multi-device mode
(tests/test_gpu_devices.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def allsum(vals):
    t = torch.tensor(vals, dtype=torch.float64)
    dist.all_reduce(t)
    return t.tolist()


def sharded_two_loop(q, S, Y, m, used, st, sy, yy):
    """The sweep chain of kernels.hip on one shard; S, Y are [m][n_local]."""
    row = lambda i: (st + i) % m
    k = used
    alpha = [0.0] * k
    (p,) = allsum([float(S[row(k - 1)] @ q)])                        # first
    for i in range(k - 1, 0, -1):                                    # bwd
        alpha[i] = (1.0 / sy[row(i)]) * p
        q -= alpha[i] * Y[row(i)]
        (p,) = allsum([float(S[row(i - 1)] @ q)])
    alpha[0] = (1.0 / sy[row(0)]) * p                                # mid
    q -= alpha[0] * Y[row(0)]
    q *= sy[row(k - 1)] / yy[row(k - 1)]
    (p,) = allsum([float(Y[row(0)] @ q)])
    for i in range(k - 1):                                           # fwd
        beta = (1.0 / sy[row(i)]) * p
        q += (alpha[i] - beta) * S[row(i)]
        (p,) = allsum([float(Y[row(i + 1)] @ q)])
    beta = (1.0 / sy[row(k - 1)]) * p                                # fwd_last
    q += (alpha[k - 1] - beta) * S[row(k - 1)]
    ss, nonfinite = allsum([float(q @ q), float(np.sum(~np.isfinite(q)))])
    return ss, nonfinite


def worker(rank, world, port, n, m, st, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(123)                                 # same global data on every rank
    d = 0.5 + rng.random(n)
    S = 1e-3 * (rng.random((m, n)) - 0.5)
    Y = S * d
    g = rng.random(n) - 0.5
    x = 1.0 + rng.random(n)
    lo, hi = n * rank // world, n * (rank + 1) // world              # contiguous slice, bench.py's partition
    Sl, Yl, ql, xl = S[:, lo:hi].copy(), Y[:, lo:hi].copy(), g[lo:hi].copy(), x[lo:hi].copy()
    # pair statistics are all-reduced when a pair is accepted (accept_or_reject in machines.cpp)
    sy = [allsum([float(Sl[r] @ Yl[r])])[0] for r in range(m)]
    yy = [allsum([float(Yl[r] @ Yl[r])])[0] for r in range(m)]
    (n_global,) = allsum([float(hi - lo)])                           # comm_attach()
    ss, nonfinite = sharded_two_loop(ql, Sl, Yl, m, m, st, sy, yy)
    bad = nonfinite > 0 or not (np.sqrt(ss) <= 1e3 * n_global)       # identical decision on every rank
    if not bad:
        xl -= 0.1 * ql
    np.save(os.path.join(out_dir, "r%d.npy" % rank), np.concatenate([ql, xl]))
    np.save(os.path.join(out_dir, "meta%d.npy" % rank), np.array([n_global, ss, float(bad)]))
    dist.destroy_process_group()


@pytest.mark.parametrize("n,m,st", [(1001, 5, 3), (4096, 3, 0)])
def test_sharded_recursion_equals_unsharded_oracle(tmp_path, n, m, st):
    from oracle import oracle
    import socket
    world = 2
    with socket.socket() as sock:                      # a port nobody is listening on
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    mp.spawn(worker, args=(world, port, n, m, st, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(123)
    d = 0.5 + rng.random(n)
    S = 1e-3 * (rng.random((m, n)) - 0.5)
    Y = S * d
    g = rng.random(n) - 0.5
    x = 1.0 + rng.random(n)
    want = g.copy()
    oracle.two_loop(want, None, 0.0, Y.reshape(-1).copy(), S.reshape(-1).copy(), m, m, st)
    parts = [np.load(tmp_path / ("r%d.npy" % r)) for r in range(world)]
    q = np.concatenate([p[:len(p) // 2] for p in parts])
    xs = np.concatenate([p[len(p) // 2:] for p in parts])
    metas = [np.load(tmp_path / ("meta%d.npy" % r)) for r in range(world)]
    assert np.array_equal(metas[0], metas[1])                        # bit-identical scalars on all ranks
    assert metas[0][0] == n and metas[0][2] == 0.0
    assert np.linalg.norm(q - want) / np.linalg.norm(want) < 1e-12
    assert np.linalg.norm(xs - (x - 0.1 * want)) / np.linalg.norm(x) < 1e-12


# ------------------------------------------------------------------------------------------------
# the default three-pass form, sharded: THREE reductions per step
#   pass 1  k partial dots s_i'g -- 2k on the step after a pair entered the ring: the probe y_new yields the new
#           column s_i'y_new of the cached block in the same all-reduce
#   pass 2  k partial dots v_i = y_i'r0
#   pass 3  the guard sums (sum r^2, #non-finite)
# The scalar recursions between the passes run on every rank from the all-reduced numbers, so alpha, c and the
# verdict are bit-identical everywhere.  As above the per-rank arithmetic is a numpy model of the kernels.
# ------------------------------------------------------------------------------------------------
def sharded_three_pass_step(g, S, Y, m, used, st, cache, sy, yy, fresh, counts):
    rows = [(st + i) % m for i in range(used)]
    k = used
    loc = [float(S[r] @ g) for r in rows]
    if fresh is not None:                                            # the probe of pass 1
        loc += [float(S[r] @ Y[fresh]) for r in rows]
    tot = allsum(loc)
    counts.append(len(loc))
    b1 = tot[:k]
    if fresh is not None:
        for i, r in enumerate(rows):
            cache[r, fresh] = tot[k + i]
    SY = lambda a, b: sy[rows[a]] if a == b else cache[rows[a], rows[b]]
    alpha = [0.0] * k
    for i in range(k - 1, -1, -1):                                   # coef a, on every rank alike
        alpha[i] = (1.0 / sy[rows[i]]) * (b1[i] - sum(alpha[j] * SY(i, j) for j in range(i + 1, k)))
    q = g
    for j in range(k - 1, -1, -1):                                   # pass 2
        q -= alpha[j] * Y[rows[j]]
    q *= sy[rows[-1]] / yy[rows[-1]]
    v = allsum([float(Y[r] @ q) for r in rows])
    counts.append(k)
    c = [0.0] * k
    for i in range(k):                                               # coef b
        c[i] = alpha[i] - (1.0 / sy[rows[i]]) * (v[i] + sum(c[j] * SY(j, i) for j in range(i)))
    for j in range(k):                                               # pass 3
        q += c[j] * S[rows[j]]
    ss, nonfinite = allsum([float(q @ q), float(np.sum(~np.isfinite(q)))])
    counts.append(2)
    return ss, nonfinite


def worker3(rank, world, port, n, m, steps, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(321)
    d = 0.5 + rng.random(n)
    lo, hi = n * rank // world, n * (rank + 1) // world
    S, Y = np.zeros((m, hi - lo)), np.zeros((m, hi - lo))
    cache = np.full((m, m), np.nan)
    sy, yy = [np.nan] * m, [np.nan] * m
    used, st_ix, fresh = 0, 0, None
    (n_global,) = allsum([float(hi - lo)])
    outs, counts = [], []
    for t in range(steps):
        s = 1e-3 * (rng.random(n) - 0.5)                             # a new pair enters (same global numbers on every rank)
        y = s * d + 1e-5 * (rng.random(n) - 0.5)
        S[st_ix], Y[st_ix] = s[lo:hi], y[lo:hi]
        cache[st_ix, :] = np.nan
        cache[:, st_ix] = np.nan
        sy[st_ix], _, yy[st_ix] = allsum([float(S[st_ix] @ Y[st_ix]), float(S[st_ix] @ S[st_ix]), float(Y[st_ix] @ Y[st_ix])])   # accept_or_reject
        fresh = st_ix
        st_ix = (st_ix + 1) % m
        used = min(used + 1, m)
        st = st_ix if used == m else 0
        for rep in range(2):                                         # two steps per pair: with and without the probe
            g = (rng.random(n) - 0.5)[lo:hi].copy()
            before = len(counts)
            ss, nonfinite = sharded_three_pass_step(g, S, Y, m, used, st, cache, sy, yy, fresh if rep == 0 else None, counts)
            assert len(counts) - before == 3                          # three reductions per step, probe or not
            bad = nonfinite > 0 or not (np.sqrt(ss) <= 1e3 * n_global)
            outs.append(np.concatenate([g, [ss, float(bad)]]))
    np.save(os.path.join(out_dir, "t%d.npy" % rank), np.concatenate(outs))
    np.save(os.path.join(out_dir, "c%d.npy" % rank), np.array(counts))
    dist.destroy_process_group()


@pytest.mark.parametrize("n,m,steps", [(1001, 4, 7), (257, 6, 9)])
def test_sharded_three_pass_form_equals_unsharded_oracle(tmp_path, n, m, steps):
    from oracle import oracle
    import socket
    world = 2
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    mp.spawn(worker3, args=(world, port, n, m, steps, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / ("t%d.npy" % r)) for r in range(world)]
    cnt = [np.load(tmp_path / ("c%d.npy" % r)) for r in range(world)]
    assert np.array_equal(cnt[0], cnt[1]) and len(cnt[0]) == 3 * 2 * steps
    sizes = [n * (r + 1) // world - n * r // world for r in range(world)]
    rng = np.random.default_rng(321)
    d = 0.5 + rng.random(n)
    S, Y = np.zeros((m, n)), np.zeros((m, n))
    used, st_ix = 0, 0
    pos = [0] * world
    for t in range(steps):
        s = 1e-3 * (rng.random(n) - 0.5)
        S[st_ix], Y[st_ix] = s, s * d + 1e-5 * (rng.random(n) - 0.5)
        st_ix = (st_ix + 1) % m
        used = min(used + 1, m)
        st = st_ix if used == m else 0
        for rep in range(2):
            want = rng.random(n) - 0.5
            oracle.two_loop(want, None, 0.0, Y.reshape(-1).copy(), S.reshape(-1).copy(), m, used, st)
            pieces, tails = [], []
            for r in range(world):
                seg = got[r][pos[r]:pos[r] + sizes[r] + 2]
                pos[r] += sizes[r] + 2
                pieces.append(seg[:-2])
                tails.append(seg[-2:])
            assert np.array_equal(tails[0], tails[1])                # ss and the verdict: bit-identical on both ranks
            assert tails[0][1] == 0.0
            q = np.concatenate(pieces)
            assert np.linalg.norm(q - want) <= 1e-11 * np.linalg.norm(want), (t, rep)
            assert abs(tails[0][0] - want @ want) <= 1e-10 * (want @ want)
