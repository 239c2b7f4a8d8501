"""CPU check (no GPU) of the algebra behind the default three-pass form of the two-loop recursion
(stochqn_amd/csrc/kernels.hip "three-pass form"): with the products s_a'y_b of every pair a
older than b cached, the recursion of reference src/stochqn.c:663-708 is

    pass 1   b_i = s_i'g
    coef a   alpha_i = rho_i (b_i - sum_{j>i} alpha_j s_i'y_j)                 (newest pair first)
    pass 2   q0 = g - sum_j alpha_j y_j ;  r0 = gamma q0 | h0 q0 | H0 .* q0 ;  v_i = y_i'r0
    coef b   beta_i = rho_i (v_i + sum_{j<i} c_j s_j'y_i) ;  c_i = alpha_i - beta_i   (oldest pair first)
    pass 3   r = r0 + sum_j c_j s_j

-- S is read twice, Y once, and ONLY the upper triangle (older, newer) of S'Y is ever used.  The numpy model
below does exactly that (it never touches an entry s_a'y_b with a newer than b) and must agree with the
oracle's sequential recursion for every ring geometry.  The kernels themselves are held to the oracle by the
-m gpu tests; this file pins the mathematics where no GPU is present."""
import numpy as np
import pytest

from oracle import oracle


def three_pass(g, S, Y, m, used, st, h0=0.0, H0=None):
    rows = [(st + i) % m for i in range(used)]                 # logical order: oldest .. newest
    k = used
    SY = np.full((k, k), np.nan)                                # NaN where the form must never look
    for a in range(k):
        for b in range(a, k):
            SY[a, b] = S[rows[a]] @ Y[rows[b]]
    rho = 1.0 / np.diag(SY)
    b1 = np.array([S[r] @ g for r in rows])                     # pass 1
    alpha = np.zeros(k)
    for i in range(k - 1, -1, -1):                              # coef a
        alpha[i] = rho[i] * (b1[i] - sum(alpha[j] * SY[i, j] for j in range(i + 1, k)))
    q0 = g.copy()                                               # pass 2
    for j in range(k - 1, -1, -1):
        q0 -= alpha[j] * Y[rows[j]]
    if H0 is not None:
        r0 = q0 * H0
    elif h0 > 0:
        r0 = h0 * q0
    else:
        nw = rows[-1]
        r0 = (SY[k - 1, k - 1] / (Y[nw] @ Y[nw])) * q0
    v = np.array([Y[r] @ r0 for r in rows])
    c = np.zeros(k)
    for i in range(k):                                          # coef b
        beta = rho[i] * (v[i] + sum(c[j] * SY[j, i] for j in range(i)))
        c[i] = alpha[i] - beta
    r = r0.copy()                                               # pass 3
    for j in range(k):
        r += c[j] * S[rows[j]]
    assert np.all(np.isfinite(r))                               # a NaN here = the model read a forbidden entry
    return r, rho, alpha


@pytest.mark.parametrize("mode", ["gamma", "h0", "H0"])
@pytest.mark.parametrize("m,used,st", [(1, 1, 0), (5, 5, 3), (5, 2, 0), (5, 3, 4), (20, 20, 7), (20, 1, 19), (24, 24, 23)])
@pytest.mark.parametrize("n", [1, 7, 1000])
def test_three_pass_algebra_equals_the_sequential_recursion(n, m, used, st, mode):
    rng = np.random.default_rng(n * 100 + m * 7 + used + len(mode))
    d = 0.5 + rng.random(n)
    S = 1e-3 * (rng.random((m, n)) - 0.5)
    Y = S * d + 1e-5 * (rng.random((m, n)) - 0.5)               # not exactly D S: S'Y is not symmetric
    g = rng.random(n) - 0.5
    H0 = 0.5 + rng.random(n) if mode == "H0" else None
    h0 = 0.37 if mode == "h0" else 0.0
    want = g.copy()
    rho_w, alpha_w = oracle.two_loop(want, H0, h0, Y.reshape(-1).copy(), S.reshape(-1).copy(), m, used, st)
    got, rho, alpha = three_pass(g, S, Y, m, used, st, h0, H0)
    assert np.linalg.norm(got - want) <= 1e-12 * np.linalg.norm(want)
    assert np.allclose(rho, rho_w[:used], rtol=1e-12, atol=0)
    assert np.allclose(alpha, alpha_w[:used], rtol=1e-9, atol=1e-14 * np.abs(alpha_w[:used]).max())


def test_the_cached_block_survives_a_ring_wrap():
    """Entries are computed when their NEWER partner enters the ring (one column per new pair).  After the ring wraps,
    the overwritten row's old entries as the older partner are garbage -- and are never read: replay 60 insertions
    into a 5-pair ring, maintaining the cache exactly as the library does, and check every step against the oracle."""
    rng = np.random.default_rng(3)
    n, m = 300, 5
    d = 0.5 + rng.random(n)
    S, Y = np.zeros((m, n)), np.zeros((m, n))
    cache = np.full((m, m), np.nan)                             # cache[a, b] = s_a'y_b, physical rows
    used, st_ix = 0, 0
    for t in range(60):
        s = 1e-3 * (rng.random(n) - 0.5)
        S[st_ix], Y[st_ix] = s, s * d + 1e-5 * (rng.random(n) - 0.5)
        cache[st_ix, :] = np.nan                                # everything involving the old occupant is gone ...
        cache[:, st_ix] = np.nan
        new = st_ix
        st_ix = (st_ix + 1) % m
        used = min(used + 1, m)
        st = 0 if st_ix == used and used < m else (st_ix if used == m else 0)
        rows = [(st + i) % m for i in range(used)]
        assert rows[-1] == new
        for a in rows:                                          # ... and ONE column is computed: s_a'y_new for every pair in use
            cache[a, new] = S[a] @ Y[new]
        g = rng.random(n) - 0.5
        want = g.copy()
        oracle.two_loop(want, None, 0.0, Y.reshape(-1).copy(), S.reshape(-1).copy(), m, used, st)
        k = used
        rho = np.array([1.0 / cache[r, r] for r in rows])
        b1 = np.array([S[r] @ g for r in rows])
        alpha = np.zeros(k)
        for i in range(k - 1, -1, -1):
            alpha[i] = rho[i] * (b1[i] - sum(alpha[j] * cache[rows[i], rows[j]] for j in range(i + 1, k)))
        q0 = g - sum(alpha[j] * Y[rows[j]] for j in range(k))
        r0 = (cache[new, new] / (Y[new] @ Y[new])) * q0
        v = np.array([Y[r] @ r0 for r in rows])
        c = np.zeros(k)
        for i in range(k):
            c[i] = alpha[i] - rho[i] * (v[i] + sum(c[j] * cache[rows[j], rows[i]] for j in range(i)))
        got = r0 + sum(c[j] * S[rows[j]] for j in range(k))
        assert np.all(np.isfinite(got)), t                      # a NaN = an entry that was never (re)computed was read
        assert np.linalg.norm(got - want) <= 1e-11 * np.linalg.norm(want), t
