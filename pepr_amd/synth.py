"""Seeded synthetic WAG+Gamma alignments (SURVEY.md section 8d "Synthetic inputs").

Generator: random-join topology, branch lengths 0.01 + Exp(mean 0.1), sites i.i.d. from WAG
(full-precision pi) with per-site rate drawn from a 16-bin discrete Gamma(alpha) (mean 1).
Used by tests and bench.py only; it is host-side numpy and not part of the scoring path.
"""
import json
import os

import numpy as np

AA = "ARNDCQEGHILKMFPSTWYV"
_GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "wag_constants.json")


def wag_constants():
    with open(_GOLD) as f:
        d = json.load(f)
    S = np.zeros((20, 20))
    k = 0
    for i in range(1, 20):
        for j in range(i):
            S[i, j] = S[j, i] = d["S_lower"][k]
            k += 1
    return S, np.array(d["pi_full"]), np.array(d["pi_raxml_3dp"])


def wag_q(pi):
    S, _, _ = wag_constants()
    pi = pi / pi.sum()
    Q = S * pi[None, :]
    np.fill_diagonal(Q, 0.0)
    np.fill_diagonal(Q, -Q.sum(1))
    Q /= -(pi * np.diag(Q)).sum()
    return Q


def _eig(pi):
    Q = wag_q(pi)
    sp = np.sqrt(pi / pi.sum())
    B = sp[:, None] * Q / sp[None, :]
    B = 0.5 * (B + B.T)
    lam, V = np.linalg.eigh(B)
    return lam, V / sp[:, None], V.T * sp[None, :]


def gamma_mean_rates(alpha, K):
    from scipy.special import gammainc, gammaincinv
    cuts = gammaincinv(alpha, np.arange(1, K) / K)
    cdf = np.concatenate([[0.0], gammainc(alpha + 1, cuts), [1.0]])
    return np.diff(cdf) * K


def random_tree(ntax, rng, names=None):
    """Returns (newick, children dict) of a random-join rooted binary tree."""
    names = names or ["t%d" % i for i in range(ntax)]
    nodes = [(n, None) for n in names]
    live = list(range(ntax))
    kids = {}
    blen = {}
    nid = ntax
    while len(live) > 1:
        i, j = rng.choice(len(live), 2, replace=False)
        a, b = live[i], live[j]
        for x in (a, b):
            blen[x] = 0.01 + rng.exponential(0.1)
        kids[nid] = (a, b)
        live = [x for k, x in enumerate(live) if k not in (i, j)] + [nid]
        nid += 1
    root = live[0]

    def nw(v):
        if v < ntax:
            return names[v]
        a, b = kids[v]
        return "(%s:%.6f,%s:%.6f)" % (nw(a), blen[a], nw(b), blen[b])
    import sys
    sys.setrecursionlimit(max(10000, 4 * ntax))
    return nw(root) + ";", kids, blen, root


def simulate_alignment(ntax, nsites, seed, alpha=0.8, missing_frac=0.0, names=None):
    """Returns (names, rows, newick_true).  rows are python str of length nsites."""
    rng = np.random.default_rng(seed)
    names = names or ["t%d" % i for i in range(ntax)]
    newick, kids, blen, root = random_tree(ntax, rng, names)
    _, pi_full, _ = wag_constants()
    pi = pi_full / pi_full.sum()
    lam, U, Uinv = _eig(pi)
    ncat = 16
    rates = gamma_mean_rates(alpha, ncat)
    cat = rng.integers(0, ncat, nsites)
    order = np.argsort(cat, kind="stable")
    bounds = np.searchsorted(cat[order], np.arange(ncat + 1))
    states = {root: rng.choice(20, size=nsites, p=pi)}
    stack = [root]
    while stack:
        v = stack.pop()
        if v < ntax:
            continue
        for c in kids[v]:
            out = np.empty(nsites, dtype=np.int64)
            u = rng.random(nsites)
            for k in range(ncat):
                idx = order[bounds[k]:bounds[k + 1]]
                if idx.size == 0:
                    continue
                P = (U * np.exp(lam * rates[k] * blen[c])[None, :]) @ Uinv
                P = np.clip(P, 0, None)
                cum = np.cumsum(P, axis=1)
                cum /= cum[:, -1:]
                out[idx] = (u[idx, None] > cum[states[v][idx]]).sum(1)
            states[c] = np.minimum(out, 19)
            stack.append(c)
    aa = np.frombuffer(AA.encode(), dtype=np.uint8)
    rows = []
    for i in range(ntax):
        r = aa[states[i]].copy()
        if missing_frac > 0:
            # block-wise '?' (absent genes in a concatenation) plus scattered '-' and 'X'
            nblk = max(1, nsites // 50)
            for b in range(nblk):
                if rng.random() < missing_frac:
                    r[b * 50:(b + 1) * 50] = ord("?")
            m = rng.random(nsites) < missing_frac * 0.1
            r[m] = ord("-")
            m = rng.random(nsites) < missing_frac * 0.02
            r[m] = ord("X")
        rows.append(r.tobytes().decode())
    return names, rows, newick


def simulate_genes(ngenes, ntax, nsites, seed0=1, alpha=0.8):
    """C3/C4/C5-style gene sets: gene g uses seed seed0+g (seeds 1..G)."""
    return [simulate_alignment(ntax, nsites, seed0 + g, alpha) for g in range(ngenes)]
