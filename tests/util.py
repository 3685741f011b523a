"""Test helpers: an independent numpy pruning implementation (different language, different
underflow scheme: per-node max normalisation in log space) used to cross-check the C oracle."""
import re

import numpy as np

from pepr_amd import synth

AA = synth.AA


def parse_newick(nw):
    """Minimal parser -> nested (children, name, length)."""
    s = nw.strip().rstrip(";")
    pos = 0

    def node():
        nonlocal pos
        kids = []
        if s[pos] == "(":
            pos += 1
            while True:
                kids.append(node())
                if s[pos] == ",":
                    pos += 1
                    continue
                if s[pos] == ")":
                    pos += 1
                    break
        m = re.match(r"[^,():;]*", s[pos:])
        name = m.group(0)
        pos += len(name)
        length = 0.0
        if pos < len(s) and s[pos] == ":":
            m = re.match(r":([-+0-9.eE]+)", s[pos:])
            length = float(m.group(1))
            pos += len(m.group(0))
        return (kids, name, length)
    return node()


def numpy_lnl(names, rows, newick, alpha, pi_mode="raxml", ncat=4):
    """Per-site lnL by pruning in numpy with log-space normalisation. Returns (total, per-site)."""
    S, pi_full, pi_3 = synth.wag_constants()
    pi = np.asarray(pi_mode, float) if not isinstance(pi_mode, str) else (pi_3 if pi_mode == "raxml" else pi_full)    # or 20 explicit frequencies
    pi = pi / pi.sum()
    lam, U, Uinv = synth._eig(pi)
    rates = synth.gamma_mean_rates(alpha, ncat) if ncat > 1 else np.ones(1)
    idx = {n: i for i, n in enumerate(names)}
    L = len(rows[0])
    code = {c: i for i, c in enumerate(AA)}

    def tipvec(row):
        v = np.zeros((L, 20))
        for s, ch in enumerate(row.upper()):
            if ch in code:
                v[s, code[ch]] = 1
            elif ch == "B":
                v[s, [2, 3]] = 1
            elif ch == "Z":
                v[s, [5, 6]] = 1
            else:
                v[s, :] = 1
        return v

    def P(t):
        return np.stack([(U * np.exp(lam * r * t)[None, :]) @ Uinv for r in rates])   # [K,20,20]

    def rec(nd):
        kids, name, _ = nd
        if not kids:
            v = tipvec(rows[idx[name]])
            return np.repeat(v[None], len(rates), 0), np.zeros(L)     # [K,L,20], logscale[L]
        out = None
        ls = np.zeros(L)
        for k in kids:
            cv, cl = rec(k)
            x = np.einsum("kij,klj->kli", P(max(k[2], 0.0)), cv)
            out = x if out is None else out * x
            ls = ls + cl
        mx = out.max(axis=(0, 2))
        out = out / mx[None, :, None]
        return out, ls + np.log(mx)

    tree = parse_newick(newick)
    cv, ls = rec(tree)
    site = np.log((cv * pi[None, None, :]).sum(2).mean(0)) + ls
    return site.sum(), site


def splits(newick):
    """dict {frozenset(smaller-side taxa) -> branch length} over the internal edges of a tree."""
    tree = parse_newick(newick)
    allt = []

    def leaves(nd):
        if not nd[0]:
            return [nd[1]]
        return sum((leaves(k) for k in nd[0]), [])
    allt = frozenset(leaves(tree))
    out = {}

    def rec(nd, top):
        if not nd[0]:
            return frozenset([nd[1]])
        s = frozenset().union(*[rec(k, False) for k in nd[0]])
        if not top and 1 < len(s) < len(allt) - 1:
            key = s if (len(s) * 2 < len(allt) or (len(s) * 2 == len(allt) and min(allt) in s)) else allt - s
            out[key] = out.get(key, 0.0) + nd[2]
        return s
    rec(tree, True)
    return out


def rf_collapsed(nw_a, nw_b, min_len=1e-5):
    """Robinson-Foulds distance that ignores internal branches of (near) zero length: a split only
    counts as a difference if it is supported by a branch longer than min_len in its own tree."""
    a, b = splits(nw_a), splits(nw_b)
    d = sum(1 for s, l in a.items() if l > min_len and s not in b) + sum(1 for s, l in b.items() if l > min_len and s not in a)
    return d


def fitch_length(names, rows, newick):
    """Independent Fitch (1971) length in numpy over raw columns (no pattern compression):
    gap/?/X = any state, B = N|D, Z = Q|E.  The unrooted tree is rooted on its first child."""
    idx = {n: i for i, n in enumerate(names)}
    arr = np.frombuffer("".join(rows).upper().encode(), dtype=np.uint8).reshape(len(rows), -1)
    masks = np.full(256, (1 << 20) - 1, dtype=np.int64)
    for i, ch in enumerate(AA):
        masks[ord(ch)] = 1 << i
    masks[ord("B")] = (1 << AA.index("N")) | (1 << AA.index("D"))
    masks[ord("Z")] = (1 << AA.index("Q")) | (1 << AA.index("E"))
    total = 0

    def down(nd):
        nonlocal total
        kids, name, _ = nd
        if not kids:
            return masks[arr[idx[name]]]
        sets = [down(k) for k in kids]
        # resolve a multifurcation as a caterpillar rooted at the LAST child: exact for 3 children
        # (an unrooted trifurcation), which is all the callers pass
        cur = sets[0]
        for s in sets[1:]:
            inter = cur & s
            empty = inter == 0
            total += int(empty.sum())
            cur = np.where(empty, cur | s, inter)
        return cur
    down(parse_newick(newick))
    return total


def prune_newick(nw, keep):
    """The tree induced on the leaf set `keep` (degree-2 nodes suppressed, their branch lengths added), as an unrooted
    Newick with a trifurcation at the top: what a gene that lacks some taxa can at best recover of the generating tree."""
    keep = set(keep)

    def rec(nd):
        kids, name, length = nd
        if not kids:
            return (([], name, length) if name in keep else None)
        sub = [x for x in (rec(k) for k in kids) if x is not None]
        if not sub:
            return None
        if len(sub) == 1:
            return (sub[0][0], sub[0][1], sub[0][2] + length)
        return (sub, "", length)

    def fmt(nd):
        kids, name, length = nd
        return ("(" + ",".join(fmt(k) for k in kids) + ")" if kids else name) + ":%.8f" % length

    t = rec(parse_newick(nw))
    kids = list(t[0])
    while len(kids) == 2:                    # rooted binary top -> unrooted trifurcation
        a, b = kids
        if a[0]:
            kids = list(a[0]) + [(b[0], b[1], b[2] + a[2])]
        elif b[0]:
            kids = [(a[0], a[1], a[2] + b[2])] + list(b[0])
        else:
            break
    return "(" + ",".join(fmt(k) for k in kids) + ");"
