"""Thin Python view of the C ABI (include/peprml.h): Context, resident Batch, one-shot calls.

PyTorch is not needed here; it is used only by bench.py / distributed.py for rank plumbing.
"""
import ctypes as C

import numpy as np

from . import _lib

KERNELS = {"pmat": 0, "newview": 1, "evaluate": 2, "sumtable": 3, "newton": 4, "reduce": 5,
           "host_build": 6, "host_wait": 7}
PI_RAXML_3DP, PI_WAG_FULL, PI_EMPIRICAL = 0, 1, 2      # PI_EMPIRICAL = PROTGAMMAWAGF (frequencies counted per gene)


class PmlError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        msg = _lib.load().pml_strerror(code).decode()
        super().__init__("%s (%d)%s" % (msg, code, ": " + detail if detail else ""))


def _aln_struct(names, rows, keep):
    n = len(names)
    if n != len(rows):
        raise ValueError("names/rows length mismatch")
    L = len(rows[0]) if n else 0
    for r in rows:
        if len(r) != L:
            raise ValueError("alignment rows have different lengths")
    na = (C.c_char_p * n)(*[s.encode() if isinstance(s, str) else s for s in names])
    ra = (C.c_char_p * n)(*[s.encode() if isinstance(s, str) else s for s in rows])
    keep.extend([na, ra])
    return _lib.Alignment(n, L, na, ra)


def _model(ncat=4, alpha=1.0, pi_mode=PI_RAXML_3DP):
    return _lib.Model(ncat, alpha, pi_mode)


def _opts(optimize_alpha=True, nni=True, spr_radius=0, epsilon=0.0, seed=0, constraints=None, keep=None):
    """constraints: (names, rows) -- FastTree-style 0/1/- matrix, one row per named taxon."""
    o = _lib.SearchOpts(int(optimize_alpha), int(nni), int(spr_radius), float(epsilon), int(seed), 0, 0, None, None)
    if constraints is not None:
        names, rows = constraints
        n = len(names)
        na = (C.c_char_p * n)(*[s.encode() for s in names])
        ra = (C.c_char_p * n)(*[s.encode() for s in rows])
        if keep is not None:
            keep.extend([na, ra])
        o.nconstraints = len(rows[0]) if n else 0
        o.constraint_ntax = n
        o.constraint_names = na
        o.constraint_rows = ra
        o._keep = (na, ra)
    return o


def constraints_from_tree(newick):
    """The 0/1 matrix FastTreeRunner.getFastTreeConstraintsForTree builds (FastTreeRunner.java:243-273):
    taxa sorted by name, one column per node of the constraint tree (1 = leaf below that node)."""
    import re
    s = newick.strip().rstrip(";")
    pos = 0
    cols = []

    def node():
        nonlocal pos
        leaves = []
        if s[pos] == "(":
            pos += 1
            while True:
                leaves += node()
                if s[pos] == ",":
                    pos += 1
                    continue
                if s[pos] == ")":
                    pos += 1
                    break
            m = re.match(r"[^,():;]*(:[-+0-9.eE]+)?", s[pos:])
            pos += len(m.group(0))
        else:
            m = re.match(r"([^,():;]*)(:[-+0-9.eE]+)?", s[pos:])
            pos += len(m.group(0))
            leaves = [m.group(1)]
        cols.append(set(leaves))
        return leaves
    taxa = sorted(node())
    rows = ["".join("1" if t in c else "0" for c in cols) for t in taxa]
    return taxa, rows


class Context:
    """One engine context = one HIP device + stream (one per rank / per GPU)."""

    def __init__(self, device=0, profile=False, arena_bytes=0):
        self.L = _lib.load()
        self.ptr = C.c_void_p()
        cfg = _lib.Config(device, int(profile), int(arena_bytes))
        rc = self.L.pml_create(C.byref(cfg), C.byref(self.ptr))
        if rc:
            raise PmlError(rc, self.L.pml_last_error(None).decode())

    def _check(self, rc):
        if rc:
            raise PmlError(rc, self.L.pml_last_error(self.ptr).decode())

    def close(self):
        if self.ptr:
            self.L.pml_destroy(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- one-shot, gene-batched ----
    def _oneshot(self, fn, genes, newicks, model, extra):
        keep = []
        n = len(genes)
        alns = (_lib.Alignment * n)(*[_aln_struct(g[0], g[1], keep) for g in genes])
        nw = None
        if newicks is not None:
            nw = (C.c_char_p * n)(*[(s.encode() if s is not None else None) for s in newicks])
        res = (_lib.Result * n)()
        rc = fn(self.ptr, n, alns, nw, C.byref(model), *extra, res)
        out = []
        if rc == 0:
            for r in res:
                d = {"lnl": r.lnl, "alpha": r.alpha, "tree_length": r.tree_length, "npatterns": r.npatterns,
                     "nsites": r.nsites, "newick": C.string_at(r.newick).decode() if r.newick else None}
                if r.site_lnl:
                    d["site_lnl"] = np.ctypeslib.as_array(r.site_lnl, shape=(max(r.nsites, 1),))[:r.nsites].copy()
                out.append(d)
        for r in res:
            self.L.pml_result_free(C.byref(r))
        self._check(rc)
        return out

    def score(self, genes, newicks, alpha=1.0, ncat=4, pi_mode=PI_RAXML_3DP, site_lnl=False):
        """genes: list of (names, rows); newicks: list of str.  Fixed tree + lengths + alpha -> lnL."""
        return self._oneshot(self.L.pml_score_batch, genes, newicks, _model(ncat, alpha, pi_mode), (1 if site_lnl else 0,))

    def optimize(self, genes, newicks, alpha=1.0, ncat=4, pi_mode=PI_RAXML_3DP, optimize_alpha=True, epsilon=1e-4):
        o = _opts(optimize_alpha, False, 0, epsilon)
        return self._oneshot(self.L.pml_optimize_batch, genes, newicks, _model(ncat, alpha, pi_mode), (C.byref(o),))

    def search(self, genes, start_newicks=None, alpha=1.0, ncat=4, pi_mode=PI_RAXML_3DP, optimize_alpha=True,
               nni=True, spr_radius=0, epsilon=1e-3, constraints=None, seed=0):
        """seed != 0: RAxML-style randomised stepwise-addition parsimony start trees instead of NJ."""
        o = _opts(optimize_alpha, nni, spr_radius, epsilon, seed=seed, constraints=constraints)
        return self._oneshot(self.L.pml_search_batch, genes, start_newicks, _model(ncat, alpha, pi_mode), (C.byref(o),))

    def sh_support(self, genes, newicks, alpha=1.0, ncat=4, pi_mode=PI_RAXML_3DP, nboot=1000, seed=314159):
        """FastTree's SH-like local supports for given trees: list of dicts, "newick" carries 0-1 labels (3 decimals)."""
        return self._oneshot(self.L.pml_sh_support_batch, genes, newicks, _model(ncat, alpha, pi_mode), (int(nboot), int(seed)))

    def gamma20(self, genes, newicks, pi_mode=PI_WAG_FULL):
        """FastTree's `-gamma` step on given trees: list of {"lnl" (Gamma20), "alpha", "rescale", "newick" (lengths x rescale)}."""
        keep = []
        n = len(genes)
        alns = (_lib.Alignment * n)(*[_aln_struct(g[0], g[1], keep) for g in genes])
        nw = (C.c_char_p * n)(*[s.encode() for s in newicks])
        res = (_lib.Result * n)()
        rs = (C.c_double * n)()
        m = _model(4, 1.0, pi_mode)
        rc = self.L.pml_gamma20_batch(self.ptr, n, alns, nw, C.byref(m), res, rs)
        out = []
        if rc == 0:
            for r, s_ in zip(res, rs):
                out.append({"lnl": r.lnl, "alpha": r.alpha, "rescale": float(s_), "tree_length": r.tree_length,
                            "npatterns": r.npatterns, "newick": C.string_at(r.newick).decode()})
        for r in res:
            self.L.pml_result_free(C.byref(r))
        self._check(rc)
        return out

    def bootstrap(self, gene, reps=100, seed=1, spr_radius=5, epsilon=1e-3, alpha=1.0, ncat=4, pi_mode=PI_RAXML_3DP):
        """`raxmlHPC -f a -x seed -N reps`: best ML tree with percent supports + the replicate trees."""
        keep = []
        a = _aln_struct(gene[0], gene[1], keep)
        m = _model(ncat, alpha, pi_mode)
        res, rep = _lib.Result(), C.c_void_p()
        rc = self.L.pml_bootstrap(self.ptr, C.byref(a), C.byref(m), reps, seed, spr_radius, epsilon, C.byref(res), C.byref(rep))
        self._check(rc)
        out = {"lnl": res.lnl, "alpha": res.alpha, "tree_length": res.tree_length, "newick": C.string_at(res.newick).decode(),
               "replicates": C.string_at(rep).decode().splitlines() if rep else []}
        self.L.pml_result_free(C.byref(res))
        if rep:
            self.L.pml_free(rep)
        return out

    def parsimony(self, genes, seed=0, spr_radius=20):
        """`raxmlHPC -y` start trees (RAxMLRunner.java:215-251): list of {"newick" (topology only), "length"}."""
        keep = []
        n = len(genes)
        alns = (_lib.Alignment * n)(*[_aln_struct(g[0], g[1], keep) for g in genes])
        o = _lib.ParsimonyOpts(int(seed), int(spr_radius))
        res = (_lib.Result * n)()
        mp = (C.c_longlong * n)()
        rc = self.L.pml_parsimony_batch(self.ptr, n, alns, C.byref(o), res, mp)
        out = []
        if rc == 0:
            out = [{"newick": C.string_at(r.newick).decode(), "length": int(mp[i]), "npatterns": r.npatterns}
                   for i, r in enumerate(res)]
        for r in res:
            self.L.pml_result_free(C.byref(r))
        self._check(rc)
        return out

    def jackknife(self, genes, reps=100, subset_size=0, seed=0, spr_radius_full=5, epsilon=1e-3, alpha=1.0,
                  ncat=4, pi_mode=PI_RAXML_3DP, shard=(0, 1)):
        """Full tree + `reps` gene-subset support trees + support counts (PhylogenomicPipeline2.java:994-1126).
        genes: list of (names, rows), possibly over different taxon subsets."""
        keep = []
        n = len(genes)
        alns = (_lib.Alignment * n)(*[_aln_struct(g[0], g[1], keep) for g in genes])
        o = _lib.JackknifeOpts(reps, subset_size, seed, spr_radius_full, epsilon, int(shard[0]), int(shard[1]))
        m = _model(ncat, alpha, pi_mode)
        res = _lib.Result()
        sup = C.c_void_p()
        rc = self.L.pml_jackknife(self.ptr, n, alns, C.byref(m), C.byref(o), C.byref(res), C.byref(sup))
        self._check(rc)
        out = {"lnl": res.lnl, "alpha": res.alpha, "tree_length": res.tree_length, "npatterns": res.npatterns,
               "nsites": res.nsites, "newick": C.string_at(res.newick).decode() if res.newick else None,
               "support_trees": C.string_at(sup).decode().splitlines() if sup else []}
        self.L.pml_result_free(C.byref(res))
        if sup:
            self.L.pml_free(sup)
        return out

    def debug_gather(self, genes, sel=None):
        """Test hook: the replicate code matrix k_gather builds on the device for the gene selection,
        read back -> (names, codes uint8[ntax, npat], weights float64[npat])."""
        import numpy as np
        keep = []
        n = len(genes)
        alns = (_lib.Alignment * n)(*[_aln_struct(g[0], g[1], keep) for g in genes])
        nt, npat, mpad = C.c_int(), C.c_int(), C.c_int()
        codes, w, names = C.c_void_p(), C.c_void_p(), C.c_void_p()
        arr = (C.c_int * len(sel))(*sel) if sel is not None else None
        rc = self.L.pml_debug_gather(self.ptr, n, alns, len(sel) if sel is not None else 0, arr, C.byref(nt), C.byref(npat),
                                     C.byref(mpad), C.byref(codes), C.byref(w), C.byref(names))
        self._check(rc)
        cm = np.frombuffer(C.string_at(codes, nt.value * mpad.value), dtype=np.uint8).reshape(nt.value, mpad.value).copy()
        wv = np.frombuffer(C.string_at(w, 8 * mpad.value), dtype=np.float64).copy()
        nm = C.string_at(names).decode().splitlines()
        for p in (codes, w, names):
            self.L.pml_free(p)
        assert np.all(wv[npat.value:] == 0) and np.all(cm[:, npat.value:] == 22), "padding patterns must be weightless gaps"
        return nm, cm[:, :npat.value], wv[:npat.value]

    # ---- single-gene calls (what one Java thread issues); concurrent ones are coalesced inside the library ----
    def _single(self, fn, gene, newick, model, extra):
        keep = []
        a = _aln_struct(gene[0], gene[1], keep)
        res = _lib.Result()
        rc = fn(self.ptr, C.byref(a), newick.encode() if newick is not None else None, C.byref(model), *extra, C.byref(res))
        d = None
        if rc == 0:
            d = {"lnl": res.lnl, "alpha": res.alpha, "tree_length": res.tree_length, "npatterns": res.npatterns,
                 "nsites": res.nsites, "newick": C.string_at(res.newick).decode() if res.newick else None}
        self.L.pml_result_free(C.byref(res))
        self._check(rc)
        return d

    def score_one(self, gene, newick, alpha=1.0, ncat=4, pi_mode=PI_RAXML_3DP):
        return self._single(self.L.pml_score, gene, newick, _model(ncat, alpha, pi_mode), (0,))

    def optimize_one(self, gene, newick, alpha=1.0, ncat=4, pi_mode=PI_RAXML_3DP, optimize_alpha=True, epsilon=1e-4):
        o = _opts(optimize_alpha, False, 0, epsilon)
        return self._single(self.L.pml_optimize, gene, newick, _model(ncat, alpha, pi_mode), (C.byref(o),))

    def search_one(self, gene, start=None, alpha=1.0, ncat=4, pi_mode=PI_RAXML_3DP, optimize_alpha=True, nni=True,
                   spr_radius=0, epsilon=1e-3):
        o = _opts(optimize_alpha, nni, spr_radius, epsilon)
        return self._single(self.L.pml_search, gene, start, _model(ncat, alpha, pi_mode), (C.byref(o),))

    def coalescing_stats(self):
        b, r = C.c_longlong(), C.c_longlong()
        self._check(self.L.pml_coalescing_stats(self.ptr, C.byref(b), C.byref(r)))
        return {"batches": b.value, "requests": r.value}

    def newton_fallbacks(self):
        """how often k_newton's bounded exchange wait gave up on this context and work was re-issued through the no-exchange form"""
        a, b, c = C.c_longlong(), C.c_longlong(), C.c_longlong()
        self._check(self.L.pml_newton_fallbacks(self.ptr, C.byref(a), C.byref(b), C.byref(c)))
        return {"giveups": a.value, "reissued": b.value, "seq_launches": c.value}

    def kernel_stats(self, reset=False):
        out = {}
        for name, k in KERNELS.items():
            n, ms, by = C.c_longlong(), C.c_double(), C.c_double()
            self._check(self.L.pml_kernel_stats(self.ptr, k, C.byref(n), C.byref(ms), C.byref(by)))
            fl = C.c_double()
            self._check(self.L.pml_kernel_flops(self.ptr, k, C.byref(fl)))
            out[name] = {"launches": n.value, "ms": ms.value, "algo_bytes": by.value, "algo_flops": fl.value}
        if reset:
            self.L.pml_kernel_stats_reset(self.ptr)
        return out


class Batch:
    """Genes encoded and resident in HBM; repeated evaluation without host transfers."""

    def __init__(self, ctx, genes, newicks=None, alpha=1.0, ncat=4, pi_mode=PI_RAXML_3DP):
        self.ctx, self.L = ctx, ctx.L
        keep = []
        n = len(genes)
        alns = (_lib.Alignment * n)(*[_aln_struct(g[0], g[1], keep) for g in genes])
        nw = None
        if newicks is not None:
            nw = (C.c_char_p * n)(*[(s.encode() if s is not None else None) for s in newicks])
        m = _model(ncat, alpha, pi_mode)
        self.ptr = C.c_void_p()
        ctx._check(self.L.pml_batch_create(ctx.ptr, n, alns, nw, C.byref(m), C.byref(self.ptr)))
        self.n = n

    def close(self):
        if self.ptr:
            self.L.pml_batch_destroy(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def npatterns(self):
        return [self.L.pml_batch_npatterns(self.ptr, g) for g in range(self.n)]

    def score(self, stored=False):
        """full post-order pass + root evaluation of every gene; stored=True writes every CLV (the traversal a search runs)"""
        out = np.zeros(self.n)
        fn = self.L.pml_batch_score_stored if stored else self.L.pml_batch_score
        self.ctx._check(fn(self.ptr, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def site_lnl(self, g, nsites):
        out = np.zeros(max(nsites, 1))
        self.ctx._check(self.L.pml_batch_site_lnl(self.ptr, g, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out[:nsites]

    def set_alpha(self, alpha, g=-1):
        self.ctx._check(self.L.pml_batch_set_alpha(self.ptr, g, alpha))

    def root_derivs(self):
        a, b, c = np.zeros(self.n), np.zeros(self.n), np.zeros(self.n)
        dp = C.POINTER(C.c_double)
        self.ctx._check(self.L.pml_batch_root_derivs(self.ptr, a.ctypes.data_as(dp), b.ctypes.data_as(dp), c.ctypes.data_as(dp)))
        return a, b, c

    def optimize(self, optimize_alpha=True, epsilon=1e-4):
        o = _opts(optimize_alpha, False, 0, epsilon)
        lnl, al = np.zeros(self.n), np.zeros(self.n)
        dp = C.POINTER(C.c_double)
        self.ctx._check(self.L.pml_batch_optimize(self.ptr, C.byref(o), lnl.ctypes.data_as(dp), al.ctypes.data_as(dp)))
        return lnl, al

    def search(self, optimize_alpha=True, nni=True, spr_radius=0, epsilon=1e-3, constraints=None):
        o = _opts(optimize_alpha, nni, spr_radius, epsilon, constraints=constraints)
        lnl, al = np.zeros(self.n), np.zeros(self.n)
        dp = C.POINTER(C.c_double)
        self.ctx._check(self.L.pml_batch_search(self.ptr, C.byref(o), lnl.ctypes.data_as(dp), al.ctypes.data_as(dp)))
        return lnl, al

    def newick(self, g, digits=10):
        p = C.c_void_p()
        self.ctx._check(self.L.pml_batch_newick(self.ptr, g, digits, C.byref(p)))
        s = C.string_at(p).decode()
        self.L.pml_free(p)
        return s


def rf_distance(newick_a, newick_b):
    L = _lib.load()
    rf = C.c_int()
    rc = L.pml_rf_distance(newick_a.encode(), newick_b.encode(), C.byref(rf))
    if rc:
        raise PmlError(rc, L.pml_last_error(None).decode())
    return rf.value


def fpenv_seen():
    """MXCSR control states callers entered the library with (diagnostic; the library computes under the default state)."""
    L = _lib.load()
    if not hasattr(L, "pml_debug_fpenv"):          # an older build loaded through PEPRML_LIB (A/B runs)
        return []
    v = (C.c_uint * 16)()
    n = L.pml_debug_fpenv(v, 16)
    return [int(x) for x in v[:min(n, 16)]]


def refine_next(supported_newick, cutoff=100, done=()):
    """Next clade to refine (PhylogeneticTreeRefiner.getNextIndexToRefine :298-359): (sorted leaf list or None,
    floor(mean descendant support) per node in order of appearance)."""
    L = _lib.load()
    n = len(done)
    arr = (C.c_char_p * max(n, 1))(*[",".join(sorted(d)).encode() for d in done]) if n else None
    p, nn, mp = C.c_void_p(), C.c_int(), C.c_void_p()
    rc = L.pml_refine_next(supported_newick.encode(), cutoff, n, arr, C.byref(p), C.byref(nn), C.byref(mp))
    if rc:
        raise PmlError(rc, L.pml_last_error(None).decode())
    ingroup = C.string_at(p).decode().split(",") if p else None
    means = list(C.cast(mp, C.POINTER(C.c_int))[:nn.value]) if mp else []
    if p:
        L.pml_free(p)
    if mp:
        L.pml_free(mp)
    return ingroup, means


def support_tree(main_newick, support_newicks, digits=6):
    """Main tree decorated with integer bipartition counts (TreeSupportDecorator.addSupportValues)."""
    L = _lib.load()
    n = len(support_newicks)
    arr = (C.c_char_p * max(n, 1))(*[s.encode() for s in support_newicks]) if n else None
    p = C.c_void_p()
    rc = L.pml_support_tree(main_newick.encode(), n, arr, digits, C.byref(p))
    if rc:
        raise PmlError(rc, L.pml_last_error(None).decode())
    s = C.string_at(p).decode()
    L.pml_free(p)
    return s


def jackknife_draw(ngenes, reps, subset_size=0, seed=0):
    """The gene subsets pml_jackknife draws (host only): list of `reps` ascending index lists."""
    L = _lib.load()
    k = subset_size if subset_size > 0 else max(1, ngenes // 2)
    k = max(1, min(k, ngenes))
    out = (C.c_int * (reps * k))()
    rc = L.pml_jackknife_draw(ngenes, reps, subset_size, seed, out)
    if rc < 0:
        raise PmlError(rc)
    assert rc == k
    return [list(out[r * k:(r + 1) * k]) for r in range(reps)]


def concatenate(genes, sel=None):
    """Sorted-taxon-union concatenation with '?' padding (MSAConcatenator.java:78-189) -> (names, rows)."""
    L = _lib.load()
    keep = []
    n = len(genes)
    alns = (_lib.Alignment * n)(*[_aln_struct(g[0], g[1], keep) for g in genes])
    p = C.c_void_p()
    if sel is None:
        rc = L.pml_concatenate(n, alns, 0, None, C.byref(p))
    else:
        arr = (C.c_int * len(sel))(*sel)
        rc = L.pml_concatenate(n, alns, len(sel), arr, C.byref(p))
    if rc:
        raise PmlError(rc)
    txt = C.string_at(p).decode()
    L.pml_free(p)
    lines = txt.splitlines()
    return [l[1:] for l in lines[0::2]], lines[1::2]
