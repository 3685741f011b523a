"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by pepr_amd.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.po_model_init.argtypes = [vp, C.c_int]
        L.po_pmatrix.argtypes = [vp, C.c_double, dp]
        L.po_model_init_freqs.argtypes = [vp, dp]
        L.po_empirical_freqs.argtypes = [vp, dp]
        L.po_gamma_rates.argtypes = [C.c_double, C.c_int, C.c_int, dp]
        L.po_incgamma.restype = C.c_double
        L.po_incgamma.argtypes = [C.c_double, C.c_double]
        L.po_gamma_quantile.restype = C.c_double
        L.po_gamma_quantile.argtypes = [C.c_double, C.c_double]
        L.po_wag_tables.argtypes = [dp, dp, dp]
        L.po_aln_create.restype = vp
        L.po_aln_create.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int]
        L.po_aln_free.argtypes = [vp]
        L.po_tree_parse.restype = vp
        L.po_tree_parse.argtypes = [C.c_char_p, vp, C.c_char_p, C.c_int]
        L.po_tree_free.argtypes = [vp]
        L.po_tree_copy.restype = vp
        L.po_tree_copy.argtypes = [vp]
        L.po_tree_newick.restype = vp
        L.po_tree_newick.argtypes = [vp, vp, C.c_int]
        L.po_tree_rf.argtypes = [vp, vp]
        L.po_tree_length.restype = C.c_double
        L.po_tree_length.argtypes = [vp]
        L.po_engine_create.restype = vp
        L.po_engine_create.argtypes = [vp, vp, C.c_int, C.c_double]
        L.po_engine_free.argtypes = [vp]
        L.po_engine_set_alpha.argtypes = [vp, C.c_double]
        L.po_engine_alpha.restype = C.c_double
        L.po_engine_alpha.argtypes = [vp]
        L.po_engine_lnl.restype = C.c_double
        L.po_engine_lnl.argtypes = [vp, vp, dp]
        L.po_engine_site_lnl.restype = C.c_double
        L.po_engine_site_lnl.argtypes = [vp, vp, dp]
        L.po_engine_optimize.restype = C.c_double
        L.po_engine_optimize.argtypes = [vp, vp, C.c_int, C.c_double]
        L.po_engine_branch_derivs.argtypes = [vp, vp, C.c_int, C.c_int, dp, dp, dp]
        L.po_bruteforce_lnl.restype = C.c_double
        L.po_bruteforce_lnl.argtypes = [vp, vp, C.c_int, C.c_double, vp]
        L.po_engine_search.restype = C.c_double
        L.po_engine_search.argtypes = [vp, C.POINTER(vp), C.c_int, C.c_double]
        L.po_nj_tree.restype = vp
        L.po_nj_tree.argtypes = [vp]
        L.po_engine_set_constraints.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]
        L.po_nj_tree_constrained.restype = vp
        L.po_nj_tree_constrained.argtypes = [vp]
        L.po_engine_tree_displays.argtypes = [vp, vp]
        L.po_gamma20.restype = C.c_double
        L.po_gamma20.argtypes = [vp, vp, vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.po_g20_weights.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double)]
        L.po_g20_rates.argtypes = [C.POINTER(C.c_double)]
        L.po_engine_sh_support.restype = C.c_int
        L.po_engine_sh_support.argtypes = [vp, vp, C.c_int, C.c_ulonglong, C.POINTER(C.c_double)]
        L.po_parsimony_length.restype = C.c_longlong
        L.po_parsimony_length.argtypes = [vp, vp]
        L.po_parsimony_tree.restype = vp
        L.po_parsimony_tree.argtypes = [vp, C.c_uint, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_int)]
        L.free = C.CDLL(None).free
        L.free.argtypes = [vp]
        _LIB = L
    return _LIB


class _ModelStruct(C.Structure):
    _fields_ = [("pi", C.c_double * 20), ("Q", C.c_double * 400), ("eval", C.c_double * 20),
                ("U", C.c_double * 400), ("Uinv", C.c_double * 400)]


PI_RAXML3DP, PI_FULL = 0, 1


class Model:
    def __init__(self, pi_mode=PI_RAXML3DP, pi=None):
        """pi_mode: RAxML's 3-decimal or FastTree's full-precision WAG frequencies; pi: 20 explicit frequencies instead
        (PROTGAMMAWAGF: pass empirical_freqs(alignment))"""
        self.s = _ModelStruct()
        if pi is not None:
            arr = np.ascontiguousarray(pi, dtype=np.float64)
            lib().po_model_init_freqs(C.byref(self.s), arr.ctypes.data_as(C.POINTER(C.c_double)))
        else:
            lib().po_model_init(C.byref(self.s), pi_mode)
        self.ptr = C.cast(C.byref(self.s), C.c_void_p)
        self.pi = np.array(self.s.pi)
        self.Q = np.array(self.s.Q).reshape(20, 20)
        self.eval = np.array(self.s.eval)
        self.U = np.array(self.s.U).reshape(20, 20)
        self.Uinv = np.array(self.s.Uinv).reshape(20, 20)

    def pmatrix(self, t):
        out = np.zeros(400)
        lib().po_pmatrix(self.ptr, t, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out.reshape(20, 20)


def empirical_freqs(aln):
    """RAxML "F" model frequencies of an Alignment (numpy array of 20)"""
    out = np.zeros(20)
    lib().po_empirical_freqs(aln.ptr, out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


def gamma_rates(alpha, K=4, median=False):
    out = np.zeros(K)
    lib().po_gamma_rates(alpha, K, int(median), out.ctypes.data_as(C.POINTER(C.c_double)))
    return out


class _AlnStruct(C.Structure):
    _fields_ = [("ntax", C.c_int), ("nsites", C.c_int), ("npat", C.c_int), ("names", C.c_void_p),
                ("codes", C.POINTER(C.c_ubyte)), ("weight", C.POINTER(C.c_int)), ("site2pat", C.POINTER(C.c_int))]


class Alignment:
    def __init__(self, names, rows, compress=True):
        n = len(names)
        self.names = list(names)
        L = len(rows[0]) if n else 0
        assert all(len(r) == L for r in rows)
        na = (C.c_char_p * n)(*[s.encode() for s in names])
        ra = (C.c_char_p * n)(*[s.encode() for s in rows])
        self.ptr = C.c_void_p(lib().po_aln_create(n, L, na, ra, int(compress)))
        s = C.cast(self.ptr, C.POINTER(_AlnStruct)).contents
        self.ntax, self.nsites, self.npat = s.ntax, s.nsites, s.npat
        self.weight = np.ctypeslib.as_array(s.weight, shape=(max(self.npat, 1),))[:self.npat].copy()
        self.site2pat = np.ctypeslib.as_array(s.site2pat, shape=(max(self.nsites, 1),))[:self.nsites].copy()

    def __del__(self):
        try:
            lib().po_aln_free(self.ptr)
        except Exception:
            pass


class Tree:
    def __init__(self, newick=None, aln=None, ptr=None):
        self.aln = aln
        if ptr is not None:
            self.ptr = C.c_void_p(ptr)
            return
        err = C.create_string_buffer(256)
        p = lib().po_tree_parse(newick.encode(), aln.ptr, err, 256)
        if not p:
            raise ValueError("tree parse: " + err.value.decode())
        self.ptr = C.c_void_p(p)

    def newick(self, digits=10):
        p = lib().po_tree_newick(self.ptr, self.aln.ptr, digits)
        s = C.string_at(p).decode()
        lib().free(p)
        return s

    def copy(self):
        return Tree(aln=self.aln, ptr=lib().po_tree_copy(self.ptr))

    def rf(self, other):
        return lib().po_tree_rf(self.ptr, other.ptr)

    def length(self):
        return lib().po_tree_length(self.ptr)

    def __del__(self):
        try:
            lib().po_tree_free(self.ptr)
        except Exception:
            pass


class Engine:
    def __init__(self, aln, model, ncat=4, alpha=1.0):
        self.aln, self.model = aln, model
        self.ptr = C.c_void_p(lib().po_engine_create(aln.ptr, model.ptr, ncat, alpha))

    def set_alpha(self, a):
        lib().po_engine_set_alpha(self.ptr, a)

    @property
    def alpha(self):
        return lib().po_engine_alpha(self.ptr)

    def lnl(self, tree, patterns=False):
        if patterns:
            out = np.zeros(max(self.aln.npat, 1))
            v = lib().po_engine_lnl(self.ptr, tree.ptr, out.ctypes.data_as(C.POINTER(C.c_double)))
            return v, out[:self.aln.npat]
        return lib().po_engine_lnl(self.ptr, tree.ptr, None)

    def site_lnl(self, tree):
        out = np.zeros(max(self.aln.nsites, 1))
        v = lib().po_engine_site_lnl(self.ptr, tree.ptr, out.ctypes.data_as(C.POINTER(C.c_double)))
        return v, out[:self.aln.nsites]

    def optimize(self, tree, opt_alpha=True, eps=1e-4):
        return lib().po_engine_optimize(self.ptr, tree.ptr, int(opt_alpha), eps)

    def branch_derivs(self, tree, u, v):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        lib().po_engine_branch_derivs(self.ptr, tree.ptr, u, v, C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def sh_support(self, tree, nboot=1000, seed=314159):
        """SH-like local supports in internal-edge order (numpy array)."""
        out = np.zeros(max(self.aln.ntax - 3, 1))
        k = lib().po_engine_sh_support(self.ptr, tree.ptr, nboot, seed, out.ctypes.data_as(C.POINTER(C.c_double)))
        return out[:k]

    def set_constraints(self, names, rows):
        """FastTree -constraints matrix: one row of '0' '1' '-' per named taxon, one column per split (hard constraints)."""
        n = len(names)
        na = (C.c_char_p * n)(*[s.encode() for s in names]); ra = (C.c_char_p * n)(*[s.encode() for s in rows])
        lib().po_engine_set_constraints(self.ptr, len(rows[0]) if n else 0, n, na, ra)

    def displays(self, tree):
        return bool(lib().po_engine_tree_displays(self.ptr, tree.ptr))

    def nj_constrained(self):
        return Tree(aln=self.aln, ptr=lib().po_nj_tree_constrained(self.ptr))

    def search(self, start=None, spr_radius=0, eps=1e-3):
        """NJ start (or a copy of `start`), NNI hill climbing; returns (lnL, Tree)."""
        p = C.c_void_p(lib().po_tree_copy(start.ptr) if start is not None else None)
        lnl = lib().po_engine_search(self.ptr, C.byref(p), spr_radius, eps)
        return lnl, Tree(aln=self.aln, ptr=p.value)

    def __del__(self):
        try:
            lib().po_engine_free(self.ptr)
        except Exception:
            pass


def gamma20(aln, model, tree, table=False):
    """FastTree's -gamma step: (Gamma20 lnL, alpha, rescale[, npat x 20 per-pattern ln likelihoods])."""
    a, r = C.c_double(), C.c_double()
    tab = np.zeros((max(aln.npat, 1), 20)) if table else None
    v = lib().po_gamma20(aln.ptr, model.ptr, tree.ptr, C.byref(a), C.byref(r), tab.ctypes.data_as(C.POINTER(C.c_double)) if table else None)
    return (v, a.value, r.value, tab[:aln.npat]) if table else (v, a.value, r.value)


def g20_weights(alpha, mult):
    w = np.zeros(20)
    lib().po_g20_weights(alpha, mult, w.ctypes.data_as(C.POINTER(C.c_double)))
    return w


def g20_rates():
    r = np.zeros(20)
    lib().po_g20_rates(r.ctypes.data_as(C.POINTER(C.c_double)))
    return r


def nj_tree(aln):
    return Tree(aln=aln, ptr=lib().po_nj_tree(aln.ptr))


def bruteforce_lnl(aln, model, tree, ncat=4, alpha=1.0):
    return lib().po_bruteforce_lnl(aln.ptr, model.ptr, ncat, alpha, tree.ptr)


def parsimony_length(aln, tree):
    return lib().po_parsimony_length(aln.ptr, tree.ptr)


def parsimony_tree(aln, seed=0, radius=20):
    """-> (Tree, weighted Fitch length, number of SPR moves applied)"""
    ln, mv = C.c_longlong(), C.c_int()
    p = lib().po_parsimony_tree(aln.ptr, seed, radius, C.byref(ln), C.byref(mv))
    return Tree(aln=aln, ptr=p), ln.value, mv.value
