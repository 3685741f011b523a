"""ctypes loader for pepr_amd/libpeprml.so (the C ABI of include/peprml.h).

There is no Python or CPU fallback: if the shared library is missing this raises, and every
likelihood is computed by the HIP kernels inside it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PEPRML_LIB: another build of the same library (same-box A/B of kernel variants, tools/bench_ab.sh); never a fallback
SO_PATH = os.environ.get("PEPRML_LIB") or os.path.join(_HERE, "libpeprml.so")

# every symbol include/peprml.h declares (tests check the library exports all of them)
SYMBOLS = [
    "pml_create", "pml_destroy", "pml_strerror", "pml_last_error", "pml_version",
    "pml_score", "pml_optimize", "pml_search",
    "pml_score_batch", "pml_optimize_batch", "pml_search_batch", "pml_result_free",
    "pml_batch_create", "pml_batch_destroy", "pml_batch_size", "pml_batch_npatterns",
    "pml_batch_score", "pml_batch_score_stored", "pml_batch_site_lnl", "pml_batch_set_alpha", "pml_batch_optimize",
    "pml_batch_search", "pml_batch_newick", "pml_batch_root_derivs", "pml_free",
    "pml_rf_distance", "pml_support_tree", "pml_jackknife", "pml_jackknife_draw", "pml_debug_gather", "pml_concatenate", "pml_parsimony", "pml_parsimony_batch", "pml_refine_next", "pml_bootstrap", "pml_coalescing_stats", "pml_newton_fallbacks", "pml_sh_support", "pml_sh_support_batch", "pml_gamma20", "pml_gamma20_batch", "pml_debug_fpenv", "pml_kernel_stats", "pml_kernel_flops", "pml_kernel_stats_reset",
]


class Config(C.Structure):
    _fields_ = [("device", C.c_int), ("profile", C.c_int), ("arena_bytes", C.c_size_t)]


class Alignment(C.Structure):
    _fields_ = [("ntax", C.c_int), ("nsites", C.c_int),
                ("names", C.POINTER(C.c_char_p)), ("rows", C.POINTER(C.c_char_p))]


class Model(C.Structure):
    _fields_ = [("ncat", C.c_int), ("alpha", C.c_double), ("pi_mode", C.c_int)]


class SearchOpts(C.Structure):
    _fields_ = [("optimize_alpha", C.c_int), ("nni", C.c_int), ("spr_radius", C.c_int),
                ("epsilon", C.c_double), ("seed", C.c_uint),
                ("nconstraints", C.c_int), ("constraint_ntax", C.c_int),
                ("constraint_names", C.POINTER(C.c_char_p)), ("constraint_rows", C.POINTER(C.c_char_p))]


class JackknifeOpts(C.Structure):
    _fields_ = [("reps", C.c_int), ("subset_size", C.c_int), ("seed", C.c_ulonglong),
                ("spr_radius_full", C.c_int), ("epsilon", C.c_double),
                ("shard_rank", C.c_int), ("shard_world", C.c_int)]


class ParsimonyOpts(C.Structure):
    _fields_ = [("seed", C.c_uint), ("spr_radius", C.c_int)]


class Result(C.Structure):
    _fields_ = [("status", C.c_int), ("lnl", C.c_double), ("alpha", C.c_double),
                ("tree_length", C.c_double), ("npatterns", C.c_int), ("nsites", C.c_int),
                ("newick", C.c_void_p), ("site_lnl", C.POINTER(C.c_double))]


_lib = None


def load():
    """Returns the loaded library; raises OSError with a clear message when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise OSError("pepr_amd/libpeprml.so is not built -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(SO_PATH)
    vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)
    L.pml_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.pml_destroy.argtypes = [vp]
    L.pml_destroy.restype = None
    L.pml_strerror.restype = C.c_char_p
    L.pml_strerror.argtypes = [C.c_int]
    L.pml_last_error.restype = C.c_char_p
    L.pml_last_error.argtypes = [vp]
    L.pml_version.restype = C.c_char_p
    ap, mp, sp, rp = C.POINTER(Alignment), C.POINTER(Model), C.POINTER(SearchOpts), C.POINTER(Result)
    L.pml_score.argtypes = [vp, ap, C.c_char_p, mp, C.c_int, rp]
    L.pml_optimize.argtypes = [vp, ap, C.c_char_p, mp, sp, rp]
    L.pml_search.argtypes = [vp, ap, C.c_char_p, mp, sp, rp]
    cpp = C.POINTER(C.c_char_p)
    L.pml_score_batch.argtypes = [vp, C.c_int, ap, cpp, mp, C.c_int, rp]
    L.pml_optimize_batch.argtypes = [vp, C.c_int, ap, cpp, mp, sp, rp]
    L.pml_search_batch.argtypes = [vp, C.c_int, ap, cpp, mp, sp, rp]
    L.pml_result_free.argtypes = [rp]
    L.pml_result_free.restype = None
    L.pml_batch_create.argtypes = [vp, C.c_int, ap, cpp, mp, C.POINTER(vp)]
    L.pml_batch_destroy.argtypes = [vp]
    L.pml_batch_destroy.restype = None
    L.pml_batch_size.argtypes = [vp]
    L.pml_batch_npatterns.argtypes = [vp, C.c_int]
    L.pml_batch_score.argtypes = [vp, dp]
    L.pml_batch_score_stored.argtypes = [vp, dp]
    L.pml_batch_site_lnl.argtypes = [vp, C.c_int, dp]
    L.pml_batch_set_alpha.argtypes = [vp, C.c_int, C.c_double]
    L.pml_batch_optimize.argtypes = [vp, sp, dp, dp]
    L.pml_batch_search.argtypes = [vp, sp, dp, dp]
    L.pml_batch_newick.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.pml_batch_root_derivs.argtypes = [vp, dp, dp, dp]
    L.pml_free.argtypes = [vp]
    L.pml_free.restype = None
    L.pml_rf_distance.argtypes = [C.c_char_p, C.c_char_p, ip]
    L.pml_support_tree.argtypes = [C.c_char_p, C.c_int, cpp, C.c_int, C.POINTER(vp)]
    L.pml_jackknife.argtypes = [vp, C.c_int, ap, mp, C.POINTER(JackknifeOpts), rp, C.POINTER(vp)]
    L.pml_jackknife_draw.argtypes = [C.c_int, C.c_int, C.c_int, C.c_ulonglong, ip]
    L.pml_debug_gather.argtypes = [vp, C.c_int, ap, C.c_int, ip, ip, ip, ip, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.pml_parsimony.argtypes = [vp, ap, C.POINTER(ParsimonyOpts), rp, C.POINTER(C.c_longlong)]
    L.pml_parsimony_batch.argtypes = [vp, C.c_int, ap, C.POINTER(ParsimonyOpts), rp, C.POINTER(C.c_longlong)]
    L.pml_refine_next.argtypes = [C.c_char_p, C.c_int, C.c_int, cpp, C.POINTER(vp), ip, C.POINTER(vp)]
    L.pml_bootstrap.argtypes = [vp, ap, mp, C.c_int, C.c_ulonglong, C.c_int, C.c_double, rp, C.POINTER(vp)]
    L.pml_sh_support.argtypes = [vp, ap, C.c_char_p, mp, C.c_int, C.c_ulonglong, rp]
    L.pml_sh_support_batch.argtypes = [vp, C.c_int, ap, cpp, mp, C.c_int, C.c_ulonglong, rp]
    L.pml_gamma20.argtypes = [vp, ap, C.c_char_p, mp, rp, dp]
    L.pml_gamma20_batch.argtypes = [vp, C.c_int, ap, cpp, mp, rp, dp]
    L.pml_debug_fpenv.argtypes = [C.POINTER(C.c_uint), C.c_int]
    L.pml_concatenate.argtypes = [C.c_int, ap, C.c_int, ip, C.POINTER(vp)]
    L.pml_kernel_stats.argtypes = [vp, C.c_int, C.POINTER(C.c_longlong), dp, dp]
    L.pml_kernel_flops.argtypes = [vp, C.c_int, dp]
    L.pml_kernel_stats_reset.argtypes = [vp]
    L.pml_newton_fallbacks.argtypes = [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.pml_coalescing_stats.argtypes = [vp, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    _lib = L
    return L
