// api.cpp -- the C ABI declared in include/peprml.h.  Nothing here computes likelihoods on the
// CPU: every numeric result comes from the HIP kernels; without a device pml_create() fails.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <thread>

#include "../../include/peprml.h"
#include "api_types.hpp"

using namespace pml;

static thread_local std::string g_err;

static std::mutex g_fp_mu; static unsigned g_fp_seen[16]; static int g_fp_n = 0;
void pml_fpguard::note(unsigned v) {
    v &= ~0x3Fu;                                   // control bits only (the low six are sticky exception flags)
    std::lock_guard<std::mutex> lk(g_fp_mu);
    for (int i = 0; i < g_fp_n; ++i) if (g_fp_seen[i] == v) return;
    if (g_fp_n < 16) g_fp_seen[g_fp_n++] = v;
}

extern "C" {

const char *pml_version(void) { return "peprml 0.2 (gfx950)"; }
/* diagnostic: the distinct MXCSR control states (exception flags masked out) callers entered the library with */
int pml_debug_fpenv(unsigned *values, int cap) {
    std::lock_guard<std::mutex> lk(g_fp_mu);
    for (int i = 0; i < g_fp_n && i < cap; ++i) values[i] = g_fp_seen[i];
    return g_fp_n;
}

const char *pml_strerror(int code) {
    switch (code) {
        case PML_OK: return "ok";
        case PML_EINVAL: return "invalid argument";
        case PML_EPARSE: return "parse error";
        case PML_ENODEVICE: return "no HIP device";
        case PML_ENOMEM: return "out of memory";
        case PML_EDEVICE: return "HIP runtime error";
        case PML_ENOTFOUND: return "not found";
        default: return "unknown error";
    }
}

const char *pml_last_error(pml_ctx *ctx) { return ctx ? ctx->c.last_error.c_str() : g_err.c_str(); }

int pml_create(const pml_config *cfg, pml_ctx **out) {
    if (!out) return PML_EINVAL;
    *out = nullptr;
    pml_ctx *c = new (std::nothrow) pml_ctx();
    if (!c) return PML_ENOMEM;
    const int rc = c->c.init(cfg ? cfg->device : 0, cfg ? cfg->profile != 0 : false);
    if (rc) { g_err = c->c.last_error; delete c; return rc; }
    if (cfg && cfg->arena_bytes > 0) {
        if (hipMalloc((void **)&c->c.arena_cache, cfg->arena_bytes) != hipSuccess) {
            g_err = "arena_bytes does not fit on the device"; c->c.destroy(); delete c; return PML_ENOMEM;
        }
        c->c.arena_cache_bytes = cfg->arena_bytes;
    }
    *out = c;
    return PML_OK;
}

void pml_destroy(pml_ctx *ctx) {
    if (!ctx) return;
    for (auto &w : ctx->workers) w->destroy();
    ctx->c.destroy();
    delete ctx;
}

void pml_free(void *p) { std::free(p); }

void pml_result_free(pml_result *r) {
    if (!r) return;
    std::free(r->newick); std::free(r->site_lnl);
    r->newick = nullptr; r->site_lnl = nullptr;
}

static char *dup_string(const std::string &s) {
    char *p = (char *)std::malloc(s.size() + 1);
    if (p) std::memcpy(p, s.c_str(), s.size() + 1);
    return p;
}

// ---- resident batches ---------------------------------------------------------------------
static int batch_create_impl(pml_ctx *ctx, int n, const pml_alignment *alns, const char *const *newicks,
                             const pml_model *model, bool score_only, pml_batch **out) {
    pml_fpguard fpg;
    if (!ctx || !out || !alns || n <= 0) return PML_EINVAL;
    *out = nullptr;
    pml_batch *b = new (std::nothrow) pml_batch();
    if (!b) return PML_ENOMEM;
    b->owner = ctx;
    pml_drop_worker_caches(ctx);
    static_assert(sizeof(pml_alignment) == sizeof(pml_alignment_view), "alignment view layout");
    const int ncat = model ? model->ncat : 4;
    const double alpha = model ? model->alpha : 1.0;
    const int pm = model ? model->pi_mode : PML_PI_RAXML_3DP;
    int rc;
    try {
        rc = b->b.create(&ctx->c, n, reinterpret_cast<const pml_alignment_view *>(alns), newicks, pm, ncat, alpha, score_only);
    } catch (const std::bad_alloc &) { rc = ctx->c.fail(PML_ENOMEM, "host allocation failed"); }
    catch (const std::exception &e) { rc = ctx->c.fail(PML_EINVAL, e.what()); }
    if (rc) { b->b.destroy(); delete b; return rc; }
    *out = b;
    return PML_OK;
}

int pml_batch_create(pml_ctx *ctx, int n, const pml_alignment *alns, const char *const *newicks,
                     const pml_model *model, pml_batch **out) {
    if (!ctx) return PML_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->c.mu);
    return batch_create_impl(ctx, n, alns, newicks, model, false, out);
}

void pml_batch_destroy(pml_batch *b) {
    if (!b) return;
    std::lock_guard<std::mutex> lk(b->owner->c.mu);
    b->b.destroy();
    delete b;
}

int pml_batch_size(const pml_batch *b) { return b ? (int)b->b.genes.size() : 0; }
int pml_batch_npatterns(const pml_batch *b, int g) {
    if (!b || g < 0 || g >= (int)b->b.genes.size()) return PML_EINVAL;
    return b->b.genes[g].aln.npat;
}

#define LOCKED(b) std::lock_guard<std::mutex> lk((b)->owner->c.mu)
#define GUARD(expr)                                                                            \
    try { return (expr); }                                                                     \
    catch (const std::bad_alloc &) { return b->owner->c.fail(PML_ENOMEM, "host allocation failed"); } \
    catch (const std::exception &e) { return b->owner->c.fail(PML_EINVAL, e.what()); }

int pml_batch_score(pml_batch *b, double *lnl) {
    if (!b || !lnl) return PML_EINVAL;
    pml_fpguard fpg;
    LOCKED(b);
    GUARD(b->b.score(std::vector<char>(), lnl));
}
int pml_batch_score_stored(pml_batch *b, double *lnl) {
    if (!b || !lnl) return PML_EINVAL;
    pml_fpguard fpg;
    LOCKED(b);
    GUARD(b->b.score(std::vector<char>(), lnl, true));
}
int pml_batch_site_lnl(pml_batch *b, int g, double *out) {
    if (!b || !out || g < 0 || g >= (int)b->b.genes.size()) return PML_EINVAL;
    pml_fpguard fpg;
    LOCKED(b);
    GUARD(b->b.site_lnl(g, out));
}
int pml_batch_set_alpha(pml_batch *b, int g, double alpha) {
    if (!b || g >= (int)b->b.genes.size() || !(alpha > 0)) return PML_EINVAL;
    pml_fpguard fpg;
    LOCKED(b);
    if (g < 0) for (int i = 0; i < (int)b->b.genes.size(); ++i) b->b.set_alpha(i, alpha);
    else b->b.set_alpha(g, alpha);
    return PML_OK;
}
int pml_batch_root_derivs(pml_batch *b, double *lnl, double *d1, double *d2) {
    if (!b || !lnl || !d1 || !d2) return PML_EINVAL;
    pml_fpguard fpg;
    LOCKED(b);
    GUARD(b->b.root_derivs(lnl, d1, d2));
}
int pml_batch_optimize(pml_batch *b, const pml_search_opts *opts, double *lnl, double *alpha) {
    if (!b || !lnl) return PML_EINVAL;
    pml_fpguard fpg;
    LOCKED(b);
    const bool oa = opts ? opts->optimize_alpha != 0 : true;
    const double eps = (opts && opts->epsilon > 0) ? opts->epsilon : 1e-4;
    int rc;
    try { rc = b->b.optimize(oa, eps, lnl); }
    catch (const std::exception &e) { return b->owner->c.fail(PML_EINVAL, e.what()); }
    if (rc) return rc;
    if (alpha) for (size_t g = 0; g < b->b.genes.size(); ++g) alpha[g] = b->b.genes[g].alpha;
    return PML_OK;
}
int pml_batch_search(pml_batch *b, const pml_search_opts *opts, double *lnl, double *alpha) {
    if (!b || !lnl) return PML_EINVAL;
    pml_fpguard fpg;
    LOCKED(b);
    int rc;
    try { rc = opts ? b->b.set_constraints(opts->nconstraints, opts->constraint_ntax, opts->constraint_names, opts->constraint_rows) : 0;
          if (!rc) rc = b->b.search(opts ? opts->nni != 0 : true, opts ? opts->spr_radius : 0,
                           opts ? opts->optimize_alpha != 0 : true,
                           (opts && opts->epsilon > 0) ? opts->epsilon : 1e-3, lnl); }
    catch (const std::exception &e) { return b->owner->c.fail(PML_EINVAL, e.what()); }
    if (rc) return rc;
    if (alpha) for (size_t g = 0; g < b->b.genes.size(); ++g) alpha[g] = b->b.genes[g].alpha;
    return PML_OK;
}
int pml_batch_newick(pml_batch *b, int g, int digits, char **out) {
    if (!b || !out || g < 0 || g >= (int)b->b.genes.size()) return PML_EINVAL;
    pml_fpguard fpg;
    LOCKED(b);
    const Gene &G = b->b.genes[g];
    *out = dup_string(G.tree.newick(G.aln.names, digits));
    return *out ? PML_OK : PML_ENOMEM;
}

// ---- one-shot wrappers --------------------------------------------------------------------
enum { OP_SCORE, OP_OPTIMIZE, OP_SEARCH };

// upper bound of the HBM a gene needs in a batch (patterns <= columns)
static size_t gene_bytes_bound(const pml_alignment &a, bool score_only) {
    const size_t mp = ((size_t)std::max(a.nsites, 1) + 31) / 32 * 32, nt = (size_t)std::max(a.ntax, 3);
    const size_t slots = (score_only ? nt - 2 : 3 * (nt - 2)) + NSCRATCH + MAXTAIL;
    return slots * CLV_ROWS * ((mp + 127) / 128 * 128) * 8 + slots * mp * 4 + nt * mp + 64 * mp + (1 << 16);
}

// one device batch on context `c` (the caller's own or one of its workers): start trees, the requested operation, results
static int oneshot_on(Ctx &c, int op, int n, const pml_alignment *alns, const char *const *newicks,
                      const pml_model *model, const pml_search_opts *opts, int flags, pml_result *const *out, int share) {
    pml_fpguard fpg;
    // RAxML starts `-f d` from a randomised stepwise-addition parsimony tree (-p seed): opts->seed != 0 asks for
    // that start for every gene without a given start tree (seed 0 = the deterministic NJ start)
    std::vector<std::string> pstart; std::vector<const char *> pnw;
    static_assert(sizeof(pml_alignment) == sizeof(pml_alignment_view), "alignment view layout");
    if (op == OP_SEARCH && opts && opts->seed != 0) {
        std::vector<int> idx;
        for (int i = 0; i < n; ++i) if (!newicks || !newicks[i]) idx.push_back(i);
        if (!idx.empty()) {
            std::vector<pml_alignment_view> views;
            for (int i : idx) views.push_back(pml_alignment_view{alns[i].ntax, alns[i].nsites, alns[i].names, alns[i].rows});
            std::vector<Tree> trees; std::vector<EncodedAlignment> enc; std::vector<long long> len; std::vector<int> moves;
            if (int prc = parsimony_batch(&c, (int)idx.size(), views.data(), opts->seed, 20, trees, enc, len, moves)) return prc;
            pstart.resize(n); pnw.assign(n, nullptr);
            for (int i = 0; i < n; ++i) if (newicks && newicks[i]) pnw[i] = newicks[i];
            for (size_t k = 0; k < idx.size(); ++k) { pstart[idx[k]] = trees[k].newick(enc[k].names, 6); pnw[idx[k]] = pstart[idx[k]].c_str(); }
            newicks = pnw.data();
        }
    }
    Batch b;
    struct Drop { Batch &b; ~Drop() { b.destroy(); } } drop{b};
    b.share = share;
    int rc;
    try {
        rc = b.create(&c, n, reinterpret_cast<const pml_alignment_view *>(alns), newicks, model ? model->pi_mode : PML_PI_RAXML_3DP,
                      model ? model->ncat : 4, model ? model->alpha : 1.0, op == OP_SCORE);
        if (rc) return rc;
        std::vector<double> lnl(n);
        if (op == OP_SCORE) rc = b.score(std::vector<char>(), lnl.data());
        else if (op == OP_OPTIMIZE)
            rc = b.optimize(opts ? opts->optimize_alpha != 0 : true, (opts && opts->epsilon > 0) ? opts->epsilon : 1e-4, lnl.data());
        else {
            rc = opts ? b.set_constraints(opts->nconstraints, opts->constraint_ntax, opts->constraint_names, opts->constraint_rows) : 0;
            if (!rc) rc = b.search(opts ? opts->nni != 0 : true, opts ? opts->spr_radius : 0, opts ? opts->optimize_alpha != 0 : true,
                                   (opts && opts->epsilon > 0) ? opts->epsilon : 1e-3, lnl.data());
        }
        for (int i = 0; i < n && !rc; ++i) {
            const Gene &G = b.genes[i];
            pml_result &r = *out[i];
            r.lnl = lnl[i]; r.alpha = G.alpha; r.tree_length = G.tree.length();
            r.npatterns = G.aln.npat; r.nsites = G.aln.nsites;
            r.newick = dup_string(G.tree.newick(G.aln.names, op == OP_SCORE ? 10 : 20));
            if (op == OP_SCORE && (flags & PML_WANT_SITE_LNL)) {
                r.site_lnl = (double *)std::malloc(sizeof(double) * (G.aln.nsites > 0 ? G.aln.nsites : 1));
                if (!r.site_lnl) rc = c.fail(PML_ENOMEM, "host allocation failed");
                else rc = b.site_lnl(i, r.site_lnl);
            }
        }
    } catch (const std::bad_alloc &) { rc = c.fail(PML_ENOMEM, "host allocation failed"); }
    catch (const std::exception &e) { rc = c.fail(PML_EINVAL, e.what()); }
    return rc;
}

// Worker contexts: a search is a chain of thousands of small dependent launches with host decisions in between (pass set-up,
// NNI / SPR candidate selection, Brent steps), so ONE batch leaves the device idle whenever its host thread thinks and
// latency-bound whenever a kernel of the chain is (four bench ranks forced onto one GPU searched 512 C3 genes at 181 / 118
// gene-trees/s against 171 / 75 for one rank alone, profiles/r03_rehearsal_4_ranks_on_1_gpu.json).  Genes are independent and
// a gene's arithmetic does not depend on what shares its batch, so a search call deals its genes over a few GROUPS, each a
// batch of its own on its own stream driven by its own host thread: one group's host work and latency-bound kernels overlap
// the others' device work.  Same bits as the undivided call (tests/test_gpu_parity.py: composition independence).
static Ctx *worker_ctx(pml_ctx *ctx, int k) {
    while ((int)ctx->workers.size() <= k) {
        std::unique_ptr<Ctx> w(new Ctx());
        if (w->init_worker(ctx->c)) { ctx->c.last_error = w->last_error; return nullptr; }
        ctx->workers.push_back(std::move(w));
    }
    return ctx->workers[k].get();
}
static int search_groups(int n, const pml_alignment *alns) {
    // PML_GROUPS overrides (1 = undivided).  Default: as many groups (<= 2) as keep >= 32 genes each, and only for genes of at
    // most 16 tiles (2048 patterns): the groups' fused-Newton kernels run side by side, and an XCD is only GUARANTEED room for 32
    // of their workgroups -- each group's newest, partially staffed gene must fit next to the others' (kernels.hip launch_oplist).
    // Measured (C3, 128 genes, gpurun_out/r3j_groups.txt): 1 group 171 gene-trees/s (RAxML path 74), 2 groups 195 (97), 3 groups
    // 215 (95); the groups' streams are of the highest priority class (their own pool of hardware queues).
    static const int env = std::getenv("PML_GROUPS") ? std::atoi(std::getenv("PML_GROUPS")) : 0;
    int maxcols = 0;
    for (int i = 0; i < n; ++i) maxcols = std::max(maxcols, alns[i].nsites);
    const int tiles = (maxcols + 127) / 128;
    int g = env > 0 ? std::min(env, 8) : 2;                // two by default: 3 measured 179-215 from run to run, 2 stays at 195-200
    if (env <= 0) { while (g > 1 && n / g < 32) --g; }
    if (!std::getenv("PML_GROUPS_FORCE")) { while (g > 1 && tiles * g > 32) --g; }       // (experiment switch: lifts the tile rule)
    return std::max(g, 1);
}

static int oneshot_chunk(pml_ctx *ctx, int op, int n, const pml_alignment *alns, const char *const *newicks,
                         const pml_model *model, const pml_search_opts *opts, int flags, pml_result *out) {
    const int G = (op == OP_SEARCH) ? search_groups(n, alns) : 1;
    if (G <= 1) {
        std::vector<pml_result *> outs(n);
        for (int i = 0; i < n; ++i) outs[i] = out + i;
        return oneshot_on(ctx->c, op, n, alns, newicks, model, opts, flags, outs.data(), 1);
    }
    // genes are dealt round-robin (the callers pass genes of similar size next to each other), group k on worker context k
    for (int k = 0; k < G; ++k) if (!worker_ctx(ctx, k)) return PML_EDEVICE;
    if (ctx->c.arena_cache) { hipFree(ctx->c.arena_cache); ctx->c.arena_cache = nullptr; ctx->c.arena_cache_bytes = 0; }   // the groups bring their own arenas
    std::vector<int> rcs(G, 0);
    std::vector<std::thread> th;
    for (int k = 0; k < G; ++k) th.emplace_back([&, k]() {
        std::vector<pml_alignment> a; std::vector<const char *> nw; std::vector<pml_result *> outs;
        for (int i = k; i < n; i += G) { a.push_back(alns[i]); nw.push_back(newicks ? newicks[i] : nullptr); outs.push_back(out + i); }
        try { rcs[k] = oneshot_on(*ctx->workers[k], op, (int)a.size(), a.data(), newicks ? nw.data() : nullptr, model, opts, flags, outs.data(), G); }
        catch (const std::exception &e) { rcs[k] = ctx->workers[k]->fail(PML_EINVAL, e.what()); }
    });
    for (auto &t : th) t.join();
    int rc = 0;
    for (int k = 0; k < G; ++k) {
        Ctx &w = *ctx->workers[k];
        if (rcs[k] && !rc) { rc = rcs[k]; ctx->c.last_error = w.last_error; }
        for (int i = 0; i < K_COUNT; ++i) {            // the workers' launch statistics count as the context's
            ctx->c.stats[i].launches += w.stats[i].launches; ctx->c.stats[i].ms += w.stats[i].ms; ctx->c.stats[i].bytes += w.stats[i].bytes; ctx->c.stats[i].flops += w.stats[i].flops;
            w.stats[i] = Ctx::KStat();
        }
        ctx->c.newton_giveups += w.newton_giveups; ctx->c.newton_reissued += w.newton_reissued; ctx->c.newton_seq_launches += w.newton_seq_launches;
        w.newton_giveups = w.newton_reissued = w.newton_seq_launches = 0;
    }
    return rc;
}

// one-shot batched call; gene lists that do not fit in free HBM at once are processed in
// consecutive sub-batches (e.g. BASELINE config C5: 250 genes x 500 taxa x 2000 sites per GPU)
static int oneshot(pml_ctx *ctx, int op, int n, const pml_alignment *alns, const char *const *newicks,
                   const pml_model *model, const pml_search_opts *opts, int flags, pml_result *out) {
    if (!ctx || !alns || !out || n <= 0) return PML_EINVAL;
    pml_fpguard fpg;                               // also when this thread is the leader running other callers' requests
    for (int i = 0; i < n; ++i) std::memset(&out[i], 0, sizeof(pml_result));
    if (op != OP_SEARCH) {
        if (!newicks) return ctx->c.fail(PML_EINVAL, "newick required");
        for (int i = 0; i < n; ++i) if (!newicks[i]) return ctx->c.fail(PML_EINVAL, "newick required");
    }
    std::lock_guard<std::mutex> lk(ctx->c.mu);
    size_t free_b = 0, total_b = 0;
    hipSetDevice(ctx->c.device);
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)1 << 40;
    free_b += ctx->c.arena_cache_bytes;            // the cached arena is reused or released by the next batch
    // the workers' cached arenas: reused by a grouped call, released before any other
    if (op == OP_SEARCH) { for (auto &w : ctx->workers) free_b += w->arena_cache_bytes; }
    else { for (auto &w : ctx->workers) free_b += w->arena_cache_bytes; pml_drop_worker_caches(ctx); }
    size_t budget = (size_t)(0.85 * (double)free_b);
    if (const char *e = std::getenv("PML_HBM_BUDGET_MB")) budget = (size_t)std::atoll(e) << 20;   // test hook
    int rc = 0, begin = 0;
    while (begin < n && !rc) {
        size_t used = 0; int end = begin;
        while (end < n) {
            const size_t need = gene_bytes_bound(alns[end], op == OP_SCORE);
            if (end > begin && used + need > budget) break;
            used += need; ++end;
        }
        rc = oneshot_chunk(ctx, op, end - begin, alns + begin, newicks ? newicks + begin : nullptr, model, opts, flags, out + begin);
        begin = end;
    }
    for (int i = 0; i < n; ++i) out[i].status = rc;
    return rc;
}

}  // extern "C"

// ---- coalescing of concurrent single-gene calls ---------------------------------------------
// PEPR calls the tree builder from up to `tree_threads` Java threads at once (PhylogenomicPipeline2.java:
// 1039-1054, 1233-1254), each blocking on ONE alignment.  A single small gene cannot fill 256 CUs, so instead of
// serialising such calls the first caller that finds no batch in flight becomes the leader: it takes every queued
// request with the same operation / model / options, runs them as one device batch and wakes their owners.
// Requests that arrive meanwhile form the next batch.  Blocking semantics and results are those of the
// single call (a gene's arithmetic does not depend on what shares its batch: tests/test_gpu_parity.py).
struct pml_request {
    int op, flags; const pml_alignment *aln; const char *nw;
    pml_model model; pml_search_opts opts; pml_result *out;
    int rc = 0; bool done = false; std::string err;
};
static bool compatible(const pml_request &a, const pml_request &b) {
    return a.op == b.op && a.flags == b.flags && a.model.ncat == b.model.ncat && a.model.alpha == b.model.alpha &&
           a.model.pi_mode == b.model.pi_mode && a.opts.optimize_alpha == b.opts.optimize_alpha && a.opts.nni == b.opts.nni &&
           a.opts.spr_radius == b.opts.spr_radius && a.opts.epsilon == b.opts.epsilon && a.opts.seed == b.opts.seed;
}
static void run_group(pml_ctx *ctx, std::vector<pml_request *> &grp) {
    const int n = (int)grp.size();
    std::vector<pml_alignment> alns(n); std::vector<const char *> nws(n); std::vector<pml_result> res(n);
    bool any_nw = false;
    for (int i = 0; i < n; ++i) { alns[i] = *grp[i]->aln; nws[i] = grp[i]->nw; any_nw |= grp[i]->nw != nullptr; }
    pml_request &f = *grp[0];
    int rc = oneshot(ctx, f.op, n, alns.data(), any_nw ? nws.data() : nullptr, &f.model, &f.opts, f.flags, res.data());
    if (rc && n > 1) {                      // one bad input must not fail its neighbours: redo one by one
        for (int i = 0; i < n; ++i) {
            pml_result_free(&res[i]);
            grp[i]->rc = oneshot(ctx, f.op, 1, &alns[i], nws[i] ? &nws[i] : nullptr, &f.model, &f.opts, f.flags, grp[i]->out);
            if (grp[i]->rc) { std::lock_guard<std::mutex> g(ctx->c.mu); grp[i]->err = ctx->c.last_error; }
        }
        return;
    }
    std::string err;
    if (rc) { std::lock_guard<std::mutex> g(ctx->c.mu); err = ctx->c.last_error; }
    for (int i = 0; i < n; ++i) { *grp[i]->out = res[i]; grp[i]->rc = rc; grp[i]->err = err; }
}
static int single(pml_ctx *ctx, int op, const pml_alignment *aln, const char *nw, const pml_model *model,
                  const pml_search_opts *opts, int flags, pml_result *out) {
    if (!ctx || !aln || !out) return PML_EINVAL;
    if (op != OP_SEARCH && !nw) { std::memset(out, 0, sizeof *out); return ctx->c.fail(PML_EINVAL, "newick required"); }
    if (opts && opts->nconstraints > 0) return oneshot(ctx, op, 1, aln, nw ? &nw : nullptr, model, opts, flags, out);   // caller-owned matrix: alone
    pml_request r;
    r.op = op; r.flags = flags; r.aln = aln; r.nw = nw; r.out = out;
    r.model = model ? *model : pml_model{4, 1.0, PML_PI_RAXML_3DP};
    if (opts) r.opts = *opts;
    else { std::memset(&r.opts, 0, sizeof r.opts); r.opts.optimize_alpha = 1; r.opts.nni = 1; }   // epsilon 0 = the operation's default
    r.opts.nconstraints = 0; r.opts.constraint_ntax = 0; r.opts.constraint_names = nullptr; r.opts.constraint_rows = nullptr;
    std::unique_lock<std::mutex> lk(ctx->qmu);
    ctx->queue.push_back(&r);
    while (!r.done) {
        if (ctx->leader) { ctx->qcv.wait(lk); continue; }
        ctx->leader = true;
        std::vector<pml_request *> grp, rest;
        for (pml_request *q : ctx->queue) (compatible(*ctx->queue.front(), *q) ? grp : rest).push_back(q);
        ctx->queue.swap(rest);
        ctx->coalesced_batches++; ctx->coalesced_requests += (long long)grp.size();
        lk.unlock();
        try { run_group(ctx, grp); }
        catch (...) { for (pml_request *q : grp) { q->rc = PML_ENOMEM; q->err = "host allocation failed"; } }
        lk.lock();
        for (pml_request *q : grp) q->done = true;
        ctx->leader = false;
        ctx->qcv.notify_all();
    }
    if (r.rc) { std::lock_guard<std::mutex> g(ctx->c.mu); ctx->c.last_error = r.err; }
    return r.rc;
}

extern "C" {

int pml_coalescing_stats(pml_ctx *ctx, long long *batches, long long *requests) {
    if (!ctx) return PML_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->qmu);
    if (batches) *batches = ctx->coalesced_batches;
    if (requests) *requests = ctx->coalesced_requests;
    return PML_OK;
}
int pml_score(pml_ctx *ctx, const pml_alignment *aln, const char *newick, const pml_model *model, int flags, pml_result *out) {
    return single(ctx, OP_SCORE, aln, newick, model, nullptr, flags, out);
}
int pml_optimize(pml_ctx *ctx, const pml_alignment *aln, const char *newick, const pml_model *model,
                 const pml_search_opts *opts, pml_result *out) {
    return single(ctx, OP_OPTIMIZE, aln, newick, model, opts, 0, out);
}
int pml_search(pml_ctx *ctx, const pml_alignment *aln, const char *start, const pml_model *model,
               const pml_search_opts *opts, pml_result *out) {
    return single(ctx, OP_SEARCH, aln, start, model, opts, 0, out);
}
int pml_score_batch(pml_ctx *ctx, int n, const pml_alignment *alns, const char *const *newicks, const pml_model *model,
                    int flags, pml_result *out) {
    return oneshot(ctx, OP_SCORE, n, alns, newicks, model, nullptr, flags, out);
}
int pml_optimize_batch(pml_ctx *ctx, int n, const pml_alignment *alns, const char *const *newicks,
                       const pml_model *model, const pml_search_opts *opts, pml_result *out) {
    return oneshot(ctx, OP_OPTIMIZE, n, alns, newicks, model, opts, 0, out);
}
int pml_search_batch(pml_ctx *ctx, int n, const pml_alignment *alns, const char *const *starts,
                     const pml_model *model, const pml_search_opts *opts, pml_result *out) {
    return oneshot(ctx, OP_SEARCH, n, alns, starts, model, opts, 0, out);
}

int pml_rf_distance(const char *a, const char *b, int *rf) {
    if (!a || !b || !rf) return PML_EINVAL;
    try {
        std::vector<std::string> na, nb; Tree ta, tb; std::string err;
        if (!Tree::parse_free(a, na, ta, err)) { g_err = err; return PML_EPARSE; }
        if (!Tree::parse(b, na, tb, err)) { g_err = err; return PML_EPARSE; }
        *rf = rf_distance(ta, tb);
    } catch (const std::exception &e) { g_err = e.what(); return PML_EINVAL; }
    return PML_OK;
}

int pml_support_tree(const char *main_newick, int ntrees, const char *const *support, int digits, char **out) {
    if (!main_newick || !out || ntrees < 0 || (ntrees > 0 && !support)) return PML_EINVAL;
    *out = nullptr;
    try {
        std::vector<std::string> names; Tree main; std::string err;
        if (!Tree::parse_free(main_newick, names, main, err)) { g_err = err; return PML_EPARSE; }
        std::vector<Tree> others((size_t)ntrees);
        for (int i = 0; i < ntrees; ++i)
            if (!support[i] || !Tree::parse(support[i], names, others[i], err)) { g_err = "support tree " + std::to_string(i) + ": " + err; return PML_EPARSE; }
        *out = dup_string(main.newick_labeled(names, digits, support_counts(main, others)));
    } catch (const std::exception &e) { g_err = e.what(); return PML_EINVAL; }
    return *out ? PML_OK : PML_ENOMEM;
}

int pml_sh_support_batch(pml_ctx *ctx, int n, const pml_alignment *alns, const char *const *newicks, const pml_model *model,
                         int nboot, unsigned long long seed, pml_result *out) {
    if (!ctx || !alns || !newicks || !out || n <= 0 || nboot <= 0) return PML_EINVAL;
    pml_fpguard fpg;
    for (int i = 0; i < n; ++i) { std::memset(&out[i], 0, sizeof(pml_result)); if (!newicks[i]) return ctx->c.fail(PML_EINVAL, "newick required"); }
    std::lock_guard<std::mutex> lk(ctx->c.mu);
    pml_batch *b = nullptr;
    int rc = batch_create_impl(ctx, n, alns, newicks, model, false, &b);
    if (rc) return rc;
    try {
        std::vector<std::vector<double>> sup; std::vector<std::vector<std::pair<int, int>>> edges;
        rc = b->b.sh_support(nboot, seed, sup, edges);
        std::vector<double> lnl(n);
        if (!rc) rc = b->b.evaluate(std::vector<char>(n, 1), lnl.data());
        for (int i = 0; i < n && !rc; ++i) {
            const Gene &G = b->b.genes[i]; const Tree &T = G.tree;
            std::vector<std::vector<double>> lab((size_t)T.nnodes(), std::vector<double>(3, -1.0));
            for (size_t e = 0; e < edges[i].size(); ++e) {
                const int u = edges[i][e].first, v = edges[i][e].second;
                lab[u][T.slot(u, v)] = sup[i][e]; lab[v][T.slot(v, u)] = sup[i][e];
            }
            out[i].lnl = lnl[i]; out[i].alpha = G.alpha; out[i].tree_length = T.length(); out[i].npatterns = G.aln.npat; out[i].nsites = G.aln.nsites;
            out[i].newick = dup_string(T.newick_labeled(G.aln.names, 10, lab, 3));
            if (!out[i].newick) rc = ctx->c.fail(PML_ENOMEM, "host allocation failed");
        }
    } catch (const std::bad_alloc &) { rc = ctx->c.fail(PML_ENOMEM, "host allocation failed"); }
    catch (const std::exception &e) { rc = ctx->c.fail(PML_EINVAL, e.what()); }
    b->b.destroy(); delete b;
    for (int i = 0; i < n; ++i) out[i].status = rc;
    return rc;
}
int pml_sh_support(pml_ctx *ctx, const pml_alignment *aln, const char *newick, const pml_model *model, int nboot, unsigned long long seed, pml_result *out) {
    return pml_sh_support_batch(ctx, 1, aln, &newick, model, nboot, seed, out);
}

static int gamma20_chunk(pml_ctx *ctx, int n, const pml_alignment *alns, const char *const *newicks, const pml_model &m,
                         pml_result *out, double *rescale_out) {
    pml_batch *b = nullptr;
    int rc = batch_create_impl(ctx, n, alns, newicks, &m, true, &b);
    if (rc) return rc;
    try {
        std::vector<double> lnl, al, rs;
        rc = b->b.gamma20(lnl, al, rs);
        for (int i = 0; i < n && !rc; ++i) {
            const Gene &G = b->b.genes[i];
            Tree T = G.tree;
            for (auto &l : T.len) for (double &x : l) x *= rs[i];
            out[i].lnl = lnl[i]; out[i].alpha = al[i]; out[i].tree_length = T.length(); out[i].npatterns = G.aln.npat; out[i].nsites = G.aln.nsites;
            out[i].newick = dup_string(T.newick(G.aln.names, 10));
            if (!out[i].newick) rc = ctx->c.fail(PML_ENOMEM, "host allocation failed");
            if (rescale_out) rescale_out[i] = rs[i];
        }
    } catch (const std::bad_alloc &) { rc = ctx->c.fail(PML_ENOMEM, "host allocation failed"); }
    catch (const std::exception &e) { rc = ctx->c.fail(PML_EINVAL, e.what()); }
    b->b.destroy(); delete b;
    return rc;
}

int pml_gamma20_batch(pml_ctx *ctx, int n, const pml_alignment *alns, const char *const *newicks, const pml_model *model,
                      pml_result *out, double *rescale_out) {
    if (!ctx || !alns || !newicks || !out || n <= 0) return PML_EINVAL;
    pml_fpguard fpg;
    for (int i = 0; i < n; ++i) { std::memset(&out[i], 0, sizeof(pml_result)); if (!newicks[i]) return ctx->c.fail(PML_EINVAL, "newick required"); }
    std::lock_guard<std::mutex> lk(ctx->c.mu);
    pml_drop_worker_caches(ctx);
    pml_model m = model ? *model : pml_model{4, 1.0, PML_PI_WAG_FULL};
    m.ncat = 4;
    // gene lists that do not fit in free HBM at once go through in consecutive sub-batches, like every other one-shot call
    // (oneshot()); a gene needs its score-only arena plus the 20 x mpad likelihood table and 5 x mpad scaling counts
    size_t free_b = 0, total_b = 0;
    hipSetDevice(ctx->c.device);
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)1 << 40;
    free_b += ctx->c.arena_cache_bytes;
    size_t budget = (size_t)(0.85 * (double)free_b);
    if (const char *e = std::getenv("PML_HBM_BUDGET_MB")) budget = (size_t)std::atoll(e) << 20;   // test hook
    int rc = 0, begin = 0;
    while (begin < n && !rc) {
        size_t used = 0; int end = begin;
        while (end < n) {
            const size_t mp = ((size_t)std::max(alns[end].nsites, 1) + 31) / 32 * 32;
            const size_t need = gene_bytes_bound(alns[end], true) + mp * (G20_RATES * 8 + (G20_RATES / 4) * 4) + 256;
            if (end > begin && used + need > budget) break;
            used += need; ++end;
        }
        rc = gamma20_chunk(ctx, end - begin, alns + begin, newicks + begin, m, out + begin, rescale_out ? rescale_out + begin : nullptr);
        begin = end;
    }
    for (int i = 0; i < n; ++i) out[i].status = rc;
    return rc;
}
int pml_gamma20(pml_ctx *ctx, const pml_alignment *aln, const char *newick, const pml_model *model, pml_result *out, double *rescale_out) {
    return pml_gamma20_batch(ctx, 1, aln, &newick, model, out, rescale_out);
}

int pml_refine_next(const char *newick, int cutoff, int ndone, const char *const *done, char **ingroup_out, int *nnodes_out, int **mean_out) {
    if (!newick || !ingroup_out || ndone < 0 || (ndone > 0 && !done)) return PML_EINVAL;
    *ingroup_out = nullptr; if (mean_out) *mean_out = nullptr; if (nnodes_out) *nnodes_out = 0;
    try {
        std::vector<std::string> d; for (int i = 0; i < ndone; ++i) d.push_back(done[i] ? done[i] : "");
        std::string in, err; std::vector<int> means;
        if (!refine_query(newick, cutoff, d, in, means, err)) { g_err = err; return PML_EPARSE; }
        if (!in.empty() && !(*ingroup_out = dup_string(in))) return PML_ENOMEM;
        if (nnodes_out) *nnodes_out = (int)means.size();
        if (mean_out) {
            *mean_out = (int *)std::malloc(sizeof(int) * std::max<size_t>(means.size(), 1));
            if (!*mean_out) return PML_ENOMEM;
            std::memcpy(*mean_out, means.data(), sizeof(int) * means.size());
        }
    } catch (const std::exception &e) { g_err = e.what(); return PML_EINVAL; }
    return PML_OK;
}

int pml_parsimony_batch(pml_ctx *ctx, int n, const pml_alignment *alns, const pml_parsimony_opts *opts, pml_result *out, long long *mp_length) {
    if (!ctx || !alns || !out || n <= 0) return PML_EINVAL;
    pml_fpguard fpg;
    std::lock_guard<std::mutex> lk(ctx->c.mu);
    pml_drop_worker_caches(ctx);
    for (int i = 0; i < n; ++i) std::memset(&out[i], 0, sizeof(pml_result));
    try {
        if (hipSetDevice(ctx->c.device) != hipSuccess) return ctx->c.fail(PML_EDEVICE, "hipSetDevice failed");
        // gene lists whose Fitch vectors do not fit in free HBM at once go through in consecutive sub-batches
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)1 << 40;
        size_t budget = (size_t)(0.5 * (double)(free_b + ctx->c.arena_cache_bytes));
        if (const char *e = std::getenv("PML_HBM_BUDGET_MB")) budget = (size_t)std::atoll(e) << 20;   // test hook
        const int radius = opts ? opts->spr_radius : 20;
        for (int begin = 0; begin < n;) {
            size_t used = 0; int end = begin;
            while (end < n) {
                const size_t mp = ((size_t)std::max(alns[end].nsites, 1) + 31) / 32 * 32, nt = (size_t)std::max(alns[end].ntax, 3);
                const size_t need = (4 * nt + 64 * (size_t)(radius + 2) + 8) * mp * 4;       // tips + 3 messages/node + path vectors of <= 64 groups
                if (end > begin && used + need > budget) break;
                used += need; ++end;
            }
            std::vector<Tree> trees; std::vector<EncodedAlignment> enc; std::vector<long long> len; std::vector<int> moves;
            const int rc = parsimony_batch(&ctx->c, end - begin, reinterpret_cast<const pml_alignment_view *>(alns) + begin, opts ? opts->seed : 0u,
                                           radius, trees, enc, len, moves);
            if (rc) { for (int i = 0; i < n; ++i) out[i].status = rc; return rc; }
            for (int i = begin; i < end; ++i) {
                out[i].npatterns = enc[i - begin].npat; out[i].nsites = enc[i - begin].nsites;
                out[i].newick = dup_string(trees[i - begin].newick(enc[i - begin].names, -1));
                if (!out[i].newick) return ctx->c.fail(PML_ENOMEM, "host allocation failed");
                if (mp_length) mp_length[i] = len[i - begin];
            }
            begin = end;
        }
    } catch (const std::bad_alloc &) { return ctx->c.fail(PML_ENOMEM, "host allocation failed"); }
    catch (const std::exception &e) { return ctx->c.fail(PML_EINVAL, e.what()); }
    return PML_OK;
}
int pml_parsimony(pml_ctx *ctx, const pml_alignment *aln, const pml_parsimony_opts *opts, pml_result *out, long long *mp_length) {
    return pml_parsimony_batch(ctx, 1, aln, opts, out, mp_length);
}

int pml_kernel_stats(pml_ctx *ctx, int k, long long *launches, double *ms, double *bytes) {
    if (!ctx || k < 0 || k >= K_COUNT) return PML_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->c.mu);
    if (launches) *launches = ctx->c.stats[k].launches;
    if (ms) *ms = ctx->c.stats[k].ms;
    if (bytes) *bytes = ctx->c.stats[k].bytes;
    return PML_OK;
}
int pml_newton_fallbacks(pml_ctx *ctx, long long *giveups, long long *reissued, long long *seq_launches) {
    if (!ctx) return PML_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->c.mu);
    if (giveups) *giveups = ctx->c.newton_giveups;
    if (reissued) *reissued = ctx->c.newton_reissued;
    if (seq_launches) *seq_launches = ctx->c.newton_seq_launches;
    return PML_OK;
}
int pml_kernel_flops(pml_ctx *ctx, int k, double *flops) {
    if (!ctx || !flops || k < 0 || k >= K_COUNT) return PML_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->c.mu);
    *flops = ctx->c.stats[k].flops;
    return PML_OK;
}
int pml_kernel_stats_reset(pml_ctx *ctx) {
    if (!ctx) return PML_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->c.mu);
    for (auto &s : ctx->c.stats) s = Ctx::KStat();
    return PML_OK;
}

}  // extern "C"
