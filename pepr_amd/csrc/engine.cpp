// engine.cpp -- batch engine: HBM arena, lazy directional-CLV bookkeeping, level-synchronous
// newview scheduling across genes, branch-length / alpha optimisation drivers.
//
// Scheduling model (DESIGN.md "Scheduling"): every gene keeps one CLV per DIRECTED inner edge
// ("message" v->w: likelihood of the subtree hanging off v when edge (v,w) is cut).  A request
// (lnL, branch derivatives) names the messages it needs; need() walks the tree lazily and emits
// the missing newviews in dependency levels; run() uploads all descriptors of all genes once,
// launches k_pmat, one k_nv<NEWVIEW> per level, the tail kernels, and syncs once.
#include "engine.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <thread>
#include <cstring>

namespace pml {

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return ctx->fail(-5, std::string(#expr) + ": " + hipGetErrorString(e_));          \
    } while (0)

// ------------------------------------------------------------------------------------------
// Ctx
// ------------------------------------------------------------------------------------------
int Ctx::init(int dev, bool prof) {
    Ctx *ctx = this;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return fail(-3, "no HIP device available");
    if (dev < 0 || dev >= count) return fail(-3, "device ordinal out of range");
    device = dev; profile = prof;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    // the second lane's stream exists only when the lanes experiment is on (PML_LANES=1): HIP multiplexes a process's streams
    // onto a few hardware queues, and an idle stream per context would cost the search groups (api.cpp) a queue each
    if (std::getenv("PML_LANES")) HIPCHK(hipStreamCreateWithFlags(&stream2, hipStreamNonBlocking));
    else stream2 = stream;
    return 0;
}
int Ctx::init_worker(const Ctx &parent) {
    Ctx *ctx = this;
    device = parent.device; profile = false;
    HIPCHK(hipSetDevice(device));
    // A stream of the HIGHEST priority class: HIP keeps a separate pool of hardware queues per priority, so the groups' streams
    // do not end up multiplexed onto a queue with each other's or the application's normal-priority streams (measured: two
    // groups on one hardware queue search 150 C3 gene-trees/s, on two queues 200, one undivided batch 175).
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = hi = 0; }
    if (hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, hi) != hipSuccess) HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    stream2 = stream;
    return 0;
}
int Ctx::sync(hipStream_t s) {
    Ctx *ctx = this;
    if (!ev_sync) HIPCHK(hipEventCreateWithFlags(&ev_sync, hipEventDisableTiming));
    HIPCHK(hipEventRecord(ev_sync, s));
    HIPCHK(hipEventSynchronize(ev_sync));
    return 0;
}
void Ctx::destroy() {
    hipSetDevice(device);
    if (ev_sync) { hipEventDestroy(ev_sync); ev_sync = nullptr; }
    for (auto &e : pending) { pool.push_back(e.a); pool.push_back(e.b); }
    pending.clear();
    for (auto e : pool) hipEventDestroy(e);
    pool.clear();
    for (int i = 0; i < 2; ++i) { if (d_model[i]) hipFree(d_model[i]); if (d_eigfrags[i]) hipFree(d_eigfrags[i]); d_model[i] = nullptr; d_eigfrags[i] = nullptr; }
    if (arena_cache) hipFree(arena_cache);
    arena_cache = nullptr; arena_cache_bytes = 0;
    if (stream && owns_stream) hipStreamDestroy(stream);
    if (stream2 && stream2 != stream && owns_stream) hipStreamDestroy(stream2);
    stream = stream2 = nullptr;
}
static void fill_model_dev(const Model &m, ModelDev &h) {
    std::memcpy(h.eval, m.eval, sizeof h.eval);
    std::memcpy(h.U, m.U, sizeof h.U);
    std::memcpy(h.Uinv, m.Uinv, sizeof h.Uinv);
    std::memcpy(h.pi, m.pi, sizeof h.pi);
    for (int k = 0; k < NS; ++k) for (int j = 0; j < NS; ++j) h.UinvT[j * NS + k] = m.Uinv[k * NS + j];
}
// PROTGAMMAWAGF (pi_mode 2): empirical frequencies per gene (host.cpp empirical_freqs), WAG exchangeabilities, one
// eigen-decomposition per gene on the host (20 x 20 Jacobi), models + eigen-basis fragment sets uploaded once per batch
int Batch::build_gene_models() {
    const size_t n = genes.size();
    std::vector<ModelDev> h(n);
    for (size_t g = 0; g < n; ++g) {
        double pi[20];
        empirical_freqs(genes[g].aln, pi);
        Model m; m.init_pi(pi);
        fill_model_dev(m, h[g]);
    }
    HIPCHK(hipMalloc((void **)&d_gmodel, n * sizeof(ModelDev)));
    HIPCHK(hipMalloc((void **)&d_geig, n * 2 * PFRAG * sizeof(double)));
    HIPCHK(hipMemcpyAsync(d_gmodel, h.data(), n * sizeof(ModelDev), hipMemcpyHostToDevice, ctx->stream));
    launch_eigfrags_n(d_gmodel, d_geig, (int)n, ctx->stream);
    if (int rc = ctx->sync(ctx->stream)) return rc;      // h goes out of scope
    return 0;
}
int Ctx::ensure_model(int pm) {
    Ctx *ctx = this;
    if (pm == 2) return 0;                  // PROTGAMMAWAGF: per-gene models live in the batch (Batch::build_gene_models)
    if (pm < 0 || pm > 2) return fail(-1, "bad pi_mode");
    if (model_ready[pm]) return 0;
    model[pm].init(pm);
    ModelDev h;
    fill_model_dev(model[pm], h);
    HIPCHK(hipMalloc(&d_model[pm], sizeof(ModelDev)));
    HIPCHK(hipMalloc(&d_eigfrags[pm], sizeof(double) * 2 * PFRAG));
    HIPCHK(hipMemcpy(d_model[pm], &h, sizeof h, hipMemcpyHostToDevice));
    launch_eigfrags(d_model[pm], d_eigfrags[pm], stream);
    if (int rc = sync(stream)) return rc;
    model_ready[pm] = true;
    return 0;
}
hipEvent_t Ctx::get_event() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e; hipEventCreate(&e); return e;
}
void Ctx::tic(int kind, double bytes, double flops) {
    stats[kind].launches++; stats[kind].bytes += bytes; stats[kind].flops += flops;
    if (!profile) return;
    Ev ev{kind, get_event(), get_event()};
    hipEventRecord(ev.a, tic_stream ? tic_stream : stream);
    pending.push_back(ev);
}
void Ctx::toc() {
    if (!profile) return;
    hipEventRecord(pending.back().b, tic_stream ? tic_stream : stream);
}
void Ctx::resolve_events() {
    for (auto &e : pending) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) stats[e.kind].ms += ms;
        pool.push_back(e.a); pool.push_back(e.b);
    }
    pending.clear();
}

// ------------------------------------------------------------------------------------------
// Batch: creation / layout
// ------------------------------------------------------------------------------------------
static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// PML_DET_LOG=<file>: every value the host consumes from the device (evaluations, Newton results) is recorded in memory
// as a binary record and written out at process exit -- a determinism diagnostic (tools/dbg_detlog_diff.py) whose cost per
// record is a few stores, so it does not perturb the timing it is meant to observe; off unless the variable is set
struct DetRec { int batch, ntax, npat, nsites, kind, a, b, pad; double x, y, z; };
static std::vector<DetRec> *g_det = nullptr;
static std::mutex g_det_mu;
static bool det_on() {
    static const bool on = [] {
        const char *p = std::getenv("PML_DET_LOG");
        if (!p) return false;
        g_det = new std::vector<DetRec>(); g_det->reserve(1 << 22);
        std::atexit([] {
            const char *q = std::getenv("PML_DET_LOG"); FILE *f = q ? std::fopen(q, "w") : nullptr;
            if (!f) return;
            for (const DetRec &r : *g_det) std::fprintf(f, "B%d g%d_%d_%d %c %d %d %a %a %a\n", r.batch, r.ntax, r.npat, r.nsites, (char)r.kind, r.a, r.b, r.x, r.y, r.z);
            std::fclose(f);
        });
        return true;
    }();
    return on;
}
void det_record(int batch, const Gene &G, char kind, int a, int b, double x, double y, double z) {
    if (!det_on()) return;
    std::lock_guard<std::mutex> lk(g_det_mu);
    g_det->push_back(DetRec{batch, G.aln.ntax, G.aln.npat, G.aln.nsites, kind, a, b, 0, x, y, z});
}
static std::atomic<int> g_batch_id{0};
static double now_ms();

int Batch::create(Ctx *c, int n, const pml_alignment_view *alns, const char *const *newicks, int pm, int nc,
                  double alpha, bool score_only) {
    ctx = c; pi_mode = pm; ncat = nc; score_only_batch = score_only;
    det_id = ++g_batch_id;
    virtual_cherries = std::getenv("PML_NO_CHERRY") == nullptr;
    virtual_pitch = virtual_cherries && std::getenv("PML_NO_PITCH") == nullptr;
    if (n <= 0) return ctx->fail(-1, "empty batch");
    if (nc != 1 && nc != 4) return ctx->fail(-1, "ncat must be 1 or 4");
    if (int rc = ctx->ensure_model(pm)) return rc;
    HIPCHK(hipSetDevice(ctx->device));
    genes.resize(n);
    const double t_create0 = now_ms();
    {   // encode / parse / NJ are independent per gene: host threads (plain std::thread, no GPU work)
        const int nthreads = std::max(1, std::min({n, 16, (int)std::thread::hardware_concurrency()}));
        std::vector<std::string> errs(n);
        std::atomic<int> next{0};
        auto work = [&]() {
            for (int g = next++; g < n; g = next++) {
                Gene &G = genes[g];
                try {
                    if (!G.aln.encode(alns[g].ntax, alns[g].nsites, alns[g].names, alns[g].rows, errs[g])) continue;
                    if (newicks && newicks[g]) { if (!Tree::parse(newicks[g], G.aln.names, G.tree, errs[g])) continue; }
                    else G.tree = nj_tree(G.aln);
                } catch (const std::exception &e) { errs[g] = e.what(); }
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < nthreads; ++t) pool.emplace_back(work);
        work();
        for (auto &t : pool) t.join();
        for (int g = 0; g < n; ++g) if (!errs[g].empty()) return ctx->fail(-2, "gene " + std::to_string(g) + ": " + errs[g]);
    }
    if (std::getenv("PML_TRACE")) fprintf(stderr, "[pml] create: encode + start trees of %d genes %.1f ms\n", n, now_ms() - t_create0);
    if (int rc = layout(alpha, score_only)) return rc;
    return pm == 2 ? build_gene_models() : 0;
}

// device arena + per-gene pointers from (ntax, mpad) alone; host-encoded codes/weights are uploaded when present
// (replicates built by create_replicates() have none: k_gather fills them)
int Batch::layout(double alpha, bool score_only) {
    const int n = (int)genes.size();
    size_t total = 0;
    std::vector<size_t> off(n);
    for (int g = 0; g < n; ++g) {
        Gene &G = genes[g];
        const int nt = G.aln.ntax, mp = G.aln.mpad, ndir = 3 * (nt - 2);
        G.slot_cap = score_only ? (nt - 2) : ndir;
        G.slot_of.assign(ndir, -1); G.valid.assign(ndir, 0); G.pend_level.assign(ndir, -1);
        G.mark_all();
        off[g] = total;
        total += align_up((size_t)nt * mp, 256);                       // codes
        total += align_up((size_t)mp * 8, 256);                        // weight
        total += (size_t)(G.slot_cap + NSCRATCH) * clv_doubles(mp) * 8;  // clv (+ scratch), tiled: whole 128-pattern tiles
        total += align_up((size_t)(G.slot_cap + NSCRATCH) * mp * 4, 256);   // scalers
        total += (size_t)MAXTAIL * clv_doubles(mp) * 8;                  // sumtables
        total += (size_t)MAXTAIL * align_up((size_t)mp * 4, 256);      // sumtable scalers
        total += (size_t)MAXTAIL * align_up((size_t)mp * 8, 256);      // per-pattern lnL
    }
    arena_bytes = total;
    const double t_alloc0 = now_ms();
    if (ctx->arena_cache && ctx->arena_cache_bytes >= total) {        // reuse: no driver allocation, no zero-fill
        arena = ctx->arena_cache; arena_bytes = ctx->arena_cache_bytes;
        ctx->arena_cache = nullptr; ctx->arena_cache_bytes = 0;
    } else {
        if (ctx->arena_cache) { hipFree(ctx->arena_cache); ctx->arena_cache = nullptr; ctx->arena_cache_bytes = 0; }
        if (hipMalloc((void **)&arena, total) != hipSuccess) {
            arena = nullptr;
            return ctx->fail(-4, "device arena of " + std::to_string(total >> 20) + " MiB does not fit");
        }
    }
    // debugging aid: a reused arena holds stale data; poisoning it (all-ones = NaN doubles, -1 counts) makes any read of
    // a location this batch has not written show up in the results
    if (std::getenv("PML_POISON_ARENA")) HIPCHK(hipMemsetAsync(arena, 0xFF, total, ctx->stream));
    for (int g = 0; g < n; ++g) {
        Gene &G = genes[g];
        const int nt = G.aln.ntax, mp = G.aln.mpad;
        char *p = arena + off[g];
        G.d_codes = (uint8_t *)p; p += align_up((size_t)nt * mp, 256);
        G.d_weight = (double *)p; p += align_up((size_t)mp * 8, 256);
        G.d_clv = (double *)p; p += (size_t)(G.slot_cap + NSCRATCH) * clv_doubles(mp) * 8;
        G.d_scl = (int *)p; p += align_up((size_t)(G.slot_cap + NSCRATCH) * mp * 4, 256);
        for (int k = 0; k < MAXTAIL; ++k) { G.d_sumtab[k] = (double *)p; p += clv_doubles(mp) * 8; }
        for (int k = 0; k < MAXTAIL; ++k) { G.d_sumscl[k] = (int *)p; p += align_up((size_t)mp * 4, 256); }
        for (int k = 0; k < MAXTAIL; ++k) { G.d_patlnl[k] = (double *)p; p += align_up((size_t)mp * 8, 256); }
        if (!G.aln.codes.empty()) {
            HIPCHK(hipMemcpyAsync(G.d_codes, G.aln.codes.data(), (size_t)nt * mp, hipMemcpyHostToDevice, ctx->stream));
            HIPCHK(hipMemcpyAsync(G.d_weight, G.aln.weight.data(), (size_t)mp * 8, hipMemcpyHostToDevice, ctx->stream));
        }
        set_alpha(g, alpha);
    }
    // results (8 doubles per gene) are written by the kernels straight into mapped pinned host
    // memory: no device-to-host copy node per step
    // Results (8 doubles per gene and tail slot) live in DEVICE memory and reach the host by an explicit copy on the engine's
    // stream before every synchronisation (fetch_results).  Rounds 1-2 let the kernels store them straight into mapped host
    // memory; the explicit copy keeps PCIe writes out of the kernels and makes the hand-over an ordinary stream operation
    // (it was one of the suspects of the reproducibility hunt of DESIGN.md 9 r02-g and changed nothing there).
    scalars_doubles = (size_t)8 * MAXTAIL * n;
    HIPCHK(hipMalloc((void **)&d_scalars, sizeof(double) * scalars_doubles));
    HIPCHK(hipHostMalloc((void **)&h_scalars, sizeof(double) * scalars_doubles, hipHostMallocDefault));
    std::memset(h_scalars, 0, sizeof(double) * scalars_doubles);
    { if (int rc_ = ctx->sync(ctx->stream)) return rc_; }
    if (std::getenv("PML_TRACE")) fprintf(stderr, "[pml] layout: arena %.1f GiB allocated + uploaded in %.1f ms\n", (double)total / (1 << 30), now_ms() - t_alloc0);
    return 0;
}

void Batch::destroy() {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (arena) {
        if (ctx->arena_cache_bytes < arena_bytes) {                  // keep the larger one for the next batch
            if (ctx->arena_cache) hipFree(ctx->arena_cache);
            ctx->arena_cache = arena; ctx->arena_cache_bytes = arena_bytes;
        } else hipFree(arena);
    }
    if (h_stage) hipHostFree(h_stage);
    if (d_stage) hipFree(d_stage);
    if (d_frags) hipFree(d_frags);
    if (d_nsync) hipFree(d_nsync);
    if (d_nctl) {
        if (std::getenv("PML_TRACE")) {
            NewtonCtl h[2];
            if (hipMemcpy(h, d_nctl, sizeof h, hipMemcpyDeviceToHost) == hipSuccess && h[0].n_requests)
                fprintf(stderr, "[pml] branch Newton: %llu requests, %.2f evaluations each\n", h[0].n_requests, (double)h[0].n_evals / (double)h[0].n_requests);
        }
        hipFree(d_nctl); d_nctl = nullptr;
    }
    if (d_gmodel) { hipFree(d_gmodel); d_gmodel = nullptr; }
    if (d_geig) { hipFree(d_geig); d_geig = nullptr; }
    if (ev_stagger) { hipEventDestroy(ev_stagger); ev_stagger = nullptr; }
    if (d_nsync2) hipFree(d_nsync2);
    if (d_frags2) hipFree(d_frags2);
    d_nsync = d_nsync2 = nullptr; d_frags2 = nullptr; nsync_cap = nsync_cap2 = 0; frag_cap2 = 0;
    if (plan.h) hipHostFree(plan.h);
    if (plan.d) hipFree(plan.d);
    plan = Plan();
    if (h_scalars) hipHostFree(h_scalars);
    if (d_scalars) hipFree(d_scalars);
    if (h_chain) hipHostFree(h_chain);
    if (d_chain) hipFree(d_chain);
    if (d_lenpool) hipFree(d_lenpool);
    if (d_tailpool) { hipFree(d_tailpool); d_tailpool = nullptr; tailpool_cap = 0; }
    if (d_site2pat) { hipFree(d_site2pat); d_site2pat = nullptr; }
    h_chain = d_chain = nullptr; d_lenpool = nullptr; chain_cap = 0;
    arena = nullptr; h_stage = d_stage = nullptr; d_frags = nullptr; d_scalars = h_scalars = nullptr;
}

// ------------------------------------------------------------------------------------------
// GeneStore / create_replicates: jackknife concatenation on the device
// ------------------------------------------------------------------------------------------
int GeneStore::create(Ctx *c, int n, const pml_alignment_view *alns) {
    ctx = c;
    HIPCHK(hipSetDevice(ctx->device));
    items.resize(n);
    std::vector<std::string> errs(n);
    std::atomic<int> next{0};
    auto work = [&]() {
        for (int g = next++; g < n; g = next++) {
            try { if (items[g].aln.encode(alns[g].ntax, alns[g].nsites, alns[g].names, alns[g].rows, errs[g])) pair_counts(items[g].aln, items[g].cmp, items[g].diff); }
            catch (const std::exception &e) { errs[g] = e.what(); }
        }
    };
    const int nthreads = std::max(1, std::min({n, 16, (int)std::thread::hardware_concurrency()}));
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; ++t) pool.emplace_back(work);
    work();
    for (auto &t : pool) t.join();
    for (int g = 0; g < n; ++g) if (!errs[g].empty()) return ctx->fail(-2, "gene " + std::to_string(g) + ": " + errs[g]);
    size_t total = 0;
    for (auto &it : items) total += align_up((size_t)it.aln.ntax * it.aln.mpad, 256) + align_up((size_t)it.aln.mpad * 8, 256);
    if (hipMalloc((void **)&arena, total) != hipSuccess) { arena = nullptr; return ctx->fail(-4, "gene store does not fit on the device"); }
    char *p = arena;
    for (auto &it : items) {
        it.d_codes = (uint8_t *)p; p += align_up((size_t)it.aln.ntax * it.aln.mpad, 256);
        it.d_w = (double *)p; p += align_up((size_t)it.aln.mpad * 8, 256);
        HIPCHK(hipMemcpyAsync(it.d_codes, it.aln.codes.data(), (size_t)it.aln.ntax * it.aln.mpad, hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(hipMemcpyAsync(it.d_w, it.aln.weight.data(), (size_t)it.aln.mpad * 8, hipMemcpyHostToDevice, ctx->stream));
    }
    { if (int rc_ = ctx->sync(ctx->stream)) return rc_; }
    return 0;
}
void GeneStore::destroy() { if (arena) { hipSetDevice(ctx->device); hipFree(arena); arena = nullptr; } items.clear(); }

int Batch::create_replicates(Ctx *c, const GeneStore &store, const std::vector<std::vector<int>> &sel, int pm, int nc, double alpha) {
    ctx = c; pi_mode = pm; ncat = nc; score_only_batch = false;
    virtual_cherries = std::getenv("PML_NO_CHERRY") == nullptr;
    virtual_pitch = virtual_cherries && std::getenv("PML_NO_PITCH") == nullptr;
    const int n = (int)sel.size();
    if (n <= 0) return ctx->fail(-1, "empty batch");
    if (pm == 2) return ctx->fail(-1, "PROTGAMMAWAGF (empirical frequencies) is built for score / optimize / search calls, not for device-gathered replicates");
    if (int rc = ctx->ensure_model(pm)) return rc;
    HIPCHK(hipSetDevice(ctx->device));
    genes.resize(n);
    struct SegH { int rep, gene, off; size_t rowmap_off; };
    std::vector<SegH> segs; std::vector<int> rowmaps;
    int max_npat = 0;
    for (int r = 0; r < n; ++r) {
        Gene &G = genes[r]; EncodedAlignment &A = G.aln;
        std::vector<std::string> names;
        for (int g : sel[r]) {
            if (g < 0 || g >= (int)store.items.size()) return ctx->fail(-1, "gene index out of range");
            names.insert(names.end(), store.items[g].aln.names.begin(), store.items[g].aln.names.end());
        }
        std::sort(names.begin(), names.end()); names.erase(std::unique(names.begin(), names.end()), names.end());   // MSAConcatenator.java:78-189: sorted union
        const int nt = (int)names.size();
        if (nt < 3) return ctx->fail(-2, "replicate " + std::to_string(r) + ": fewer than 3 taxa");
        A.ntax = nt; A.names = names; A.nsites = 0; A.npat = 0;
        std::vector<int64_t> cmp((size_t)nt * nt, 0), diff((size_t)nt * nt, 0);
        for (int g : sel[r]) {
            const GeneStore::Item &it = store.items[g];
            std::vector<int> local(it.aln.ntax);                 // gene row -> replicate row
            SegH sh{r, g, A.npat, rowmaps.size()};
            rowmaps.resize(rowmaps.size() + nt, -1);
            for (int i = 0; i < it.aln.ntax; ++i) {
                local[i] = (int)(std::lower_bound(names.begin(), names.end(), it.aln.names[i]) - names.begin());
                rowmaps[sh.rowmap_off + local[i]] = i;
            }
            for (int i = 0; i < it.aln.ntax; ++i) for (int j = 0; j < it.aln.ntax; ++j) {
                cmp[(size_t)local[i] * nt + local[j]] += it.cmp[(size_t)i * it.aln.ntax + j]; diff[(size_t)local[i] * nt + local[j]] += it.diff[(size_t)i * it.aln.ntax + j];
            }
            segs.push_back(sh);
            A.npat += it.aln.npat; A.nsites += it.aln.nsites; max_npat = std::max(max_npat, it.aln.npat);
        }
        A.mpad = (A.npat + 31) / 32 * 32;
        G.tree = nj_from_counts(nt, cmp, diff);
    }
    if (int rc = layout(alpha, false)) return rc;
    // padding patterns: gap code, weight 0; then one gather launch fills every replicate
    for (auto &G : genes) {
        HIPCHK(hipMemsetAsync(G.d_codes, NCODES - 1, (size_t)G.aln.ntax * G.aln.mpad, ctx->stream));
        HIPCHK(hipMemsetAsync(G.d_weight, 0, (size_t)G.aln.mpad * 8, ctx->stream));
    }
    std::vector<GatherSeg> hs(segs.size());
    void *d_buf = nullptr;
    const size_t seg_bytes = align_up(hs.size() * sizeof(GatherSeg), 256), map_bytes = rowmaps.size() * sizeof(int);
    HIPCHK(hipMalloc(&d_buf, seg_bytes + map_bytes));
    const int *d_maps = (const int *)((char *)d_buf + seg_bytes);
    for (size_t i = 0; i < segs.size(); ++i) {
        const GeneStore::Item &it = store.items[segs[i].gene]; Gene &G = genes[segs[i].rep];
        hs[i] = GatherSeg{it.d_codes, it.d_w, G.d_codes, G.d_weight, d_maps + segs[i].rowmap_off, it.aln.mpad, it.aln.npat, G.aln.mpad, segs[i].off, G.aln.ntax, 0};
    }
    hipError_t e1 = hipMemcpyAsync(d_buf, hs.data(), hs.size() * sizeof(GatherSeg), hipMemcpyHostToDevice, ctx->stream);
    hipError_t e2 = hipMemcpyAsync((char *)d_buf + seg_bytes, rowmaps.data(), map_bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e1 == hipSuccess && e2 == hipSuccess) launch_gather((const GatherSeg *)d_buf, (int)hs.size(), max_npat, ctx->stream);
    hipError_t e3 = ctx->sync(ctx->stream) ? hipErrorUnknown : hipSuccess;
    hipFree(d_buf);
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return ctx->fail(-4, "replicate gather failed");
    return 0;
}

// device -> host copy of the result buffers, enqueued behind the kernels that write them; the caller synchronises
int Batch::fetch_results(bool pooled) {
    HIPCHK(hipMemcpyAsync(h_scalars, d_scalars, sizeof(double) * scalars_doubles, hipMemcpyDeviceToHost, ctx->stream));
    if (pooled && results_used > 0 && d_chain)
        HIPCHK(hipMemcpyAsync(h_chain, d_chain, sizeof(double) * 4 * results_used, hipMemcpyDeviceToHost, ctx->stream));
    return 0;
}
int Batch::chain_sync() {
    if (int rc = flush_deferred()) return rc;
    if (lanes_active) { if (int rc_ = ctx->sync(ctx->stream2)) return rc_; }      // lane 1's results must be complete before the copy
    if (int rc = fetch_results(true)) return rc;
    { if (int rc_ = ctx->sync(ctx->stream)) return rc_; }
    { if (int rc_ = ctx->sync(ctx->stream2)) return rc_; }
    HIPCHK(hipGetLastError());
    ctx->resolve_events();
    chain_off = 0;
    return 0;
}
int Batch::ensure_results(size_t nresults) {
    if (nresults > chain_cap) {
        if (h_chain) hipHostFree(h_chain);
        if (d_chain) hipFree(d_chain);
        h_chain = d_chain = nullptr; chain_cap = 0;
        const size_t cap = nresults * 3 / 2 + 64;
        HIPCHK(hipMalloc((void **)&d_chain, cap * 4 * sizeof(double)));
        HIPCHK(hipHostMalloc((void **)&h_chain, cap * 4 * sizeof(double), hipHostMallocDefault));
        chain_cap = cap;
    }
    results_used = nresults;
    return 0;
}
int Batch::ensure_tailpool(size_t bytes) {
    if (bytes <= tailpool_cap) return 0;
    if (d_tailpool) hipFree(d_tailpool);
    d_tailpool = nullptr; tailpool_cap = 0;
    if (hipMalloc((void **)&d_tailpool, bytes) != hipSuccess) { d_tailpool = nullptr; return ctx->fail(-4, "sumtable pool of " + std::to_string(bytes >> 20) + " MiB does not fit"); }
    tailpool_cap = bytes;
    return 0;
}
int Batch::chain_begin(size_t nresults) {
    if (int rc = ensure_results(nresults)) return rc;
    if (!d_lenpool) {
        size_t tot = 0; for (auto &G : genes) tot += (size_t)G.tree.nnodes() * 3;
        HIPCHK(hipMalloc((void **)&d_lenpool, tot * sizeof(double)));
        tot = 0; for (auto &G : genes) { G.d_len = d_lenpool + tot; tot += (size_t)G.tree.nnodes() * 3; }
    }
    for (auto &G : genes) G.len_pending.assign((size_t)G.tree.nnodes() * 3, 0);
    // descriptors of the whole pass stay in the staging ring until the final sync: ~4 KB per (gene, step) is what
    // run() reserves (it sizes for the worst case of 10 matrix requests per operation)
    if (int rc = ensure_stage(std::min<size_t>(nresults * 4096 + (1 << 20), (size_t)256 << 20))) return rc;
    chain = true; chain_off = 0; flush_quota = 1;
    return 0;
}
int Batch::ensure_stage(size_t bytes) {
    if (bytes <= h_cap) return 0;
    if (chain) { if (int rc = chain_sync()) return rc; }
    const size_t cap = std::max(bytes * 3 / 2, (size_t)1 << 20);
    if (h_stage) hipHostFree(h_stage);
    if (d_stage) hipFree(d_stage);
    h_stage = d_stage = nullptr; h_cap = d_cap = 0;
    HIPCHK(hipHostMalloc(&h_stage, cap));
    HIPCHK(hipMalloc(&d_stage, cap));
    h_cap = d_cap = cap;
    return 0;
}
int Batch::ensure_frags(size_t sets) {
    double *&fr = lane ? d_frags2 : d_frags; size_t &fc = lane ? frag_cap2 : frag_cap;
    if (sets <= fc) return 0;
    if (chain) { if (int rc = chain_sync()) return rc; }
    const size_t cap = std::max(sets * 5 / 4, (size_t)256);
    if (fr) hipFree(fr);
    fr = nullptr; fc = 0;
    if (!lane) plan.valid = false;      // cached descriptors point into d_frags
    HIPCHK(hipMalloc((void **)&fr, cap * FRAG_STRIDE * sizeof(double)));
    fc = cap;
    return 0;
}

void Batch::set_alpha(int g, double a) {
    Gene &G = genes[g];
    a = std::min(std::max(a, ALPHA_MIN), ALPHA_MAX);
    G.alpha = a;
    if (ncat == 1) { for (double &r : G.rates) r = 1.0; }
    else gamma_rates(a, NCAT, G.rates);
    ++G.rates_epoch;
    invalidate_all(g);
}
void Batch::invalidate_all(int g) {
    Gene &G = genes[g];
    std::fill(G.valid.begin(), G.valid.end(), 0);
    if (G.slot_cap < (int)G.slot_of.size()) { std::fill(G.slot_of.begin(), G.slot_of.end(), -1); G.next_slot = 0; }
}
static void invalidate_from(Gene &G, int v, int from) {
    // iterative DFS: every message leaving v away from `from`, and onwards
    std::vector<std::pair<int, int>> st{{v, from}};
    const int nt = G.aln.ntax;
    while (!st.empty()) {
        auto [x, f] = st.back(); st.pop_back();
        if (x < nt) continue;
        for (int k = 0; k < 3; ++k) {
            const int w = G.tree.nbr[x][k];
            if (w == f || w < 0) continue;
            G.valid[(x - nt) * 3 + k] = 0;
            st.push_back({w, x});
        }
    }
}
void Batch::branch_changed(int g, int a, int b) {
    invalidate_from(genes[g], a, b); invalidate_from(genes[g], b, a);
}
int Batch::slot_for(Gene &G, int idx) {
    int s = G.slot_of[idx];
    if (s < 0) { if (G.next_slot >= G.slot_cap) return -1; s = G.slot_of[idx] = G.next_slot++; }
    return s;
}

// ------------------------------------------------------------------------------------------
// lazy collection of the newviews a message depends on
// ------------------------------------------------------------------------------------------
bool Batch::is_cherry(int g, int node, int toward) const {
    const Gene &G = genes[g];
    const int nt = G.aln.ntax;
    if (!virtual_cherries || node < nt) return false;
    for (int k = 0; k < 3; ++k) { const int w = G.tree.nbr[node][k]; if (w != toward && w >= nt) return false; }
    return true;
}
int Batch::virt_kind(int g, int node, int toward) const {
    if (is_cherry(g, node, toward)) return 1;
    if (!virtual_pitch) return 0;
    const Gene &G = genes[g];
    const int nt = G.aln.ntax;
    if (node < nt) return 0;
    int ntip = 0, inner = -1;
    for (int k = 0; k < 3; ++k) { const int w = G.tree.nbr[node][k]; if (w == toward) continue; if (w < nt) ++ntip; else inner = w; }
    return (ntip == 1 && inner >= 0 && is_cherry(g, inner, node)) ? 2 : 0;
}
Side Batch::msg(int g, int node, int toward) const {
    const Gene &G = genes[g];
    if (node < G.aln.ntax) return {SIDE_TIP, node};
    const int idx = (node - G.aln.ntax) * 3 + G.tree.slot(node, toward);
    const int vk = virt_kind(g, node, toward);
    return {vk == 1 ? SIDE_CHERRY : (vk == 2 ? SIDE_PITCH : SIDE_MSG), idx};
}

int Batch::need(int g, int v, int to, std::vector<PendingOp> &ops) {
    Gene &G = genes[g];
    const int nt = G.aln.ntax;
    if (v < nt || virt_kind(g, v, to)) return 0;
    {                                                   // most calls ask for a message that is valid or already pending
        const int idx0 = (v - nt) * 3 + G.tree.slot(v, to);
        if (G.valid[idx0]) return 0;
        if (G.pend_level[idx0] >= 0) return G.pend_level[idx0];
    }
    // explicit stack (trees can be caterpillars of depth ~ntax)
    struct Frame { int v, to, k, stage, lv[2]; };
    std::vector<Frame> st;
    st.reserve(64);
    st.push_back({v, to, G.tree.slot(v, to), 0, {0, 0}});
    int ret = 0;
    while (!st.empty()) {
        Frame &f = st.back();
        const int idx = (f.v - nt) * 3 + f.k;
        if (f.stage == 0) {
            if (G.valid[idx]) { ret = 0; st.pop_back(); continue; }
            if (G.pend_level[idx] >= 0) { ret = G.pend_level[idx]; st.pop_back(); continue; }
        }
        int ch[2], ci = 0;
        for (int q = 0; q < 3; ++q) if (q != f.k) ch[ci++] = G.tree.nbr[f.v][q];
        if (f.stage >= 1) f.lv[f.stage - 1] = ret;
        if (f.stage < 2) {
            const int c = ch[f.stage];
            f.stage++;
            if (c < nt || virt_kind(g, c, f.v)) { ret = 0; continue; }   // tip / virtual (cherry, pitchfork) child: nothing to compute
            const int fv = f.v;
            st.push_back({c, fv, G.tree.slot(c, fv), 0, {0, 0}});
            continue;
        }
        const int lvl = std::max(f.lv[0], f.lv[1]) + 1;
        PendingOp op; op.gene = g; op.out_kind = SIDE_MSG; op.out_id = idx; op.level = lvl; ci = 0;
        for (int q = 0; q < 3; ++q) if (q != f.k) { op.child[ci] = msg(g, G.tree.nbr[f.v][q], f.v); op.t[ci] = G.tree.len[f.v][q]; ci++; }
        ops.push_back(op);
        G.pend_level[idx] = lvl;
        ret = lvl;
        st.pop_back();
    }
    return ret;
}

// ------------------------------------------------------------------------------------------
// run: one upload, pmat, newview levels, tails, one sync
// ------------------------------------------------------------------------------------------
static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void Batch::newton_gave_up() {
    ++ctx->newton_giveups;
    if (std::getenv("PML_TRACE") && d_nctl) {
        NewtonCtl h[2];
        if (hipMemcpy(h, d_nctl, sizeof h, hipMemcpyDeviceToHost) == hipSuccess)
            fprintf(stderr, "[pml] k_newton exchange gave up: slice %d of %d, evaluation %d, arrived mask %08x%08x, tickets taken in its partition %d, (left * 1024 + partition) %d; tickets %d %d done %d %d oticket %d %d %d %d %d %d %d %d odone %d\n",
                    h[0].dbg[1], h[0].dbg[2], h[0].dbg[3], (unsigned)h[0].dbg[5], (unsigned)h[0].dbg[4], (unsigned)h[0].dbg[6], h[0].dbg[7], h[0].ticket[0], h[0].ticket[1], h[0].done[0], h[0].done[1],
                    h[0].oticket[0], h[0].oticket[1], h[0].oticket[2], h[0].oticket[3], h[0].oticket[4], h[0].oticket[5], h[0].oticket[6], h[0].oticket[7], h[0].odone);
        hipMemset(&d_nctl[0].dbg[0], 0, sizeof(int));
    }
    safe_left = safe_hold; safe_hold = std::min(safe_hold * 2, 1024);
}
// the sticky abort word of both lanes' control blocks back to 0 (ordered on the batch's streams, then waited for)
int Batch::clear_abort() {
    if (!d_nctl) return 0;
    for (int l = 0; l < 2; ++l) HIPCHK(hipMemsetAsync(&d_nctl[l].abort, 0, sizeof(int), l ? ctx->stream2 : ctx->stream));
    { if (int rc_ = ctx->sync(ctx->stream2)) return rc_; }
    { if (int rc_ = ctx->sync(ctx->stream)) return rc_; }
    return 0;
}

// one upload for all deferred steps (their descriptors are consecutive in the staging ring), then their launches in order
int Batch::flush_deferred() {
    if (deferred.empty()) return 0;
    // whatever happens below, the queue is empty afterwards and the event stream is reset: an error must not leave stale
    // descriptors to be uploaded and launched again by the next chain_sync() / run()
    struct Guard { Batch *b; ~Guard() { b->deferred.clear(); b->ctx->tic_stream = nullptr; } } guard{this};
    const size_t lo = deferred.front().base, hi = deferred.back().base + deferred.back().bytes;
    const hipStream_t cs = deferred.front().lane ? ctx->stream2 : ctx->stream;
    HIPCHK(hipMemcpyAsync((char *)d_stage + lo, (char *)h_stage + lo, hi - lo, hipMemcpyHostToDevice, cs));
    const ModelDev *md = pi_mode < 2 ? ctx->d_model[pi_mode] : nullptr;      // per-gene models travel in the requests
    static const bool serialize = std::getenv("PML_SERIALIZE") != nullptr;      // diagnostic: a host sync after every launch
#define PML_SER() do { if (serialize) hipStreamSynchronize(st); } while (0)
    if (serialize) hipStreamSynchronize(cs);
    for (const Deferred &L : deferred) {
        const hipStream_t st = L.lane ? ctx->stream2 : ctx->stream;
        double *const frags_buf = L.lane ? d_frags2 : d_frags;
        char *ds = (char *)d_stage + L.base;
        ctx->tic_stream = st;
        if (L.nreq) {
            ctx->tic(K_PMAT, (double)L.nreq * PFRAG * 8);
            launch_pmat(md, (const PmatReq *)(ds + L.o_req), frags_buf, (int)L.nreq, st, d_gmodel != nullptr);
            ctx->toc(); PML_SER();
        }
        if (L.nruns) {
            ctx->tic(K_NEWVIEW, L.algo_bytes, L.algo_flops);
            launch_oplist((const NvOp *)(ds + L.o_ops), (const GeneRun *)(ds + L.o_runs), (int)L.nruns, L.max_mpad, L.any_pitch, L.any_chain, st, L.fused ? d_nctl + L.lane : nullptr);
            ctx->toc(); PML_SER();
        }
        if (L.stagger) hipEventRecord(ev_stagger, st);
        if (L.neval) {
            ctx->tic(K_REDUCE, 0);
            launch_reduce((const ReduceReq *)(ds + L.o_red), (int)L.neval, st);
            ctx->toc(); PML_SER();
        }
        if (L.nt_reg + L.nt_stream > 0) {
            ctx->tic(K_NEWTON, L.newton_bytes);
            if (L.seq) { launch_newton_seq(md, (const NewtonReq *)(ds + L.o_newt), (const int *)(ds + L.o_tick), L.nt_reg, L.nt_stream, d_nctl + L.lane, st); ++ctx->newton_seq_launches; }
            else launch_newton(md, (const NewtonReq *)(ds + L.o_newt), (const int *)(ds + L.o_tick), L.nt_reg, L.nt_stream, d_nctl + L.lane, st);
            ctx->toc(); PML_SER();
        }
        ctx->tic_stream = nullptr;
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return ctx->fail(-5, std::string("kernel launch: ") + hipGetErrorString(e));
    }
#undef PML_SER
    static const size_t flush_max = std::getenv("PML_FLUSH_MAX") ? (size_t)std::atoi(std::getenv("PML_FLUSH_MAX")) : 8;      // A-B arm
    flush_quota = std::min<size_t>(flush_quota * 2, flush_max);
    return 0;
}

int Batch::run(std::vector<PendingOp> &ops, const std::vector<Tail> &tails) {
    const double t_begin = now_ms();
    HIPCHK(hipSetDevice(ctx->device));
    // group by gene, keeping each gene's dependency order (children are emitted before parents)
    std::stable_sort(ops.begin(), ops.end(), [](const PendingOp &a, const PendingOp &b) { return a.gene != b.gene ? a.gene < b.gene : a.part < b.part; });
    const size_t nops = ops.size(), ntail = tails.size();
    size_t neval = 0, nnewton = 0;
    for (auto &t : tails) { if (t.mode == MODE_EVALUATE) neval++; else if (t.mode != MODE_EVALUATE_CAT) nnewton++; }
    // <= 5 requests per side (pitchfork: 3 tables + 2 fragment sets) -- but requests across the same tree branch are
    // shared within the launch (add_req), so a gene never needs more than 5 per taxon plus those of lengths that
    // belong to no branch of the tree (SPR path / insertion operations)
    size_t nreq_max = 10 * nops + 9 * ntail;
    {
        std::vector<char> seen(genes.size(), 0);
        size_t keyed = 0, loose = 0;
        for (auto &o : ops) { if (!seen[o.gene]) { seen[o.gene] = 1; keyed += 5 * (size_t)genes[o.gene].aln.ntax; }
                              if (o.out_kind != SIDE_MSG) loose += (o.bv[0] < 0) + (o.bv[1] < 0); }      // one fragment set per child whose length is no tree branch
        for (auto &t : tails) { if (!seen[t.gene]) { seen[t.gene] = 1; keyed += 5 * (size_t)genes[t.gene].aln.ntax; }
                                if (t.mode >= MODE_EVALUATE && t.bv < 0) loose += 1; }
        nreq_max = std::min(nreq_max, keyed + loose);
    }
    bool any_pitch = false, any_chain = false;
    if (int rc = ensure_frags(std::max(nreq_max, (size_t)1))) return rc;
    double *&nsync_buf = lane ? d_nsync2 : d_nsync; size_t &nsync_c = lane ? nsync_cap2 : nsync_cap;
    if (nnewton > nsync_c) {
        if (chain) { if (int rc = chain_sync()) return rc; }
        if (nsync_buf) hipFree(nsync_buf);
        nsync_buf = nullptr; nsync_c = 0;
        const size_t cap = std::max(nnewton * 2, (size_t)256);
        HIPCHK(hipMalloc((void **)&nsync_buf, cap * NEWTON_SYNC_DOUBLES * sizeof(double)));
        // granule tags are (launch number << 10) + evaluation: a fresh block must not hold a matching tag by accident
        HIPCHK(hipMemsetAsync(nsync_buf, 0, cap * NEWTON_SYNC_DOUBLES * sizeof(double), ctx->stream));
        { if (int rc_ = ctx->sync(ctx->stream)) return rc_; }
        nsync_c = cap;
    }
    if (nnewton && !d_nctl) {
        HIPCHK(hipMalloc((void **)&d_nctl, 2 * sizeof(NewtonCtl)));
        // ON THE BATCH'S STREAM and waited for: a null-stream hipMemset returns before it has run and is not ordered against the
        // non-blocking streams the kernels use -- landing inside the first k_newton it would re-issue tickets and leave the
        // counters un-armed for the next launch (seen as time-outs and a memory fault when several batches shared the device)
        HIPCHK(hipMemsetAsync(d_nctl, 0, 2 * sizeof(NewtonCtl), ctx->stream));
        { if (int rc_ = ctx->sync(ctx->stream)) return rc_; }
    }
    const hipStream_t st = lane ? ctx->stream2 : ctx->stream;
    double *const frags_buf = lane ? d_frags2 : d_frags;
    ctx->tic_stream = st;
    const size_t ngenes = genes.size();
    // runs of the launch: one per (gene, part) -- parts of a gene are independent of each other (PendingOp::part)
    std::vector<int> &nparts = run_nparts; nparts.assign(ngenes, 1);
    for (auto &o : ops) nparts[o.gene] = std::max(nparts[o.gene], o.part + 1);
    for (auto &t : tails) nparts[t.gene] = std::max(nparts[t.gene], t.part + 1);
    std::vector<size_t> &koff = run_koff; koff.assign(ngenes + 1, 0);
    for (size_t g = 0; g < ngenes; ++g) koff[g + 1] = koff[g] + (size_t)nparts[g];
    const size_t nkeys = koff[ngenes];
    const size_t o_req = 0;
    const size_t o_ops = align_up(o_req + nreq_max * sizeof(PmatReq), 256);
    const size_t o_runs = align_up(o_ops + (nops + ntail) * sizeof(NvOp), 256);
    const size_t o_red = align_up(o_runs + nkeys * sizeof(GeneRun), 256);
    const size_t o_newt = align_up(o_red + neval * sizeof(ReduceReq), 256);
    size_t ntick_max = 0;                          // k_newton tickets: one per (request, slice)
    for (auto &t : tails) if (t.mode == MODE_SUMTABLE) ntick_max += (size_t)newton_split(genes[t.gene].aln.mpad);
    const size_t o_tick = align_up(o_newt + nnewton * sizeof(NewtonReq), 256);
    const size_t bytes = align_up(o_tick + ntick_max * sizeof(int), 256);
    if (chain && chain_off + bytes > h_cap) { if (int rc = chain_sync()) return rc; }     // ring full: drain, start over
    if (int rc = ensure_stage(bytes)) return rc;
    const size_t base = chain ? chain_off : 0;
    if (chain) chain_off += bytes;
    char *hs = (char *)h_stage + base, *ds = (char *)d_stage + base;
    PmatReq *hreq = (PmatReq *)(hs + o_req);
    NvOp *hops = (NvOp *)(hs + o_ops);
    GeneRun *hruns = (GeneRun *)(hs + o_runs);
    ReduceReq *hred = (ReduceReq *)(hs + o_red);
    NewtonReq *hnewt = (NewtonReq *)(hs + o_newt);
    int *htick = (int *)(hs + o_tick);
    if (nnewton) ++newton_launch_seq;
    const unsigned tag_base = (newton_launch_seq & 0x3FFFFFu) << 10;
    // fused branch Newton (kernels.h OPF_FUSED_NEWTON): a gene's only Newton tail, sitting behind all of its operations, is
    // iterated inside k_oplist<11> on the register-resident sumtable; PML_NO_FUSE=1 is the A-B arm, safe mode (after an exchange
    // gave up) runs unfused through the no-exchange k_newton form
    static const bool fuse_env = std::getenv("PML_NO_FUSE") == nullptr && !(std::getenv("PML_CHAIN") && std::atoi(std::getenv("PML_CHAIN")) == 0);   // PML_CHAIN=0: the plain kernel only
    // (genes of more than 32 tiles: kernels.hip launch_oplist -- one launch with one ticket partition over the device by default;
    // with PML_FUSE_BIG=0 only when the whole launch is resident at once: cut into several resident launches a step pays the
    // Newton latency once per launch, measured slower than un-fused on a C4 shard)
    bool fuse_ok = fuse_env && !newton_safe_mode();
    if (fuse_ok && nnewton && !fuse_big_genes()) {
        int mm = 0; size_t nr = 0;
        std::vector<char> seen(genes.size(), 0);
        for (auto &o : ops) if (!seen[o.gene]) { seen[o.gene] = 1; nr += (size_t)nparts[o.gene]; mm = std::max(mm, genes[o.gene].aln.mpad); }
        for (auto &t : tails) if (!seen[t.gene]) { seen[t.gene] = 1; nr += (size_t)nparts[t.gene]; mm = std::max(mm, genes[t.gene].aln.mpad); }
        const size_t bpg = (size_t)(mm + TILE_PAT - 1) / TILE_PAT;
        if (bpg > 32 && nr * bpg > (size_t)fused_oplist_capacity()) fuse_ok = false;
    }
    std::vector<char> fused_req(nnewton, 0);
    bool any_fused = false;
    std::vector<int> tail_req(ntail, -1);          // tail -> index of its NewtonReq (failure handling below)

    size_t nout = 0, nruns = 0, ie = 0, in = 0, ireq = 0, iop = 0;
    last_src.clear();
    int max_mpad = 0, newton_maxm = 0;
    double algo_bytes = 0, algo_flops = 0;
    // one transition-matrix request (fragment set or tip table) for branch (v, slot q) of gene g
    // one request per (gene, kind, tree branch) and launch; never while a plan is being recorded (a replay refreshes
    // each request from ITS branch)
    // keyed requests live in a flat table stamped with the launch number (no hashing, nothing to clear): one entry per
    // (gene, kind, directed branch slot); the undirected branch is the smaller of its two slots
    bool req_ok = req_off.size() == genes.size() + 1;
    for (size_t g = 0; req_ok && g < genes.size(); ++g) req_ok = req_off[g + 1] - req_off[g] == (size_t)9 * genes[g].tree.nnodes();
    if (!req_ok) {
        req_off.assign(genes.size() + 1, 0);
        for (size_t g = 0; g < genes.size(); ++g) req_off[g + 1] = req_off[g] + (size_t)3 * 3 * genes[g].tree.nnodes();
        req_stamp.assign(req_off.back(), 0); req_ptr.assign(req_off.back(), nullptr); req_launch = 0;
        val_bucket.assign(genes.size() * 3, {}); val_stamp.assign(genes.size() * 3, 0);
    }
    if (++req_launch == 0) {                    // the 32-bit launch number wrapped: no stale stamp may match it
        std::fill(req_stamp.begin(), req_stamp.end(), 0u); std::fill(val_stamp.begin(), val_stamp.end(), 0u); req_launch = 1;
    }
    // requests whose length belongs to no branch of the tree (SPR: joined / halved branches) are shared by VALUE: the
    // same (gene, kind, length) gives the same matrices: one list of (length bits, matrices) per (gene, kind).
    bool req_overflow = false;
    auto add_req = [&](size_t g, double t, int kind, int v, int q) -> const double * {
        uint64_t tbits = 0; size_t key = 0;
        std::vector<std::pair<uint64_t, const double *>> *bucket = nullptr;
        if (!record_plan) {
            if (v >= 0) {
                const Tree &T = genes[g].tree;
                const int w = T.nbr[v][q], a = v * 3 + q, b = w * 3 + T.slot(w, v);
                key = req_off[g] + (size_t)kind * 3 * T.nnodes() + (size_t)std::min(a, b);
                if (req_stamp[key] == req_launch) return req_ptr[key];
            } else {
                std::memcpy(&tbits, &t, 8);
                const size_t bi = g * 3 + (size_t)kind;                              // exact (gene, kind); a few hundred lengths at most
                if (val_stamp[bi] != req_launch) { val_bucket[bi].clear(); val_stamp[bi] = req_launch; }
                bucket = &val_bucket[bi];
                for (auto &e : *bucket) if (e.first == tbits) return e.second;
            }
        }
        if (ireq >= nreq_max) { req_overflow = true; return frags_buf; }
        PmatReq &r = hreq[ireq];
        r.t = t; std::memcpy(r.rates, genes[g].rates, sizeof r.rates); r.kind = kind; r.pad = 0;
        r.tp = (chain && v >= 0 && genes[g].len_pending[(size_t)v * 3 + q]) ? genes[g].d_len + (size_t)v * 3 + q : nullptr;
        r.md = d_gmodel ? d_gmodel + g : nullptr;
        last_src.push_back({(int)g, v, q, kind});
        const double *out = frags_buf + (ireq++) * FRAG_STRIDE;
        if (!record_plan) { if (v >= 0) { req_stamp[key] = req_launch; req_ptr[key] = out; } else bucket->push_back({tbits, out}); }
        return out;
    };
    // resolves one side of an op: pointers, kind, scaling counts; `want_table`: newview tip sides look
    // their contraction up in a tip table; `t_branch`/(bv,bq): the branch between this side and the op
    // node (fragment request), unless the caller supplies fixed matrices
    // bytes / flops: SURVEY 8d's per-operation figures (newview inner-inner 1920 B / 6480 flop, tip-inner 1281 B / 3280 flop,
    // tip-tip 642 B / 80 flop, evaluate 1280 B / 3360 flop per pattern).  `inner`: the side counts as an inner child of the
    // operation; `flops`: work 8d assigns to producing a side that is never materialised (virtual cherry = one tip-tip newview,
    // virtual pitchfork = that + one tip-inner newview)
    struct Resolved { OpSide s; int kind; const int *scl; double bytes; double flops = 0; bool inner = true; };
    auto resolve = [&](size_t g, const Side &sd, Resolved &R) -> int {
        Gene &G = genes[g];
        const int mp = G.aln.mpad, nt = G.aln.ntax;
        R.s = OpSide{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}; R.scl = nullptr;
        R.flops = 0; R.inner = true;
        if (sd.kind == SIDE_TIP) { R.kind = SK_TIP; R.s.p0 = G.d_codes + (size_t)sd.id * mp; R.bytes = 1; R.inner = false; return 0; }
        if (sd.kind == SIDE_CHERRY) {
            const int v = nt + sd.id / 3, k = sd.id % 3;
            int tips[2], qs[2], ci = 0;
            for (int q = 0; q < 3; ++q) if (q != k) { tips[ci] = G.tree.nbr[v][q]; qs[ci] = q; ++ci; }
            R.kind = SK_CHERRY;
            R.s.p0 = G.d_codes + (size_t)tips[0] * mp; R.s.p1 = G.d_codes + (size_t)tips[1] * mp;
            R.s.t0 = add_req(g, G.tree.len[v][qs[0]], PM_TIPTABLE, v, qs[0]);
            R.s.t1 = add_req(g, G.tree.len[v][qs[1]], PM_TIPTABLE, v, qs[1]);
            R.bytes = 640 + 642;       // SURVEY 8d accounting: the tip-tip newview (642 B) + reading its CLV (640 B)
            R.flops = 80;
            return 0;
        }
        if (sd.kind == SIDE_PITCH) {
            // X over (cherry C = tips a,b ; tip c): tables of a,b,c + the fragments of branch X-C
            const int X = nt + sd.id / 3, k = sd.id % 3;
            int qc = -1, qC = -1;
            for (int q = 0; q < 3; ++q) if (q != k) { if (G.tree.nbr[X][q] < nt) qc = q; else qC = q; }
            const int C = G.tree.nbr[X][qC], kC = G.tree.slot(C, X);
            int tips[2], qs[2], ci = 0;
            for (int q = 0; q < 3; ++q) if (q != kC) { tips[ci] = G.tree.nbr[C][q]; qs[ci] = q; ++ci; }
            R.kind = SK_PITCH;
            R.s.p0 = G.d_codes + (size_t)tips[0] * mp; R.s.p1 = G.d_codes + (size_t)tips[1] * mp;
            R.s.p2 = G.d_codes + (size_t)G.tree.nbr[X][qc] * mp;
            R.s.t0 = add_req(g, G.tree.len[C][qs[0]], PM_TIPTABLE, C, qs[0]);
            R.s.t1 = add_req(g, G.tree.len[C][qs[1]], PM_TIPTABLE, C, qs[1]);
            R.s.t2 = add_req(g, G.tree.len[X][qc], PM_TIPTABLE, X, qc);
            R.s.f = add_req(g, G.tree.len[X][qC], PM_FRAGS, X, qC);
            any_pitch = true;
            R.bytes = 640 + (640 + 642) + 1 + 640;   // read X + X's newview (cherry child, tip child, write)
            R.flops = 80 + 3280;
            return 0;
        }
        const int slot = sd.kind == SIDE_MSG ? G.slot_of[sd.id] : G.slot_cap + sd.id;
        if (slot < 0) return -1;
        R.kind = SK_CLV; R.s.p0 = G.d_clv + (size_t)slot * clv_doubles(mp); R.scl = G.d_scl + (size_t)slot * mp; R.bytes = 640;
        return 0;
    };

    // tails by gene, in submission order (<= MAXTAIL per gene per run)
    std::vector<std::vector<int>> &tails_of = run_tails_of;          // kept between launches: no allocations per launch
    if (tails_of.size() < nkeys) tails_of.resize(nkeys);
    for (auto &v : tails_of) v.clear();
    for (size_t i = 0; i < ntail; ++i) {
        if (tails[i].slot < 0 || tails[i].slot >= MAXTAIL) return ctx->fail(-1, "internal: bad tail slot");
        tails_of[koff[tails[i].gene] + (size_t)tails[i].part].push_back((int)i);
    }
    for (size_t g = 0; g < ngenes; ++g) for (int part = 0; part < nparts[g]; ++part) {
        const size_t key = koff[g] + (size_t)part;
        const bool has_ops = iop < nops && ops[iop].gene == (int)g && ops[iop].part == part;
        if (!has_ops && tails_of[key].empty()) continue;
        Gene &G = genes[g];
        const int mp = G.aln.mpad;
        GeneRun &run = hruns[nruns++];
        run.op_begin = (int)nout;
        max_mpad = std::max(max_mpad, mp);
        // Register chaining (kernels.h OPF_CHAIN_*): the gene's last newview result is still in the registers of the wave
        // that owns the patterns.  Every launch takes a child that the directly following operation of the gene consumes
        // from there instead of reading it back; whole-tree scoring passes (record_plan), whose results nobody reads
        // again, do not even write such a child (it stays invalid in memory and is recomputed if a later request wants it).
        // PML_CHAIN (A-B switch): 2 = that (default), 1 = scoring passes only, 0 = off.  Measured on one box, rotated order
        // (profiles/r02_ab_register_chaining.txt): C3 scoring launch 0.89 -> 0.72 ms, C4 shard 3.40 -> 2.92 ms, C3 search
        // 153 -> 161 gene-trees/s.
        static const int chain_env = std::getenv("PML_CHAIN") ? std::atoi(std::getenv("PML_CHAIN")) : 2;
        const bool chain_reads = chain_env == 2 || (chain_env == 1 && record_plan), chain_nostore = chain_env >= 1 && record_plan && !record_stored;
        long last_nv = -1, last_nv_op = -1;            // hops / ops index of the gene's last newview
        auto chained_from = [&](const Side &sd, int kind) {
            return chain_reads && last_nv >= 0 && kind == SK_CLV && sd.kind == ops[last_nv_op].out_kind && sd.id == ops[last_nv_op].out_id;
        };
        auto consume_last = [&]() {                     // the last newview's result is taken from the registers by the operation being built
            any_chain = true;
            if (chain_nostore || ops[last_nv_op].transient) { hops[last_nv].flags |= OPF_NO_STORE; ops[last_nv_op].unstored = true; }
        };
        auto emit_tail = [&](const Tail &t) -> int {
            NvOp &d = hops[nout++];
            std::memset(&d, 0, sizeof d);
            Resolved L, R;
            if (resolve(g, t.a, L) || resolve(g, t.b, R)) return ctx->fail(-5, "internal: tail message has no slot");
            d.l = L.s; d.r = R.s; d.l_scl = L.scl; d.r_scl = R.scl;
            d.flags = L.kind | (R.kind << 2); d.mpad = mp; d.mode = t.mode;
            if (chained_from(t.a, L.kind)) { d.flags |= OPF_CHAIN_L; consume_last(); }
            else if (t.mode >= MODE_EVALUATE && chained_from(t.b, R.kind)) { d.flags |= OPF_CHAIN_R; consume_last(); }
            double *result = t.result_dev ? t.result_dev : d_scalars + 8 * (g * MAXTAIL + t.slot);
            if (t.mode == MODE_EVALUATE_CAT) {
                if (!t.patlnl_dev || !t.scl_dev) return ctx->fail(-1, "internal: table slice missing");
                d.pl = d.pr = add_req(g, t.t0, PM_FRAGS_PI, t.bv, t.bq);
                d.out = t.patlnl_dev; d.out_scl = t.scl_dev;
                algo_bytes += (double)G.aln.npat * (L.bytes + R.bytes + 36);
                algo_flops += (double)G.aln.npat * (3360 + L.flops + R.flops);
            } else if (t.mode == MODE_EVALUATE) {
                d.pl = d.pr = add_req(g, t.t0, PM_FRAGS_PI, t.bv, t.bq);
                double *pl = t.patlnl_dev ? t.patlnl_dev : G.d_patlnl[t.slot];      // pooled when a gene has more than MAXTAIL tails
                d.out = pl; d.out_scl = nullptr;
                ReduceReq &rr = hred[ie++];
                rr.patlnl = pl; rr.weight = G.d_weight; rr.out = result; rr.mpad = mp; rr.pad = 0;
                algo_bytes += (double)G.aln.npat * (L.bytes + R.bytes + 8);
                algo_flops += (double)G.aln.npat * (3360 + L.flops + R.flops);
            } else {
                d.pl = eig_of((int)g); d.pr = d.pl + PFRAG;
                double *stab = t.sumtab_dev ? t.sumtab_dev : G.d_sumtab[t.slot];
                int *sscl = t.sumtab_dev ? reinterpret_cast<int *>(t.sumtab_dev + clv_doubles(mp)) : G.d_sumscl[t.slot];
                d.out = stab; d.out_scl = sscl;
                NewtonReq &nr = hnewt[in];
                nr.md = model_of((int)g); nr.tag_base = tag_base; nr.pad0 = 0;
                // (any number of tails per gene, anywhere in its list: every request has its own exchange block, the gene's workgroups
                // walk the list in step.  An NNI round -- three tails per internal edge -- then neither writes nor re-reads its pooled
                // sumtables, 640 B per pattern and tail.  PML_FUSE_MULTI=0: only a gene's single, last tail, the A-B arm)
                static const bool fuse_multi = !(std::getenv("PML_FUSE_MULTI") && std::atoi(std::getenv("PML_FUSE_MULTI")) == 0);
                if (fuse_ok && (fuse_multi || (tails_of[key].size() == 1 && t.after < 0)) && !t.patlnl_dev && newton_reg_form(mp)) {
                    d.flags |= OPF_FUSED_NEWTON; d.aux = (const NewtonReq *)(ds + o_newt) + in;
                    fused_req[in] = 1; any_fused = true; any_chain = true;
                }
                last_nv = -1; last_nv_op = -1;              // a sumtable operation leaves ITS tile in the wave's registers (kernels.hip chunk_op): the chain ends here
                tail_req[&t - tails.data()] = (int)in;
                nr.ticket0 = 0; nr.pad = 0;
                nr.sumtab = stab; nr.weight = G.d_weight; nr.scl = sscl;
                std::memcpy(nr.rates, G.rates, sizeof nr.rates);
                nr.t0 = t.t0; nr.tol = newton_tol; nr.out = result; nr.mpad = mp; nr.max_iter = t.max_iter;
                nr.t_dev0 = t.t_dev0; nr.t_dev1 = t.t_dev1; nr.patlnl = t.patlnl_dev;
                nr.sync = nsync_buf + (size_t)in * NEWTON_SYNC_DOUBLES; newton_maxm = std::max(newton_maxm, mp);
                algo_bytes += (double)G.aln.npat * (L.bytes + R.bytes + 640);
                algo_flops += (double)G.aln.npat * (6480 + L.flops + R.flops);      // the newview contraction with the eigen-basis matrices
                in++;
            }
            return 0;
        };
        size_t ti = 0; int emitted = 0;
        auto flush_tails = [&](bool all) -> int {
            while (ti < tails_of[key].size()) {
                const Tail &t = tails[tails_of[key][ti]];
                if (!all && (t.after < 0 || t.after > emitted)) break;
                if (int rc = emit_tail(t)) return rc;
                ++ti;
            }
            return 0;
        };
        if (int rc = flush_tails(false)) return rc;
        for (; iop < nops && ops[iop].gene == (int)g && ops[iop].part == part; ++iop) {
            PendingOp &o = ops[iop];
            const int s = o.out_kind == SIDE_MSG ? slot_for(G, o.out_id) : G.slot_cap + o.out_id;
            if (s < 0) return ctx->fail(-4, "CLV slots exhausted (score-only batch used for a multi-root request)");
            NvOp &d = hops[nout++];
            std::memset(&d, 0, sizeof d);
            d.mode = MODE_NEWVIEW;
            d.out = G.d_clv + (size_t)s * clv_doubles(mp);
            d.out_scl = G.d_scl + (size_t)s * mp;
            Resolved S[2];
            if (resolve(g, o.child[0], S[0]) || resolve(g, o.child[1], S[1])) return ctx->fail(-5, "internal: child message has no slot");
            d.mpad = mp;
            d.flags = S[0].kind | (S[1].kind << 2);
            const double *pm[2];
            for (int c = 0; c < 2; ++c) {
                // where this child's branch length lives (plan replay): output message (v, k), child c
                int bv = o.bv[c], bq = o.bq[c];
                if (o.out_kind == SIDE_MSG) {
                    bv = G.aln.ntax + o.out_id / 3; const int k = o.out_id % 3;
                    int seen = 0;
                    for (bq = 0; bq < 3; ++bq) if (bq != k) { if (seen == c) break; ++seen; }
                }
                pm[c] = add_req(g, o.t[c], PM_FRAGS, bv, bq);
            }
            // the child that is the gene's previous result goes LEFT (the two factors of a newview commute bit for bit)
            // (the PendingOp itself keeps its order: t[], bv[], bq[] belong to its children by position, and a launch set that has to
            // be issued again -- run()'s retry in safe mode -- must find it unchanged)
            Side ch[2] = {o.child[0], o.child[1]};
            if (chained_from(ch[1], S[1].kind)) { std::swap(S[0], S[1]); std::swap(pm[0], pm[1]); std::swap(ch[0], ch[1]); }
            d.flags = S[0].kind | (S[1].kind << 2);
            if (chained_from(ch[0], S[0].kind)) { d.flags |= OPF_CHAIN_L; consume_last(); }
            d.l = S[0].s; d.r = S[1].s; d.l_scl = S[0].scl; d.r_scl = S[1].scl; d.pl = pm[0]; d.pr = pm[1];
            algo_bytes += (double)G.aln.npat * (S[0].bytes + S[1].bytes + 640);
            algo_flops += (double)G.aln.npat * ((S[0].inner && S[1].inner ? 6480 : (S[0].inner || S[1].inner ? 3280 : 80)) + S[0].flops + S[1].flops);
            last_nv = (long)nout - 1; last_nv_op = (long)iop;
            ++emitted;
            if (int rc = flush_tails(false)) return rc;
        }
        if (int rc = flush_tails(true)) return rc;
        run.op_end = (int)nout;
        // Cache policy of the CLV stores, per gene.  Measured on one box, rotated order (profiles/r02_ab_nontemporal.txt): with
        // NON-TEMPORAL stores the C3 scoring launch (8 tiles per gene) takes 0.89 ms instead of 0.98 -- written CLVs no longer
        // push the transition-matrix fragments and tip tables out of L2 / Infinity Cache, which 8 workgroups per gene re-fetch
        // for every operation -- while the C4 shard (40 tiles per gene: 40 workgroups share each fragment set, and a parent
        // often finds its child's CLV still in the Infinity Cache) takes 10.15 ms instead of 9.18.  Hence by gene size.
        // PML_NT_STORE=0 never / 2 always / 3 all but results the next operation reads: A-B arms.
        static const int nt_policy = std::getenv("PML_NT_STORE") ? std::atoi(std::getenv("PML_NT_STORE")) : 1;
        const bool small_gene = mp <= 16 * TILE_PAT;
        for (int i = run.op_begin; i < run.op_end; ++i) {
            NvOp &d = hops[i];
            if (d.mode != MODE_NEWVIEW || nt_policy == 0 || (nt_policy == 1 && !small_gene)) continue;
            bool next_reads = false;
            if (nt_policy == 3 && i + 1 < run.op_end) {
                const NvOp &nx = hops[i + 1];
                next_reads = ((nx.flags & 3) == SK_CLV && nx.l.p0 == d.out) || (((nx.flags >> 2) & 3) == SK_CLV && nx.r.p0 == d.out);
            }
            if (!next_reads) d.flags |= OPF_NT_STORE;
        }
    }

    if (req_overflow) return ctx->fail(-5, "internal: transition-matrix request bound exceeded");
    // k_newton's ticket table: (request, slice) in request order, register-form requests first, then the streaming-form ones
    // (genes of more than 8192 patterns); in safe mode (SEQ form) one entry per request instead
    int nt_reg = 0, nt_stream = 0;
    const bool seq_launch = nnewton > 0 && newton_safe_mode();
    {
        int cur = 0;
        for (int pass = 0; pass < 2; ++pass) {
            for (size_t i = 0; i < nnewton; ++i) {
                if (fused_req[i] || newton_reg_form(hnewt[i].mpad) != (pass == 0)) continue;
                const int S = seq_launch ? 1 : newton_split(hnewt[i].mpad);
                hnewt[i].ticket0 = cur - (pass == 0 ? 0 : nt_reg);            // relative to its kernel's table
                for (int k = 0; k < S; ++k) htick[cur++] = (int)i;
            }
            if (pass == 0) nt_reg = cur; else nt_stream = cur - nt_reg;
        }
        if (nnewton && safe_left > 0 && !safe_now) --safe_left;
    }
    double newton_bytes = 0;
    for (size_t i = 0; i < ntail; ++i) if (tails[i].mode == MODE_SUMTABLE && !fused_req[tail_req[i]]) newton_bytes += (double)genes[tails[i].gene].aln.npat * 640;
    Deferred L;
    L.base = base; L.bytes = bytes; L.o_req = o_req; L.o_ops = o_ops; L.o_runs = o_runs; L.o_red = o_red; L.o_newt = o_newt;
    L.o_tick = o_tick; L.nt_reg = nt_reg; L.nt_stream = nt_stream; L.seq = seq_launch; L.fused = any_fused;
    L.nreq = ireq; L.nruns = nruns; L.neval = neval; L.nnewton = nnewton; L.max_mpad = max_mpad; L.newton_maxm = newton_maxm;
    L.any_pitch = any_pitch; L.any_chain = any_chain; L.algo_bytes = algo_bytes; L.algo_flops = algo_flops; L.newton_bytes = newton_bytes; L.lane = lane; L.stagger = record_stagger;
    record_stagger = false;
    // Inside a chained pass the upload + launches of a step are DEFERRED and issued in groups (1, 2, 4, 8, 8, ... steps):
    // a host-to-device copy between two kernels of one stream costs a ~20 us bubble on the compute queue (measured:
    // 2866 newton -> pmat gaps of 21 us in a C3 search, profiles/r02b), one copy per group leaves a handful per pass.
    // The host keeps building the next group while the device works on the last one.
    deferred.push_back(L);
    const bool defer = chain && !lanes_active;
    if (!defer || deferred.size() >= flush_quota) { if (int rc = flush_deferred()) return rc; }
    ctx->tic_stream = nullptr;
    const double t_launched = now_ms();
    ctx->stats[K_HOST_BUILD].launches++; ctx->stats[K_HOST_BUILD].ms += t_launched - t_begin;
    if (!chain) {
        bool pooled = false;
        for (auto &t : tails) pooled = pooled || t.result_host != nullptr;
        if (int rc = fetch_results(pooled)) return rc;
        { if (int rc_ = ctx->sync(ctx->stream)) return rc_; }
        HIPCHK(hipGetLastError());
        const double t_done = now_ms();
        ctx->stats[K_HOST_WAIT].launches++; ctx->stats[K_HOST_WAIT].ms += t_done - t_launched;
        host_phase_ms[HP_RUN_SYNCED] += t_launched - t_begin;           // descriptor build of a launch the device waited for
        ctx->resolve_events();
        // A Newton request whose cross-workgroup exchange gave up reports lnL = NaN (k_newton).  Its sumtable is still in
        // place: the affected requests are re-issued through the no-exchange SEQ form (one workgroup walks the slices; the
        // bits of the split form), here, before anybody consumes a result.
        auto result_of = [&](const Tail &t) { return t.result_host ? t.result_host : (t.result_dev ? (const double *)nullptr : res(t.gene, t.slot)); };
        std::vector<int> bad;
        for (size_t i = 0; i < ntail; ++i) {
            const Tail &t = tails[i];
            if (t.mode != MODE_SUMTABLE) continue;
            const double *h = result_of(t);
            if (h && !std::isfinite(h[1])) bad.push_back(tail_req[i]);
        }
        bool bad_fused = false;
        for (int i : bad) bad_fused = bad_fused || fused_req[i];
        if (bad_fused && !in_retry) {
            // a fused request has no stored sumtable to iterate on: the whole launch set runs again, unfused, through the
            // no-exchange form (newton_gave_up() puts the batch in safe mode); CLV results are recomputed to the same bits
            newton_gave_up(); ctx->newton_reissued += (long long)bad.size();
            if (int rc = clear_abort()) return rc;
            for (auto &o : ops) { o.unstored = false; if (o.out_kind == SIDE_MSG) genes[o.gene].pend_level[o.out_id] = -1; }
            in_retry = true;
            const int rc = run(ops, tails);
            in_retry = false;
            return rc;
        }
        if (!bad.empty() && !seq_launch && !bad_fused) {
            newton_gave_up(); ctx->newton_reissued += (long long)bad.size();
            int nr = 0, ns = 0;
            for (int pass = 0; pass < 2; ++pass) for (int i : bad) if (newton_reg_form(hnewt[i].mpad) == (pass == 0)) { htick[nr + ns] = i; ++(pass == 0 ? nr : ns); }
            if (int rc = clear_abort()) return rc;
            HIPCHK(hipMemcpyAsync(ds + o_tick, hs + o_tick, bad.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
            launch_newton_seq(nullptr, (const NewtonReq *)(ds + o_newt), (const int *)(ds + o_tick), nr, ns, d_nctl + lane, ctx->stream);
            ++ctx->newton_seq_launches;
            if (int rc = fetch_results(pooled)) return rc;
            { if (int rc_ = ctx->sync(ctx->stream)) return rc_; }
            HIPCHK(hipGetLastError());
        }
        for (auto &t : tails) {
            if (t.mode == MODE_EVALUATE_CAT) continue;
            const double *h = result_of(t);
#ifndef ABL_KEEP_GOING      // timing-only ablation builds (tools/ab_w1_ablation.sh) compute garbage on purpose
            if (h && !std::isfinite(t.mode == MODE_EVALUATE ? h[0] : h[1]))
                return ctx->fail(-5, t.mode == MODE_EVALUATE ? "device returned a non-finite likelihood" : "k_newton: non-finite branch likelihood (also from the no-exchange form)");
#endif
        }
    }
    for (auto &o : ops) if (o.out_kind == SIDE_MSG) { Gene &G = genes[o.gene]; G.valid[o.out_id] = o.unstored ? 0 : 1; G.pend_level[o.out_id] = -1; }
    if (record_plan) {                       // keep the descriptors of this full-traversal score
        record_plan = false;
        Plan &P = plan;
        if (P.bytes < bytes) {
            if (P.h) hipHostFree(P.h);
            if (P.d) hipFree(P.d);
            P.h = P.d = nullptr; P.bytes = 0;
            HIPCHK(hipHostMalloc(&P.h, bytes)); HIPCHK(hipMalloc(&P.d, bytes)); P.bytes = bytes;
        }
        std::memcpy(P.h, hs, bytes);
        // ON THE BATCH'S STREAM: a device-to-device hipMemcpy returns before the copy ran and the null stream is not
        // ordered against the (non-blocking) stream a replay uploads its refreshed requests on -- the late copy then
        // put the recorded rates back under the first replay (DESIGN r02-g: the cause of the rare different optimum)
        HIPCHK(hipMemcpyAsync(P.d, ds, bytes, hipMemcpyDeviceToDevice, ctx->stream));
        P.o_req = o_req; P.o_ops = o_ops; P.o_runs = o_runs; P.o_red = o_red;
        P.nreq = ireq; P.nruns = nruns; P.neval = neval; P.max_mpad = max_mpad; P.algo_bytes = algo_bytes; P.algo_flops = algo_flops; P.stored = record_stored; P.any_pitch = any_pitch; P.any_chain = any_chain;
        P.rates_seen.resize(genes.size()); for (size_t g = 0; g < genes.size(); ++g) P.rates_seen[g] = genes[g].rates_epoch;
        P.src = last_src; P.outs.clear();
        for (auto &o : ops) if (o.out_kind == SIDE_MSG && !o.unstored) P.outs.push_back({o.gene, o.out_id});
        P.epoch = topo_epoch; P.valid = true;
    }
    return 0;
}

// full-traversal score of all genes from cached descriptors: refresh branch lengths / rates, then
// k_pmat + k_oplist + k_reduce exactly as run() would launch them
int Batch::replay_plan(double *lnl) {
    const double t_begin = now_ms();
    Plan &P = plan;
    HIPCHK(hipSetDevice(ctx->device));
    PmatReq *hreq = (PmatReq *)((char *)P.h + P.o_req);
    std::vector<char> moved(genes.size(), 0);            // rates are rewritten only for genes whose alpha changed
    for (size_t g = 0; g < genes.size(); ++g) if (P.rates_seen[g] != genes[g].rates_epoch) { moved[g] = 1; P.rates_seen[g] = genes[g].rates_epoch; }
    bool changed = false;                       // lengths and rates already on the device are not uploaded again
    for (size_t i = 0; i < P.nreq; ++i) {
        const ReqSrc &s = P.src[i];
        const Gene &G = genes[s.gene];
        const double t = G.tree.len[s.v][s.q];
        if (hreq[i].t != t) { hreq[i].t = t; changed = true; }
        if (moved[s.gene]) { std::memcpy(hreq[i].rates, G.rates, sizeof hreq[i].rates); changed = true; }
    }
    char *ds = (char *)P.d;
    if (changed) HIPCHK(hipMemcpyAsync(ds + P.o_req, hreq, P.nreq * sizeof(PmatReq), hipMemcpyHostToDevice, ctx->stream));
    const ModelDev *md = pi_mode < 2 ? ctx->d_model[pi_mode] : nullptr;      // per-gene models travel in the requests
    ctx->tic(K_PMAT, (double)P.nreq * PFRAG * 8);
    launch_pmat(md, (const PmatReq *)(ds + P.o_req), d_frags, (int)P.nreq, ctx->stream, d_gmodel != nullptr);
    ctx->toc();
    ctx->tic(K_NEWVIEW, P.algo_bytes, P.algo_flops);
    launch_oplist((const NvOp *)(ds + P.o_ops), (const GeneRun *)(ds + P.o_runs), (int)P.nruns, P.max_mpad, P.any_pitch, P.any_chain, ctx->stream);
    ctx->toc();
    ctx->tic(K_REDUCE, 0);
    launch_reduce((const ReduceReq *)(ds + P.o_red), (int)P.neval, ctx->stream);
    ctx->toc();
    const double t_launched = now_ms();
    ctx->stats[K_HOST_BUILD].launches++; ctx->stats[K_HOST_BUILD].ms += t_launched - t_begin;
    if (!chain) {
        if (int rc = fetch_results(false)) return rc;
        { if (int rc_ = ctx->sync(ctx->stream)) return rc_; }
        HIPCHK(hipGetLastError());
        const double t_done = now_ms();
        ctx->stats[K_HOST_WAIT].launches++; ctx->stats[K_HOST_WAIT].ms += t_done - t_launched;
        ctx->resolve_events();
    }
    for (auto &o : P.outs) genes[o.first].valid[o.second] = 1;
    for (size_t g = 0; g < genes.size(); ++g) lnl[g] = res((int)g)[0];
#ifndef ABL_KEEP_GOING
    if (!chain) for (size_t g = 0; g < genes.size(); ++g) if (!std::isfinite(lnl[g])) return ctx->fail(-5, "device returned a non-finite likelihood");
#endif
    for (size_t g = 0; g < genes.size(); ++g) det_record(det_id, genes[g], 'R', 0, 0, lnl[g], genes[g].alpha, 0);
    return 0;
}

// ------------------------------------------------------------------------------------------
// requests
// ------------------------------------------------------------------------------------------
int Batch::evaluate(const std::vector<char> &active, double *lnl) {
    ++cnt_eval;
    std::vector<PendingOp> ops; std::vector<Tail> tails;
    for (int g = 0; g < (int)genes.size(); ++g) {
        if (!active.empty() && !active[g]) continue;
        Gene &G = genes[g];
        const int r = G.tree.nbr[0][0];
        need(g, r, 0, ops);
        tails.push_back({g, msg(g, 0, r), msg(g, r, 0), MODE_EVALUATE, G.tree.len[0][0], 0, 0, -1, 0, 0});
    }
    if (int rc = run(ops, tails)) return rc;
    for (auto &t : tails) lnl[t.gene] = res(t.gene)[0];
    for (auto &t : tails) det_record(det_id, genes[t.gene], 'E', 0, 0, lnl[t.gene], genes[t.gene].alpha, 0);
    return 0;
}
int Batch::score(const std::vector<char> &active, double *lnl, bool stored) {
    for (int g = 0; g < (int)genes.size(); ++g) if (active.empty() || active[g]) invalidate_all(g);
    bool all = true;
    for (char a : active) all = all && a;
    if (all && !score_only_batch) {
        if (plan.valid && plan.epoch == topo_epoch && plan.stored == stored) return replay_plan(lnl);
        record_plan = true; record_stored = stored;
    }
    const int rc = evaluate(active, lnl);
    record_plan = false; record_stored = false;
    return rc;
}
int Batch::site_lnl(int g, double *out) {
    Gene &G = genes[g];
    if ((int)G.aln.site2pat.size() != G.aln.nsites) return ctx->fail(-1, "per-site lnL is not available for device-gathered replicates");
    std::vector<char> act(genes.size(), 0); act[g] = 1;
    std::vector<double> l(genes.size());
    if (int rc = evaluate(act, l.data())) return rc;
    std::vector<double> pat(G.aln.mpad);
    HIPCHK(hipMemcpy(pat.data(), G.d_patlnl[0], sizeof(double) * G.aln.mpad, hipMemcpyDeviceToHost));
    for (int s = 0; s < G.aln.nsites; ++s) out[s] = pat[G.aln.site2pat[s]];
    return 0;
}
int Batch::root_derivs(double *lnl, double *d1, double *d2) {
    std::vector<PendingOp> ops; std::vector<Tail> tails;
    for (int g = 0; g < (int)genes.size(); ++g) {
        Gene &G = genes[g];
        const int r = G.tree.nbr[0][0];
        need(g, r, 0, ops);
        tails.push_back({g, msg(g, 0, r), msg(g, r, 0), MODE_SUMTABLE, G.tree.len[0][0], 0});
    }
    if (int rc = run(ops, tails)) return rc;
    for (int g = 0; g < (int)genes.size(); ++g) { lnl[g] = res(g)[1]; d1[g] = res(g)[2]; d2[g] = res(g)[3]; }
    return 0;
}

// one Gauss-Seidel pass over the dirty branches (DFS from taxon 0, oracle order: eng_smooth_rec)
int Batch::smooth_pass(const std::vector<char> &active, std::vector<double> &maxdelta, double thr) {
    const int n = (int)genes.size();
    maxdelta.assign(n, 0.0);
    struct SafeOff { bool &f; ~SafeOff() { f = false; } } safe_off{safe_now};     // whichever way the pass is left
    double hp_t = now_ms();
    // per-gene DFS edge order, restricted to dirty branches
    // (scratch kept between passes: this set-up runs while the device is idle)
    std::vector<std::vector<std::pair<int, int>>> &order = pass_order;
    std::vector<std::vector<uint8_t>> &next = pass_next;
    if ((int)order.size() != n) { order.assign(n, {}); next.assign(n, {}); }
    size_t maxlen = 0;
    struct F { int v, from, k; };
    std::vector<F> st; st.reserve(256);
    for (int g = 0; g < n; ++g) {
        order[g].clear();
        if (!active[g]) { next[g].clear(); continue; }
        Gene &G = genes[g];
        const Tree &T = G.tree; const int nt = T.ntax;
        if (G.dirty.size() != (size_t)T.nnodes() * 3) G.mark_all();
        next[g].assign((size_t)T.nnodes() * 3, 0);
        // emulate the recursion: visit(v, from): for k: edge (v,w); if inner recurse
        // (edge list fixed up-front: topology does not change during a pass)
        st.clear(); st.push_back({0, -1, 0});
        while (!st.empty()) {
            F &f = st.back();
            if (f.k >= 3) { st.pop_back(); continue; }
            const int k = f.k, w = T.nbr[f.v][k]; f.k++;
            if (w < 0 || w == f.from) continue;
            if (G.dirty[f.v * 3 + k]) order[g].push_back({f.v, w});
            if (w >= nt) { const int fv = f.v; st.push_back({w, fv, 0}); }
        }
        maxlen = std::max(maxlen, order[g].size());
    }
    ++cnt_passes;
    host_phase_ms[HP_PASS_SETUP] += now_ms() - hp_t; hp_t = now_ms();
    // The whole pass is enqueued without a host round trip: a branch optimised at step i has its new length in
    // Gene::d_len (written by k_newton), and every later transition-matrix request across that branch reads it from
    // there (PmatReq::tp).  The host learns the new lengths after ONE synchronisation at the end of the pass.
    static const bool no_chain = std::getenv("PML_NO_CHAIN") != nullptr;
    size_t nres = 0; for (int g = 0; g < n; ++g) nres += order[g].size();
    struct Done { int gene, v, w; double old; size_t idx; };
    std::vector<Done> done; done.reserve(nres);
    // attempt 1 re-runs the WHOLE pass through the no-exchange Newton form when k_newton's exchange gave up somewhere in the
    // chained attempt 0: the host's branch lengths are still those of the pass start (new lengths live on the device until the
    // pass is accepted), the dirty flags are untouched, and CLVs computed from unaccepted lengths are invalidated -- so the
    // second attempt computes exactly what an untroubled pass computes
    for (int attempt = 0; attempt < 2; ++attempt) {
    done.clear();
    if (!no_chain) { if (int rc = chain_begin(nres)) return rc; }
    auto fail_out = [&](int rc) { if (chain) { chain_sync(); chain = false; } return rc; };
    // two lanes: with enough genes the pass is issued as two independent halves (even / odd genes) on two streams.
    // One half's latency-bound stretches (k_newton's cross-workgroup exchanges, k_pmat, kernel boundaries) then
    // overlap the other half's HBM-bound CLV updates.  The gene -> lane map is fixed for the whole pass, so a
    // gene's steps stay ordered on one stream.
    // OFF by default (PML_LANES=1 turns it on): the gain is 3-5 % at best and depends on the two streams landing on
    // different hardware queues -- HIP multiplexes same-priority streams onto 4 queues, and in a process with more
    // streams (torch, a second context) both lanes shared one queue and ran 25 % slower than a single lane; putting
    // lane 1 in another priority class fixed C3 but doubled the C4-shard search time (the high-priority lane starves
    // the other lane's spinning k_newton workgroups).
    static const bool lanes_on = std::getenv("PML_LANES") != nullptr;
    int nact = 0; for (int g = 0; g < n; ++g) nact += active[g] && !order[g].empty();
    const bool two_lanes = chain && lanes_on && nact >= 16;
    lanes_active = two_lanes;
    for (size_t step = 0; step < maxlen; ++step) {
        ++cnt_smooth;
        const size_t first = done.size();
        for (int ln = 0; ln < (two_lanes ? 2 : 1); ++ln) {
            std::vector<PendingOp> ops; std::vector<Tail> tails;
            for (int g = 0; g < n; ++g) {
                if (!active[g] || step >= order[g].size()) continue;
                if (two_lanes && (g & 1) != ln) continue;
                auto [v, w] = order[g][step];
                Gene &G = genes[g];
                need(g, v, w, ops); need(g, w, v, ops);
                Tail t{g, msg(g, v, w), msg(g, w, v), MODE_SUMTABLE, G.tree.len[v][G.tree.slot(v, w)], 32};
                if (chain) {
                    t.result_dev = d_chain + 4 * done.size();
                    t.t_dev0 = G.d_len + (size_t)v * 3 + G.tree.slot(v, w); t.t_dev1 = G.d_len + (size_t)w * 3 + G.tree.slot(w, v);
                }
                done.push_back({g, v, w, t.t0, done.size()});
                tails.push_back(t);
            }
            if (tails.empty()) continue;
            lane = ln;
            if (two_lanes && step == 0) {
                // stagger the lanes by one CLV-update kernel so that one lane's k_newton (latency-bound) runs
                // against the other's k_oplist (HBM-bound) instead of both doing the same thing at once
                if (!ev_stagger) hipEventCreateWithFlags(&ev_stagger, hipEventDisableTiming);
                if (ln == 0) record_stagger = true;
                else hipStreamWaitEvent(ctx->stream2, ev_stagger, 0);
            }
            const int rc = run(ops, tails);
            lane = 0;
            if (rc) return fail_out(rc);
        }
        if (chain) {                 // the new length is on the device only: later requests across (v,w) take it from d_len
            for (size_t i = first; i < done.size(); ++i) {
                Gene &G = genes[done[i].gene]; const int v = done[i].v, w = done[i].w;
                G.len_pending[(size_t)v * 3 + G.tree.slot(v, w)] = 1; G.len_pending[(size_t)w * 3 + G.tree.slot(w, v)] = 1;
                branch_changed(done[i].gene, v, w);
            }
        } else {
            for (size_t i = first; i < done.size(); ++i) {
                Gene &G = genes[done[i].gene]; const int v = done[i].v, w = done[i].w;
                const double nl = res(done[i].gene)[0], old = done[i].old, dl = std::fabs(nl - old);
                maxdelta[done[i].gene] = std::max(maxdelta[done[i].gene], dl);
                if (nl != old) { G.tree.set_len(v, w, nl); branch_changed(done[i].gene, v, w); }
                if (dl > thr) { std::swap(G.dirty, next[done[i].gene]); G.mark_node(v); G.mark_node(w); std::swap(G.dirty, next[done[i].gene]); }
            }
        }
    }
    host_phase_ms[HP_PASS_STEPS] += now_ms() - hp_t;
    if (chain) {
        const double t0 = now_ms();
        if (int rc = chain_sync()) { chain = false; return rc; }
        ctx->stats[K_HOST_WAIT].launches++; ctx->stats[K_HOST_WAIT].ms += now_ms() - t0;
        host_phase_ms[HP_PASS_SYNC] += now_ms() - t0; hp_t = now_ms();
        chain = false; lanes_active = false;
        bool gave_up = false;
        for (auto &d : done) gave_up = gave_up || !std::isfinite(h_chain[4 * d.idx + 1]);
        if (gave_up) {
            if (attempt == 1 || safe_now) return ctx->fail(-5, "k_newton: non-finite branch likelihood (also from the no-exchange form)");
            newton_gave_up(); ctx->newton_reissued += (long long)done.size();
            if (int rc = clear_abort()) return rc;
            for (int g = 0; g < n; ++g) if (active[g]) { std::fill(genes[g].len_pending.begin(), genes[g].len_pending.end(), 0); invalidate_all(g); }
            safe_now = true;
            continue;
        }
        for (auto &d : done) {
            Gene &G = genes[d.gene];
            const double nl = h_chain[4 * d.idx], dl = std::fabs(nl - d.old);
            det_record(det_id, G, 'S', d.v, d.w, d.old, nl, h_chain[4 * d.idx + 1]);
            maxdelta[d.gene] = std::max(maxdelta[d.gene], dl);
            G.tree.set_len(d.v, d.w, nl);
            if (dl > thr) { std::swap(G.dirty, next[d.gene]); G.mark_node(d.v); G.mark_node(d.w); std::swap(G.dirty, next[d.gene]); }
        }
        for (auto &G : genes) std::fill(G.len_pending.begin(), G.len_pending.end(), 0);
    }
    break;
    }   // attempts
    safe_now = false;
    for (int g = 0; g < n; ++g) if (active[g]) genes[g].dirty.swap(next[g]);
    host_phase_ms[HP_PASS_POST] += now_ms() - hp_t;
    return 0;
}

// FastTree's `-gamma` likelihood (FastTreeRunner.java:67-70 always passes it): the tree's per-pattern likelihoods at 20
// fixed rates, re-weighted by a discretised Gamma(alpha) whose mean is 1/mult; alpha and mult are fitted by alternating
// one-dimensional Brent searches on log alpha / log mult in [0.01, 10] (tolerance 1e-3, <= 10 rounds, stop when a round
// gains < 1e-3), the spec of oracle/pml_oracle.c po_gamma20.  Reported: Gamma20 lnL, alpha, rescale = 1/mult (FastTree
// prints the tree with lengths x rescale).  Device work: five full traversals writing the table, then one tiny k_g20
// launch per objective evaluation for all genes together.
int Batch::gamma20(std::vector<double> &lnl20, std::vector<double> &alpha20, std::vector<double> &rescale20) {
    const int n = (int)genes.size();
    HIPCHK(hipSetDevice(ctx->device));
    double rates[G20_RATES]; g20_rates(rates);
    std::vector<size_t> off(n); size_t bytes = 0;
    for (int g = 0; g < n; ++g) { off[g] = bytes; bytes += align_up((size_t)genes[g].aln.mpad * (G20_RATES * 8 + (G20_RATES / 4) * 4), 256); }
    if (int rc = ensure_tailpool(bytes)) return rc;
    auto table = [&](int g) { return reinterpret_cast<double *>(d_tailpool + off[g]); };
    auto counts = [&](int g) { return reinterpret_cast<int *>(d_tailpool + off[g] + (size_t)genes[g].aln.mpad * G20_RATES * 8); };
    // the genes carry FastTree's fixed rates only inside the loop below: whatever way it is left, they get their Gamma4 rates back
    struct RatesBack { Batch *b; ~RatesBack() { for (int g = 0; g < (int)b->genes.size(); ++g) b->set_alpha(g, b->genes[g].alpha); } };
    for (int j = 0; j < G20_RATES / 4; ++j) {
        RatesBack back{this};
        std::vector<PendingOp> ops; std::vector<Tail> tails;
        for (int g = 0; g < n; ++g) {
            Gene &G = genes[g];
            for (int c = 0; c < 4; ++c) G.rates[c] = rates[4 * j + c];
            ++G.rates_epoch; invalidate_all(g);
            const int r = G.tree.nbr[0][0];
            need(g, r, 0, ops);
            Tail t{g, msg(g, 0, r), msg(g, r, 0), MODE_EVALUATE_CAT, G.tree.len[0][0], 0, 0, -1, 0, 0};
            t.patlnl_dev = table(g) + (size_t)4 * j * G.aln.mpad; t.scl_dev = counts(g) + (size_t)j * G.aln.mpad;
            tails.push_back(t);
        }
        if (int rc = run(ops, tails)) return rc;
    }
    if (int rc = ensure_results((size_t)n)) return rc;
    if (int rc = ensure_stage((size_t)n * sizeof(G20Req))) return rc;
    std::vector<double> la(n, 0.0), lm(n, 0.0), f(n, 0.0);               // log alpha, log mult, -lnL
    auto objective = [&](const std::vector<char> &act) -> int {
        G20Req *h = (G20Req *)h_stage; int k = 0;
        std::vector<int> who;
        for (int g = 0; g < n; ++g) {
            if (!act[g]) continue;
            G20Req &r = h[k++];
            r.table = table(g); r.cnt = counts(g); r.weight = genes[g].d_weight; r.out = d_chain + 4 * g; r.patlnl = nullptr;
            r.mpad = genes[g].aln.mpad; r.pad = 0;
            g20_weights(std::exp(la[g]), std::exp(lm[g]), r.w);
            who.push_back(g);
        }
        if (!k) return 0;
        HIPCHK(hipMemcpyAsync(d_stage, h_stage, (size_t)k * sizeof(G20Req), hipMemcpyHostToDevice, ctx->stream));
        launch_g20((const G20Req *)d_stage, k, ctx->stream);
        results_used = (size_t)n;
        if (int rc = fetch_results(true)) return rc;
        { if (int rc_ = ctx->sync(ctx->stream)) return rc_; }
        HIPCHK(hipGetLastError());
        for (int g : who) { f[g] = -h_chain[4 * g]; if (!std::isfinite(f[g])) return ctx->fail(-5, "device returned a non-finite Gamma20 likelihood"); }
        return 0;
    };
    std::vector<char> all(n, 1), active(n, 1);
    if (int rc = objective(all)) return rc;
    const double LO = std::log(0.01), HI = std::log(10.0), TOL = 1e-3;
    for (int round = 0; round < 10; ++round) {
        bool any = false; for (char a : active) any |= a;
        if (!any) break;
        const std::vector<double> start(f);
        for (int which = 0; which < 2; ++which) {                          // 0: alpha, 1: mult
            std::vector<double> &x = which ? lm : la;
            std::vector<Brent> br(n);
            std::vector<char> act(active);
            for (int g = 0; g < n; ++g) if (act[g]) br[g].start(LO, HI, x[g], f[g], TOL);
            for (;;) {
                bool moved = false;
                for (int g = 0; g < n; ++g) { if (!act[g]) continue; if (br[g].propose()) { x[g] = br[g].u; moved = true; } else act[g] = 0; }
                if (!moved) break;
                if (int rc = objective(act)) return rc;
                for (int g = 0; g < n; ++g) if (act[g]) br[g].update(f[g]);
            }
            for (int g = 0; g < n; ++g) if (active[g]) { x[g] = br[g].x; f[g] = br[g].fx; }
        }
        for (int g = 0; g < n; ++g) if (active[g] && !(f[g] < start[g] - 1e-3)) active[g] = 0;
    }
    lnl20.resize(n); alpha20.resize(n); rescale20.resize(n);
    for (int g = 0; g < n; ++g) { lnl20[g] = -f[g]; alpha20[g] = std::exp(la[g]); rescale20[g] = std::exp(-lm[g]); }
    return 0;
}

// Brent on log(alpha) per gene, all genes in lock step (one full-traversal evaluation of every active gene per
// iteration).  Control flow = the oracle's eng_opt_alpha: +-ln 4 window around the current value, doubled and
// continued when the minimum ends at a window edge that is not a global limit.
int Batch::opt_alpha(const std::vector<char> &active, double *lnl, double tol) {
    const int n = (int)genes.size();
    const double LMIN = std::log(ALPHA_MIN), LMAX = std::log(ALPHA_MAX);
    std::vector<Brent> br(n);
    std::vector<char> act(active);
    std::vector<double> f(n), W(n, std::log(4.0)), lo(n), hi(n);
    std::vector<int> win(n, 0);
    if (int rc = score(act, f.data())) return rc;
    auto open_window = [&](int g, double x, double fx) {
        lo[g] = std::max(LMIN, x - W[g]); hi[g] = std::min(LMAX, x + W[g]);
        br[g].start(lo[g], hi[g], x, fx, tol);
    };
    for (int g = 0; g < n; ++g) if (act[g]) open_window(g, std::log(genes[g].alpha), -f[g]);
    for (;;) {
        bool any = false;
        const double hp_a = now_ms();
        for (int g = 0; g < n; ++g) {
            if (!act[g]) continue;
            for (;;) {
                if (br[g].propose()) { set_alpha(g, std::exp(br[g].u)); any = true; break; }
                const double x = br[g].x, edge = 4 * tol;
                if (++win[g] < 8 && ((x - lo[g] < edge && lo[g] > LMIN) || (hi[g] - x < edge && hi[g] < LMAX))) { W[g] *= 2; open_window(g, x, br[g].fx); continue; }
                act[g] = 0; break;
            }
        }
        host_phase_ms[HP_ALPHA_HOST] += now_ms() - hp_a;
        if (!any) break;
        if (int rc = score(act, f.data())) return rc;
        for (int g = 0; g < n; ++g) if (act[g]) br[g].update(-f[g]);
    }
    for (int g = 0; g < n; ++g) if (active[g]) { set_alpha(g, std::exp(br[g].x)); lnl[g] = -br[g].fx; }
    return 0;
}

int Batch::optimize(bool opt_alpha_flag, double eps, double *lnl, const std::vector<char> *mask) {
    const int n = (int)genes.size();
    std::vector<char> active(n, 1);
    if (mask) active = *mask;
    for (int g = 0; g < n; ++g) {
        if (!active[g]) continue;
        Tree &T = genes[g].tree;
        for (auto &l : T.len) for (double &x : l) if (x < TMIN) x = TMIN;
        invalidate_all(g);
    }
    const std::vector<char> initial(active);
    std::vector<double> cur(n), nl(n), md;
    if (int rc = evaluate(active, cur.data())) return rc;
    // precision follows eps (oracle: po_engine_optimize): passes stop when max |dt| < thr =
    // clamp(eps/100, 1e-6, 1e-3), Newton stops at thr/100; <= 8 passes per round for eps >= 0.05,
    // <= 16 otherwise; every round starts with all branches dirty
    const int maxpass = eps >= 0.05 ? 8 : 16;
    const double thr = std::min(1e-3, std::max(1e-6, eps * 0.01)), save_tol = newton_tol;
    newton_tol = thr * 0.01;
    struct Restore { double &r; double v; ~Restore() { r = v; } } restore{newton_tol, save_tol};
    for (int round = 0; round < 100; ++round) {
        bool any = false; for (char a : active) any |= a;
        if (!any) break;
        std::vector<char> sm(active);
        for (int g = 0; g < n; ++g) if (sm[g]) genes[g].mark_all();
        // geometric pass budget 1, 2, 4, ... maxpass: while alpha is still moving a lot, branch lengths
        // are not polished to thr (they shift again with the next alpha)
        const int budget = opt_alpha_flag ? std::min(maxpass, 1 << std::min(round, 5)) : maxpass;
        for (int pass = 0; pass < budget; ++pass) {
            bool anys = false; for (char a : sm) anys |= a;
            if (!anys) break;
            if (int rc = smooth_pass(sm, md, thr)) return rc;
            for (int g = 0; g < n; ++g) if (sm[g] && md[g] < thr) sm[g] = 0;
        }
        if (opt_alpha_flag) { if (int rc = opt_alpha(active, nl.data(), eps >= 0.05 ? 1e-2 : 1e-4)) return rc; }
        else { if (int rc = evaluate(active, nl.data())) return rc; }
        for (int g = 0; g < n; ++g) {
            if (!active[g]) continue;
            const double gain = nl[g] - cur[g]; cur[g] = nl[g];
            if (gain < eps) active[g] = 0;
        }
    }
    for (int g = 0; g < n; ++g) if (initial[g]) lnl[g] = cur[g];
    if (std::getenv("PML_TRACE")) fprintf(stderr, "[pml] optimize(eps %g): cumulative passes %ld smooth-steps %ld evals %ld\n", eps, cnt_passes, cnt_smooth, cnt_eval);
    return 0;
}

}  // namespace pml
