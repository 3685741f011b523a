#include <chrono>
// search.cpp -- topology search driver (NNI hill climbing) on top of the batch engine.
//
// Replaces the tree search inside the external programs PEPR spawns (FastTree's ML-NNI rounds,
// FastTreeRunner.java:67-94 / SURVEY 3.3; RAxML's hill climbing, RAxMLRunner.java:115-147).
// Control flow is specified once (DESIGN.md "Search") and implemented twice: here, batched over
// genes with every likelihood evaluated by the HIP kernels, and in oracle/pml_oracle.c
// (nni_round / po_engine_search) one gene at a time on the CPU.
//
// One NNI round, per gene: for every internal edge (u,v), u<v, in node order: Newton-optimise
// the central branch for the current arrangement and for the two alternatives (the alternatives'
// end CLVs are built into two scratch slots from the four cached neighbour messages); keep the
// better alternative if it gains > 0.01 lnL; sort candidates by gain, apply those that share no
// node, re-optimise lightly (<= 2 smoothing passes), fall back to the single best move if the
// combination did not improve, and to no move if that fails too.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#include "engine.hpp"

#define HIPCHK(expr)                                                                          \
    do { hipError_t e_ = (expr);                                                              \
         if (e_ != hipSuccess) return ctx->fail(-4, std::string("HIP: ") + hipGetErrorString(e_)); } while (0)

namespace pml {
static double hp_now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

namespace {
constexpr double NNI_MIN_GAIN = 0.01;
struct Cand { int u, v, alt; double gain, t; int order; };

void others(const Tree &T, int v, int excl, int out[2], double len[2]) {
    int ci = 0;
    for (int q = 0; q < 3; ++q) if (T.nbr[v][q] != excl) { out[ci] = T.nbr[v][q]; len[ci] = T.len[v][q]; ci++; }
}
// swap subtree x (neighbour of u) with subtree y (neighbour of v); slots keep their position
void tree_swap(Tree &T, int u, int x, int v, int y) {
    const int ku = T.slot(u, x), kv = T.slot(v, y), kx = T.slot(x, u), ky = T.slot(y, v);
    const double lx = T.len[u][ku], ly = T.len[v][kv];
    T.nbr[u][ku] = y; T.len[u][ku] = ly; T.nbr[v][kv] = x; T.len[v][kv] = lx;
    T.nbr[x][kx] = v; T.nbr[y][ky] = u;
}
void nni_apply(Tree &T, int u, int v, int alt, double tnew) {
    int a[2], c[2]; double la[2], lc[2];
    others(T, u, v, a, la); others(T, v, u, c, lc);
    tree_swap(T, u, a[1], v, alt == 1 ? c[0] : c[1]);
    T.set_len(u, v, tnew);
}
}  // namespace

// FastTree -constraints semantics (FastTreeRunner.java:54-64, 243-273): every 0/1 column is a split
// the result must display; taxa the matrix does not name, or marks '-', are free in that column
int Batch::set_constraints(int ncons, int ntax, const char *const *names, const char *const *rows) {
    for (Gene &G : genes) G.cons.clear();
    if (ncons <= 0 || ntax <= 0) return 0;
    if (!names || !rows) return ctx->fail(-1, "constraint matrix missing");
    for (int i = 0; i < ntax; ++i) if (!names[i] || !rows[i] || (int)strnlen(rows[i], (size_t)ncons) < ncons) return ctx->fail(-1, "constraint row shorter than nconstraints");
    for (Gene &G : genes) {
        const int n = G.aln.ntax, words = (n + 63) / 64;
        std::vector<int> map(ntax, -1);
        for (int i = 0; i < ntax; ++i) for (int t = 0; t < n; ++t) if (G.aln.names[t] == names[i]) { map[i] = t; break; }
        for (int c = 0; c < ncons; ++c) {
            Constraint K; K.one.assign(words, 0); K.zero.assign(words, 0);
            int n1 = 0, n0 = 0;
            for (int i = 0; i < ntax; ++i) {
                const int t = map[i]; if (t < 0) continue;
                if (rows[i][c] == '1') { K.one[t >> 6] |= 1ULL << (t & 63); ++n1; }
                else if (rows[i][c] == '0') { K.zero[t >> 6] |= 1ULL << (t & 63); ++n0; }
            }
            if (n1 >= 2 && n0 >= 2) G.cons.push_back(K);        // smaller sides are trivially displayed
        }
    }
    return 0;
}

// <= 2 smoothing passes (stop when max |dt| < 1e-3), then lnL
int Batch::light_smooth(const std::vector<char> &active, double *lnl) {
    std::vector<char> sm(active);
    std::vector<double> md;
    for (int pass = 0; pass < 2; ++pass) {
        bool any = false; for (char a : sm) any |= a;
        if (!any) break;
        if (int rc = smooth_pass(sm, md, 1e-3)) return rc;              // dirty = around the moves
        for (size_t g = 0; g < sm.size(); ++g) if (sm[g] && md[g] < 1e-3) sm[g] = 0;
    }
    return evaluate(active, lnl);
}

// SH-like local supports: for every internal edge the current arrangement and its two NNI alternatives are scored as
// in nni_round (central branch Newton-optimised, nothing else re-optimised -- FastTree does not re-optimise for the
// resamples either), k_newton also leaves the per-pattern lnL of each arrangement in d_patlnl[0..2], and k_sh resamples
// the alignment columns nboot times on the device.  One run() + one k_sh launch per edge for the whole batch.
int Batch::sh_support(int nboot, unsigned long long seed, std::vector<std::vector<double>> &support, std::vector<std::vector<std::pair<int, int>>> &edges) {
    const int n = (int)genes.size();
    HIPCHK(hipSetDevice(ctx->device));
    if (!d_site2pat) {
        size_t tot = 0; site2pat_off.assign(n, 0);
        for (int g = 0; g < n; ++g) {
            if ((int)genes[g].aln.site2pat.size() != genes[g].aln.nsites) return ctx->fail(-1, "SH-like supports need the site map (not available for device-gathered replicates)");
            site2pat_off[g] = tot; tot += (size_t)genes[g].aln.nsites;
        }
        HIPCHK(hipMalloc((void **)&d_site2pat, std::max<size_t>(tot, 1) * sizeof(int)));
        for (int g = 0; g < n; ++g) HIPCHK(hipMemcpy(d_site2pat + site2pat_off[g], genes[g].aln.site2pat.data(), (size_t)genes[g].aln.nsites * sizeof(int), hipMemcpyHostToDevice));
    }
    std::vector<double> l0(n);
    if (int rc = evaluate(std::vector<char>(n, 1), l0.data())) return rc;       // every cached CLV valid
    edges.assign(n, {}); support.assign(n, {});
    size_t maxsteps = 0;
    for (int g = 0; g < n; ++g) {
        const Tree &T = genes[g].tree; const int nt = T.ntax;
        for (int u = nt; u < T.nnodes(); ++u) for (int k = 0; k < 3; ++k) { const int v = T.nbr[u][k]; if (v < nt || v < u) continue; edges[g].push_back({u, v}); }
        support[g].assign(edges[g].size(), 0.0);
        maxsteps = std::max(maxsteps, edges[g].size());
    }
    ShReq *d_req = nullptr; double *d_out = nullptr;
    HIPCHK(hipMalloc((void **)&d_req, sizeof(ShReq) * n));
    if (hipMalloc((void **)&d_out, sizeof(double) * n) != hipSuccess) { hipFree(d_req); return ctx->fail(-4, "allocation failed"); }
    struct Drop { ShReq *a; double *b; ~Drop() { hipFree(a); hipFree(b); } } drop{d_req, d_out};
    std::vector<ShReq> hreq(n); std::vector<double> hout(n); std::vector<int> who;
    for (size_t step = 0; step < maxsteps; ++step) {
        std::vector<PendingOp> ops; std::vector<Tail> tails;
        who.clear();
        for (int g = 0; g < n; ++g) {
            if (step >= edges[g].size()) continue;
            Gene &G = genes[g]; const Tree &T = G.tree;
            auto [u, v] = edges[g][step];
            const double t0 = T.len[u][T.slot(u, v)];
            int a[2], c[2]; double la[2], lc[2];
            others(T, u, v, a, la); others(T, v, u, c, lc);
            need(g, u, v, ops); need(g, v, u, ops);
            need(g, a[0], u, ops); need(g, a[1], u, ops); need(g, c[0], v, ops); need(g, c[1], v, ops);
            Tail t0t{g, msg(g, u, v), msg(g, v, u), MODE_SUMTABLE, t0, 32, 0, -1}; t0t.patlnl_dev = G.d_patlnl[0];
            tails.push_back(t0t);
            for (int alt = 1; alt <= 2; ++alt) {
                const int y = (alt == 1) ? 0 : 1, sx = 2 * alt - 2, sy = 2 * alt - 1;
                PendingOp X; X.gene = g; X.out_kind = SIDE_SCRATCH; X.out_id = sx; X.level = 0;
                X.child[0] = msg(g, a[0], u); X.t[0] = la[0]; X.child[1] = msg(g, c[y], v); X.t[1] = lc[y];
                PendingOp Y; Y.gene = g; Y.out_kind = SIDE_SCRATCH; Y.out_id = sy; Y.level = 0;
                Y.child[0] = msg(g, a[1], u); Y.t[0] = la[1]; Y.child[1] = msg(g, c[1 - y], v); Y.t[1] = lc[1 - y];
                ops.push_back(X); ops.push_back(Y);
                Tail ta{g, {SIDE_SCRATCH, sx}, {SIDE_SCRATCH, sy}, MODE_SUMTABLE, t0, 32, alt, -1}; ta.patlnl_dev = G.d_patlnl[alt];
                tails.push_back(ta);
            }
            hreq[who.size()] = ShReq{G.d_patlnl[0], G.d_patlnl[1], G.d_patlnl[2], d_site2pat + site2pat_off[g], d_out + who.size(), seed, G.aln.nsites, nboot};
            who.push_back(g);
        }
        if (int rc = run(ops, tails)) return rc;
        HIPCHK(hipMemcpyAsync(d_req, hreq.data(), sizeof(ShReq) * who.size(), hipMemcpyHostToDevice, ctx->stream));
        launch_sh(d_req, (int)who.size(), ctx->stream);
        HIPCHK(hipMemcpyAsync(hout.data(), d_out, sizeof(double) * who.size(), hipMemcpyDeviceToHost, ctx->stream));
        { if (int rc_ = ctx->sync(ctx->stream)) return rc_; }
        for (size_t i = 0; i < who.size(); ++i) support[who[i]][step] = hout[i];
    }
    return 0;
}

int Batch::nni_round(const std::vector<char> &active, std::vector<double> &lnl, std::vector<int> &applied) {
    const int n = (int)genes.size();
    ++topo_epoch;
    applied.assign(n, 0);
    double hp_t = hp_now();
    // per-gene edge lists in oracle order
    std::vector<std::vector<std::pair<int, int>>> edges(n);
    size_t maxsteps = 0;
    for (int g = 0; g < n; ++g) {
        if (!active[g]) continue;
        const Tree &T = genes[g].tree; const int nt = T.ntax;
        for (int u = nt; u < T.nnodes(); ++u) for (int k = 0; k < 3; ++k) { const int v = T.nbr[u][k]; if (v < nt || v < u) continue; edges[g].push_back({u, v}); }
        maxsteps = std::max(maxsteps, edges[g].size() * 3);
    }
    std::vector<std::vector<double>> L(n), Tn(n);
    for (int g = 0; g < n; ++g) { L[g].assign(edges[g].size() * 3, 0.0); Tn[g].assign(edges[g].size() * 3, 0.0); }
    // ALL internal edges of the round go into one run() (round 1 issued one launch set per edge index): every edge is
    // scored against the same tree, so the cached directed messages are shared; per edge the current arrangement and
    // both alternatives are three Newton tails with pooled sumtables; alternative k builds its end CLVs into scratch
    // 2k-2, 2k-1 -- a workgroup walks its gene's operations in order and a tail sits right behind the operations it
    // needs, so the four scratch slots are reused from edge to edge.  The round is cut into chunks of edges only when the
    // pooled sumtables (3 x 640 B per pattern and edge) would not fit in free HBM.
    std::vector<size_t> tb(n, 0);                 // bytes of one pooled sumtable (+ scaling counts) of gene g
    size_t per_edge = 0, need_all = 0, nres_all = 0;
    for (int g = 0; g < n; ++g) {
        if (edges[g].empty()) continue;
        tb[g] = (clv_doubles(genes[g].aln.mpad) * 8 + (size_t)genes[g].aln.mpad * 4 + 255) / 256 * 256;
        per_edge += 3 * tb[g]; need_all += 3 * tb[g] * edges[g].size(); nres_all += 3 * edges[g].size();
    }
    size_t nedge_max = maxsteps / 3, chunk = nedge_max;
    if (nedge_max > 0) {
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipSetDevice(ctx->device));
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = 0;
        size_t budget = std::max(tailpool_cap, (size_t)(0.6 * (double)(free_b / (size_t)std::max(share, 1) + tailpool_cap)));
        if (const char *e = std::getenv("PML_NNI_POOL_MB")) budget = (size_t)std::atoll(e) << 20;      // test hook
        if (need_all > budget) chunk = std::max<size_t>(1, budget / std::max<size_t>(per_edge, 1));
        if (int rc = ensure_tailpool(std::min(need_all, chunk * per_edge))) return rc;
        if (int rc = ensure_results(nres_all)) return rc;
    }
    // The edges of a gene are dealt over NNI_PARTS parts (PendingOp::part): every part is a run of its own in the launch, with its
    // own four scratch CLVs, so a gene's 3 x 47 Newton tails (C3) are worked on by several gangs of workgroups side by side instead
    // of one after the other by one -- the round was latency-bound (21 ms for 128 C3 genes whose MFMA work is 3-4 ms).  The
    // messages the edges read are brought up to date by a launch of their own first: parts must not depend on each other.
    // PML_NNI_PARTS=1 is the A-B arm (one run per gene).
    static const int parts_env = std::getenv("PML_NNI_PARTS") ? std::max(1, std::min(NNI_PARTS, std::atoi(std::getenv("PML_NNI_PARTS")))) : NNI_PARTS;
    for (size_t e0 = 0; e0 < nedge_max; e0 += chunk) {
        ++cnt_nni;
        std::vector<PendingOp> ops; std::vector<Tail> tails;
        for (int g = 0; g < n; ++g) {              // every cached message the chunk's edges read
            if (!active[g]) continue;
            const Tree &T = genes[g].tree;
            for (size_t e = e0; e < std::min(edges[g].size(), e0 + chunk); ++e) {
                auto [u, v] = edges[g][e];
                int a[2], c[2]; double la[2], lc[2];
                others(T, u, v, a, la); others(T, v, u, c, lc);
                need(g, u, v, ops); need(g, v, u, ops);
                need(g, a[0], u, ops); need(g, a[1], u, ops); need(g, c[0], v, ops); need(g, c[1], v, ops);
            }
        }
        if (!ops.empty()) {
            std::vector<Tail> none;
            if (int rc = run(ops, none)) return rc;
            ops.clear();
        }
        size_t pool_off = 0, ri = 0;
        std::vector<size_t> rbase(n, 0);
        for (int g = 0; g < n; ++g) {
            if (!active[g] || e0 >= edges[g].size()) continue;
            const Tree &T = genes[g].tree;
            int count[NNI_PARTS] = {0};          // operations emitted so far, per part
            rbase[g] = ri;
            for (size_t e = e0; e < std::min(edges[g].size(), e0 + chunk); ++e) {
                auto [u, v] = edges[g][e];
                const int part = (int)((e - e0) % (size_t)parts_env), sbase = 4 * part;
                const double t0 = T.len[u][T.slot(u, v)];
                int a[2], c[2]; double la[2], lc[2];
                others(T, u, v, a, la); others(T, v, u, c, lc);
                auto pooled = [&](Tail t) {
                    t.sumtab_dev = reinterpret_cast<double *>(d_tailpool + pool_off); pool_off += tb[g];
                    t.result_dev = d_chain + 4 * ri; t.result_host = h_chain + 4 * ri; ++ri;
                    t.part = part;
                    tails.push_back(t);
                };
                pooled({g, msg(g, u, v), msg(g, v, u), MODE_SUMTABLE, t0, 32, 0, count[part]});
                for (int alt = 1; alt <= 2; ++alt) {
                    const int y = (alt == 1) ? 0 : 1, sx = sbase + 2 * alt - 2, sy = sbase + 2 * alt - 1;
                    PendingOp X; X.gene = g; X.part = part; X.out_kind = SIDE_SCRATCH; X.out_id = sx; X.level = 0;
                    X.child[0] = msg(g, a[0], u); X.t[0] = la[0]; X.bv[0] = u; X.bq[0] = T.slot(u, a[0]);
                    X.child[1] = msg(g, c[y], v); X.t[1] = lc[y]; X.bv[1] = v; X.bq[1] = T.slot(v, c[y]);
                    PendingOp Y; Y.gene = g; Y.part = part; Y.out_kind = SIDE_SCRATCH; Y.out_id = sy; Y.level = 0;
                    Y.child[0] = msg(g, a[1], u); Y.t[0] = la[1]; Y.bv[0] = u; Y.bq[0] = T.slot(u, a[1]);
                    Y.child[1] = msg(g, c[1 - y], v); Y.t[1] = lc[1 - y]; Y.bv[1] = v; Y.bq[1] = T.slot(v, c[1 - y]);
                    ops.push_back(X); ops.push_back(Y); count[part] += 2;
                    pooled({g, {SIDE_SCRATCH, sx}, {SIDE_SCRATCH, sy}, MODE_SUMTABLE, t0, 32, 0, count[part]});
                }
            }
        }
        host_phase_ms[HP_NNI_BUILD] += hp_now() - hp_t; hp_t = hp_now();
        if (int rc = run(ops, tails)) return rc;
        host_phase_ms[HP_NNI_RUN] += hp_now() - hp_t; hp_t = hp_now();
        for (int g = 0; g < n; ++g) {
            if (!active[g] || e0 >= edges[g].size()) continue;
            size_t k = rbase[g];
            for (size_t e = e0; e < std::min(edges[g].size(), e0 + chunk); ++e)
                for (int q = 0; q < 3; ++q, ++k) { Tn[g][3 * e + q] = h_chain[4 * k]; L[g][3 * e + q] = h_chain[4 * k + 1]; det_record(det_id, genes[g], 'N', (int)e, q, h_chain[4 * k], h_chain[4 * k + 1], 0); }
        }
    }
    // candidate selection and application
    std::vector<std::vector<Cand>> cands(n);
    std::vector<Tree> backup(n);
    std::vector<double> lnl0(lnl);
    std::vector<char> stageA(n, 0);
    for (int g = 0; g < n; ++g) {
        if (!active[g]) continue;
        std::vector<std::vector<uint64_t>> leafs;
        if (!genes[g].cons.empty()) leafs = leaf_sets(genes[g].tree);
        for (size_t e = 0; e < edges[g].size(); ++e) {
            const double Lc = L[g][3 * e];
            double L1 = L[g][3 * e + 1], L2 = L[g][3 * e + 2];
            if (!genes[g].cons.empty()) {          // an alternative whose new split violates a constraint is not a candidate
                const Tree &T = genes[g].tree; const int nt = T.ntax;
                const int u = edges[g][e].first, v = edges[g][e].second;
                int a[2], c[2]; double la[2], lc[2];
                others(T, u, v, a, la); others(T, v, u, c, lc);
                auto setof = [&](int node, int toward) {
                    std::vector<uint64_t> S((nt + 63) / 64, 0);
                    if (node < nt) S[node >> 6] |= 1ULL << (node & 63); else S = leafs[(node - nt) * 3 + T.slot(node, toward)];
                    return S;
                };
                for (int alt = 1; alt <= 2; ++alt) {
                    std::vector<uint64_t> X = setof(a[0], u); const std::vector<uint64_t> Y = setof(c[alt - 1], v);
                    for (size_t w = 0; w < X.size(); ++w) X[w] |= Y[w];
                    if (!compatible_with_all(genes[g].cons, X)) (alt == 1 ? L1 : L2) = -1e300;
                }
            }
            const int best = (L2 > L1) ? 2 : 1;
            const double gain = (best == 1 ? L1 : L2) - Lc;
            if (gain > NNI_MIN_GAIN) cands[g].push_back({edges[g][e].first, edges[g][e].second, best, gain, Tn[g][3 * e + best], (int)cands[g].size()});
        }
        if (cands[g].empty()) continue;
        std::sort(cands[g].begin(), cands[g].end(), [](const Cand &a, const Cand &b) { return a.gain != b.gain ? a.gain > b.gain : a.order < b.order; });
        Tree &T = genes[g].tree;
        backup[g] = T;
        std::vector<char> used(T.nnodes(), 0);
        genes[g].mark_none();
        for (auto &c : cands[g]) {
            if (used[c.u] || used[c.v]) continue;
            used[c.u] = used[c.v] = 1;
            nni_apply(T, c.u, c.v, c.alt, c.t); applied[g]++;
            genes[g].mark_node(c.u); genes[g].mark_node(c.v);      // the five branches of the quartet
        }
        invalidate_all(g); stageA[g] = 1;
    }
    host_phase_ms[HP_NNI_SELECT] += hp_now() - hp_t;
    std::vector<double> l1(n, 0.0);
    bool any = false; for (char a : stageA) any |= a;
    if (!any) return 0;
    if (int rc = light_smooth(stageA, l1.data())) return rc;
    std::vector<char> stageB(n, 0); any = false;
    for (int g = 0; g < n; ++g) {
        if (!stageA[g]) continue;
        if (l1[g] > lnl0[g] + 1e-6) { lnl[g] = l1[g]; continue; }
        genes[g].tree = backup[g]; invalidate_all(g);
        const Cand &c = cands[g][0];
        genes[g].mark_none();
        nni_apply(genes[g].tree, c.u, c.v, c.alt, c.t); applied[g] = 1;
        genes[g].mark_node(c.u); genes[g].mark_node(c.v);
        invalidate_all(g); stageB[g] = 1; any = true;
    }
    if (any) {
        if (int rc = light_smooth(stageB, l1.data())) return rc;
        for (int g = 0; g < n; ++g) {
            if (!stageB[g]) continue;
            if (l1[g] > lnl0[g] + 1e-6) { lnl[g] = l1[g]; continue; }
            genes[g].tree = backup[g]; invalidate_all(g); applied[g] = 0; lnl[g] = lnl0[g];
        }
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// lazy SPR round (oracle: spr_round / spr_explore / spr_score).  Per gene a small state machine
// walks the prunes (node p, neighbour s) in order; every candidate regraft edge is one unit:
// [path message of the pruned tree for this depth] + insertion CLV + evaluate across the pendant
// branch.  Genes advance in lockstep, each on its own tree.
//
// Batching.  One run() carries, per gene, the candidates of SEVERAL consecutive prunes (about
// SPR_UNIT_BUDGET units): all prunes of a batch are scored against the same tree, i.e. under the
// assumption that none of the earlier ones moves anything -- almost always true after the NNI rounds.
// The results are then read in prune order: the first prune whose best candidate beats the current
// likelihood is applied exactly as the sequential procedure applies it (four branch Newtons, keep
// only if the tree really improved); if the move is kept, the scores of the later prunes of the
// batch were computed on a tree that no longer exists and are discarded -- scanning resumes behind
// that prune; if it is rejected the tree is restored and the later scores still stand.  Every
// decision is the sequential procedure's decision; only the launch granularity changed (round 1:
// at most four candidates of one prune per launch, 7898 launches for a C4 shard).
// ------------------------------------------------------------------------------------------
namespace {
constexpr double SPR_MIN_GAIN = 0.01;
constexpr int SPR_MAX_RADIUS = 6;
constexpr int SPR_INS_SLOT = 7;
constexpr size_t SPR_UNIT_BUDGET = 96;
struct PathOp { int depth; Side left; double tl; int lbv, lbq; Side right; double tr; int rbv, rbq; };
struct Unit { std::vector<PathOp> paths; int g, h, mslot; };   // score candidate edge (g,h) with path slot mslot
struct Prune { int p, ks, s, x, y; double tx, ty, ts; std::vector<Unit> units; size_t rbase; };
struct SprState {
    int p = 0, ks = 0; bool done = false;        // cursor: next prune to build
    std::vector<Prune> batch; size_t bi = 0;      // scored prunes waiting to be read, bi = next to read
    int phase = -1;                               // -1: needs scores; 0: scores in flight; 1..5: applying batch[bi]
    double best = -1e300; int bg = -1, bh = -1;
    Tree backup;
};
void spr_apply(Tree &T, int p, int x, int y, int g, int h) {
    const int kx = T.slot(p, x), ky = T.slot(p, y);
    const double tx = T.len[p][kx], ty = T.len[p][ky], tgh = T.len[g][T.slot(g, h)];
    const int sx = T.slot(x, p), sy = T.slot(y, p);
    T.nbr[x][sx] = y; T.len[x][sx] = tx + ty; T.nbr[y][sy] = x; T.len[y][sy] = tx + ty;
    const int sg = T.slot(g, h), sh = T.slot(h, g);
    T.nbr[g][sg] = p; T.len[g][sg] = 0.5 * tgh; T.nbr[h][sh] = p; T.len[h][sh] = 0.5 * tgh;
    T.nbr[p][kx] = g; T.len[p][kx] = 0.5 * tgh; T.nbr[p][ky] = h; T.len[p][ky] = 0.5 * tgh;
}
}  // namespace

int Batch::spr_round(const std::vector<char> &active, int radius, std::vector<double> &lnl, std::vector<int> &moves) {
    const int n = (int)genes.size();
    ++topo_epoch;
    moves.assign(n, 0);
    radius = std::min(radius, SPR_MAX_RADIUS);
    std::vector<SprState> st(n);
    std::vector<std::vector<double>> scores_of(n);      // per gene: candidate scores of its current batch (Prune::rbase indexes it)
    for (int g = 0; g < n; ++g) { st[g].done = !active[g] || genes[g].aln.ntax < 5; st[g].p = genes[g].aln.ntax; st[g].ks = 0; st[g].phase = -1; }

    // the unit list of the prune (p, ks) of gene g (DFS order of the oracle's recursion)
    auto build_prune = [&](int g, int p, int ks, Prune &P) {
        const Tree &T = genes[g].tree; const int nt = T.ntax;
        P.p = p; P.ks = ks; P.s = T.nbr[p][ks];
        int xy[2]; double lxy[2]; others(T, p, P.s, xy, lxy);
        P.x = xy[0]; P.y = xy[1]; P.tx = lxy[0]; P.ty = lxy[1]; P.ts = T.len[p][ks];
        P.units.clear();
        std::vector<PathOp> pending;
        // constraints: regrafting beyond edge (gg,h) turns its split into L(h side) + L(S); if that is
        // incompatible, neither this edge nor anything behind it is a candidate
        const std::vector<Constraint> &cons = genes[g].cons;
        std::vector<std::vector<uint64_t>> leafs; std::vector<uint64_t> LS;
        if (!cons.empty()) {
            leafs = leaf_sets(T);
            LS.assign((nt + 63) / 64, 0);
            if (P.s < nt) LS[P.s >> 6] |= 1ULL << (P.s & 63); else LS = leafs[(P.s - nt) * 3 + T.slot(P.s, p)];
        }
        auto allowed = [&](int gg, int h) {
            if (cons.empty()) return true;
            std::vector<uint64_t> X((nt + 63) / 64, 0);
            if (h < nt) X[h >> 6] |= 1ULL << (h & 63); else X = leafs[(h - nt) * 3 + T.slot(h, gg)];
            for (size_t w = 0; w < X.size(); ++w) X[w] |= LS[w];
            return compatible_with_all(cons, X);
        };
        std::function<void(int, int, int)> explore = [&](int gg, int h, int depth) {
            if (!allowed(gg, h)) { pending.clear(); return; }
            Unit u; u.paths = pending; pending.clear(); u.g = gg; u.h = h; u.mslot = depth - 1;
            P.units.push_back(u);
            if (h < nt || depth >= radius) return;
            int ch[2]; double lc[2]; others(T, h, gg, ch, lc);
            const double tgh = T.len[gg][T.slot(gg, h)];
            for (int i = 0; i < 2; ++i) {
                pending.push_back({depth, {SIDE_SCRATCH, depth - 1}, tgh, gg, T.slot(gg, h), msg(g, ch[1 - i], h), lc[1 - i], h, T.slot(h, ch[1 - i])});
                explore(h, ch[i], depth + 1);
            }
        };
        for (int sidei = 0; sidei < 2; ++sidei) {
            const int a = sidei == 0 ? P.x : P.y, b = sidei == 0 ? P.y : P.x;
            if (a < nt) continue;
            int ch[2]; double lc[2]; others(T, a, p, ch, lc);
            for (int i = 0; i < 2; ++i) {
                pending.push_back({0, msg(g, b, p), P.tx + P.ty, -1, 0, msg(g, ch[1 - i], a), lc[1 - i], a, T.slot(a, ch[1 - i])});
                explore(a, ch[i], 1);
            }
        }
    };
    auto cursor_next = [&](int g) {
        SprState &S = st[g];
        if (++S.ks == 3) { S.ks = 0; ++S.p; }
    };
    auto need_side = [&](int g, const Side &sd, std::vector<PendingOp> &ops) {
        if (sd.kind != SIDE_MSG) return;
        const Gene &G = genes[g];
        const int v = G.aln.ntax + sd.id / 3, to = G.tree.nbr[v][sd.id % 3];
        need(g, v, to, ops);
    };
    // which branch the gene optimises in apply phase ph of prune P (oracle order)
    auto apply_edge = [&](const SprState &S, int ph, int &u, int &v) {
        const Prune &P = S.batch[S.bi];
        if (ph == 1) { u = P.p; v = P.s; } else if (ph == 2) { u = P.p; v = S.bg; }
        else if (ph == 3) { u = P.p; v = S.bh; } else { u = P.x; v = P.y; }
    };

    for (;;) {
        std::vector<PendingOp> ops; std::vector<Tail> tails;
        std::vector<int> kind(n, -1);       // what each gene submitted: 0 scores, 1..4 newton, 5 evaluate
        std::vector<size_t> pool_off;       // pooled candidate tails: byte offsets, in tail order
        size_t nres = 0, pool_bytes = 0;
        bool any = false;
        for (int g = 0; g < n; ++g) {
            SprState &S = st[g];
            if (S.done) continue;
            const Tree &T = genes[g].tree;
            if (S.phase < 0) {
                // next batch: consecutive prunes from the cursor until the unit budget is reached
                S.batch.clear(); S.bi = 0;
                size_t units = 0;
                while (S.p < T.nnodes() && units < SPR_UNIT_BUDGET) {
                    Prune P; build_prune(g, S.p, S.ks, P);
                    cursor_next(g);
                    if (P.units.empty()) continue;
                    units += P.units.size();
                    S.batch.push_back(std::move(P));
                }
                if (S.batch.empty()) { S.done = true; continue; }
                S.phase = 0;
                // every cached message the batch reads, first
                for (const Prune &P : S.batch) {
                    need_side(g, msg(g, P.s, P.p), ops);
                    for (const Unit &u : P.units) {
                        for (const PathOp &po : u.paths) { need_side(g, po.left, ops); need_side(g, po.right, ops); }
                        need_side(g, msg(g, u.h, u.g), ops);
                    }
                }
                int count = 0;
                for (size_t q = 0; q < ops.size(); ++q) if (ops[q].gene == g) ++count;
                const size_t pl_bytes = ((size_t)genes[g].aln.mpad * 8 + 255) / 256 * 256;
                for (Prune &P : S.batch) {
                    const Side sp = msg(g, P.s, P.p);
                    P.rbase = nres;
                    for (const Unit &u : P.units) {
                        for (const PathOp &po : u.paths) {
                            PendingOp o; o.gene = g; o.out_kind = SIDE_SCRATCH; o.out_id = po.depth; o.level = 0;
                            o.child[0] = po.left; o.t[0] = po.tl; o.bv[0] = po.lbv; o.bq[0] = po.lbq;
                            o.child[1] = po.right; o.t[1] = po.tr; o.bv[1] = po.rbv; o.bq[1] = po.rbq;
                            ops.push_back(o); ++count;
                        }
                        const double tgh = T.len[u.g][T.slot(u.g, u.h)];
                        PendingOp I; I.gene = g; I.out_kind = SIDE_SCRATCH; I.out_id = SPR_INS_SLOT; I.level = 0;
                        I.transient = true;                        // read by the evaluation right behind it and by nothing else
                        I.child[0] = {SIDE_SCRATCH, u.mslot}; I.t[0] = 0.5 * tgh; I.child[1] = msg(g, u.h, u.g); I.t[1] = 0.5 * tgh;
                        ops.push_back(I); ++count;
                        Tail t{g, sp, {SIDE_SCRATCH, SPR_INS_SLOT}, MODE_EVALUATE, P.ts, 0, 0, count};
                        t.bv = P.p; t.bq = P.ks;                   // the pendant branch: one matrix set for all its candidates
                        pool_off.push_back(pool_bytes); pool_bytes += pl_bytes; ++nres;
                        tails.push_back(t);
                    }
                }
                kind[g] = 0;
            } else if (S.phase >= 1 && S.phase <= 4) {
                int u, v; apply_edge(S, S.phase, u, v);
                need(g, u, v, ops); need(g, v, u, ops);
                tails.push_back({g, msg(g, u, v), msg(g, v, u), MODE_SUMTABLE, T.len[u][T.slot(u, v)], 32});
                kind[g] = S.phase;
            } else {
                const int r = T.nbr[0][0];
                need(g, r, 0, ops);
                tails.push_back({g, msg(g, 0, r), msg(g, r, 0), MODE_EVALUATE, T.len[0][0], 0});
                kind[g] = 5;
            }
            any = true;
        }
        if (!any) break;
        ++cnt_spr;
        if (nres) {                             // pooled buffers of the candidate tails (pointers are known only now)
            if (int rc = ensure_tailpool(pool_bytes)) return rc;
            if (int rc = ensure_results(nres)) return rc;
            size_t k = 0;
            for (auto &t : tails) if (t.mode == MODE_EVALUATE && kind[t.gene] == 0) {
                t.patlnl_dev = reinterpret_cast<double *>(d_tailpool + pool_off[k]);
                t.result_dev = d_chain + 4 * k; t.result_host = h_chain + 4 * k; ++k;
            }
        }
        if (int rc = run(ops, tails)) return rc;
        for (int g = 0; g < n; ++g) {
            if (kind[g] < 0) continue;
            SprState &S = st[g]; Tree &T = genes[g].tree;
            if (kind[g] == 0) {
                for (Prune &P : S.batch) {                  // candidate scores leave the mapped result pool now: later runs reuse it
                    std::vector<double> sc(P.units.size());
                    for (size_t k = 0; k < P.units.size(); ++k) sc[k] = h_chain[4 * (P.rbase + k)];
                    P.rbase = scores_of[g].size();
                    scores_of[g].insert(scores_of[g].end(), sc.begin(), sc.end());
                }
            } else if (kind[g] <= 4) {
                int u, v; apply_edge(S, kind[g], u, v);
                const double r0 = res(g)[0], old = T.len[u][T.slot(u, v)];
                if (r0 != old) { T.set_len(u, v, r0); branch_changed(g, u, v); }
                S.phase = kind[g] + 1;
                continue;
            } else {
                const double r0 = res(g)[0];
                if (r0 > lnl[g] + 1e-6) {
                    // kept: the tree changed, so the scores of the later prunes of this batch are void -- resume
                    // right behind this prune, on the new tree
                    lnl[g] = r0; moves[g]++;
                    const Prune &P = S.batch[S.bi];
                    S.p = P.p; S.ks = P.ks; cursor_next(g);
                    S.batch.clear(); scores_of[g].clear(); S.phase = -1;
                    if (S.p >= T.nnodes()) S.done = true;
                    continue;
                }
                T = S.backup; invalidate_all(g);         // rejected: same tree as before, later scores still stand
                ++S.bi;
            }
            // read the scored prunes in order until one has a candidate that beats the current likelihood
            S.phase = -1;
            while (S.bi < S.batch.size()) {
                const Prune &P = S.batch[S.bi];
                S.best = -1e300; S.bg = S.bh = -1;
                for (size_t k = 0; k < P.units.size(); ++k) {
                    const double sc = scores_of[g][P.rbase + k];
                    if (sc > S.best) { S.best = sc; S.bg = P.units[k].g; S.bh = P.units[k].h; }
                }
                if (S.bg >= 0 && S.best > lnl[g] + SPR_MIN_GAIN) {
                    S.backup = T;
                    spr_apply(T, P.p, P.x, P.y, S.bg, S.bh);
                    invalidate_all(g);
                    S.phase = 1;
                    break;
                }
                ++S.bi;
            }
            if (S.phase < 0) { S.batch.clear(); scores_of[g].clear(); if (S.p >= T.nnodes()) S.done = true; }
        }
    }
    return 0;
}

int Batch::search(bool nni, int spr_radius, bool opt_alpha_flag, double eps, double *lnl_out) {
    const int n = (int)genes.size();
    static const bool trace = std::getenv("PML_TRACE") != nullptr;
    std::vector<double> lnl(n, 0.0);
    for (int g = 0; g < n; ++g) {         // a start tree that violates the constraints is replaced by a constrained NJ tree
        Gene &G = genes[g];
        if (!G.cons.empty() && !tree_displays(G.tree, G.cons)) { G.tree = nj_tree(G.aln, &G.cons); invalidate_all(g); ++topo_epoch; }
    }
    newton_tol = 1e-6;                    // candidate ranking and local moves: coarse Newton
    struct Restore { double &r; ~Restore() { r = 1e-8; } } restore{newton_tol};
    if (int rc = optimize(opt_alpha_flag, 0.1, lnl.data())) return rc;
    std::vector<char> active(n, (nni || spr_radius > 0) ? 1 : 0);
    for (int outer = 0; outer < 20; ++outer) {
        bool any = false; for (char a : active) any |= a;
        if (!any) break;
        std::vector<int> moves(n, 0), applied;
        std::vector<char> ract(active);
        for (int round = 0; round < 100; ++round) {
            bool anyr = false; for (char a : ract) anyr |= a;
            if (!anyr) break;
            if (int rc = nni_round(ract, lnl, applied)) return rc;
            if (trace) { int na = 0, mv = 0; for (int g = 0; g < n; ++g) { na += ract[g]; mv += applied[g]; } fprintf(stderr, "[pml] nni round %d (outer %d): %d genes active, %d moves; cumulative smooth-steps %ld nni-steps %ld\n", round, outer, na, mv, cnt_smooth, cnt_nni); }
            for (int g = 0; g < n; ++g) if (ract[g]) { if (applied[g] == 0) ract[g] = 0; else moves[g] += applied[g]; }
        }
        if (spr_radius > 0) {
            std::vector<char> sact(active);
            std::vector<int> smoves;
            for (int round = 0; round < 10; ++round) {
                bool anys = false; for (char a : sact) anys |= a;
                if (!anys) break;
                if (int rc = spr_round(sact, spr_radius, lnl, smoves)) return rc;
                for (int g = 0; g < n; ++g) if (sact[g]) { if (smoves[g] == 0) sact[g] = 0; else moves[g] += smoves[g]; }
            }
        }
        if (int rc = optimize(opt_alpha_flag, 0.1, lnl.data(), &active)) return rc;
        for (int g = 0; g < n; ++g) if (active[g] && moves[g] == 0) active[g] = 0;
    }
    if (trace) fprintf(stderr, "[pml] before final optimize: passes %ld smooth-steps %ld nni-steps %ld spr-steps %ld evals %ld\n", cnt_passes, cnt_smooth, cnt_nni, cnt_spr, cnt_eval);
    newton_tol = 1e-8;
    if (int rc = optimize(opt_alpha_flag, eps, lnl.data())) return rc;
    if (trace) fprintf(stderr, "[pml] search done: passes %ld smooth-steps %ld nni-steps %ld spr-steps %ld evals %ld\n", cnt_passes, cnt_smooth, cnt_nni, cnt_spr, cnt_eval);
    if (trace) fprintf(stderr, "[pml] host ms: pass set-up %.1f, pass steps (overlapped) %.1f, pass sync wait %.1f, pass post %.1f, nni build %.1f, nni run %.1f, nni select %.1f, alpha host %.1f, builds of synchronised launches %.1f\n",
                       host_phase_ms[HP_PASS_SETUP], host_phase_ms[HP_PASS_STEPS], host_phase_ms[HP_PASS_SYNC], host_phase_ms[HP_PASS_POST], host_phase_ms[HP_NNI_BUILD],
                       host_phase_ms[HP_NNI_RUN], host_phase_ms[HP_NNI_SELECT], host_phase_ms[HP_ALPHA_HOST], host_phase_ms[HP_RUN_SYNCED]);
    for (int g = 0; g < n; ++g) lnl_out[g] = lnl[g];
    for (int g = 0; g < n; ++g) if (!tree_displays(genes[g].tree, genes[g].cons)) return ctx->fail(-5, "internal: result violates the topological constraints");
    return 0;
}

}  // namespace pml
