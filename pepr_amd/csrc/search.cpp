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

#include "engine.hpp"

namespace pml {

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

// <= 2 smoothing passes (stop when max |dt| < 1e-3), then lnL
int Batch::light_smooth(const std::vector<char> &active, double *lnl) {
    std::vector<char> sm(active);
    std::vector<double> md;
    for (int pass = 0; pass < 2; ++pass) {
        bool any = false; for (char a : sm) any |= a;
        if (!any) break;
        if (int rc = smooth_pass(sm, md)) return rc;
        for (size_t g = 0; g < sm.size(); ++g) if (sm[g] && md[g] < 1e-3) sm[g] = 0;
    }
    return evaluate(active, lnl);
}

int Batch::nni_round(const std::vector<char> &active, std::vector<double> &lnl, std::vector<int> &applied) {
    const int n = (int)genes.size();
    applied.assign(n, 0);
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
    for (size_t step = 0; step < maxsteps; ++step) {
        std::vector<PendingOp> ops; std::vector<Tail> tails;
        for (int g = 0; g < n; ++g) {
            if (!active[g] || step >= edges[g].size() * 3) continue;
            const Tree &T = genes[g].tree;
            auto [u, v] = edges[g][step / 3];
            const int alt = (int)(step % 3);
            const double t0 = T.len[u][T.slot(u, v)];
            if (alt == 0) {
                need(g, u, v, ops); need(g, v, u, ops);
                tails.push_back({g, msg(g, u, v), msg(g, v, u), MODE_SUMTABLE, t0, 32});
            } else {
                int a[2], c[2]; double la[2], lc[2];
                others(T, u, v, a, la); others(T, v, u, c, lc);
                const int y = (alt == 1) ? 0 : 1;
                need(g, a[0], u, ops); need(g, a[1], u, ops); need(g, c[0], v, ops); need(g, c[1], v, ops);
                PendingOp X; X.gene = g; X.out_kind = SIDE_SCRATCH; X.out_id = 0; X.level = 0;
                X.child[0] = msg(g, a[0], u); X.t[0] = la[0]; X.child[1] = msg(g, c[y], v); X.t[1] = lc[y];
                PendingOp Y; Y.gene = g; Y.out_kind = SIDE_SCRATCH; Y.out_id = 1; Y.level = 0;
                Y.child[0] = msg(g, a[1], u); Y.t[0] = la[1]; Y.child[1] = msg(g, c[1 - y], v); Y.t[1] = lc[1 - y];
                ops.push_back(X); ops.push_back(Y);
                tails.push_back({g, {SIDE_SCRATCH, 0}, {SIDE_SCRATCH, 1}, MODE_SUMTABLE, t0, 32});
            }
        }
        if (int rc = run(ops, tails)) return rc;
        for (auto &t : tails) { Tn[t.gene][step] = h_scalars[8 * t.gene]; L[t.gene][step] = h_scalars[8 * t.gene + 1]; }
    }
    // candidate selection and application
    std::vector<std::vector<Cand>> cands(n);
    std::vector<Tree> backup(n);
    std::vector<double> lnl0(lnl);
    std::vector<char> stageA(n, 0);
    for (int g = 0; g < n; ++g) {
        if (!active[g]) continue;
        for (size_t e = 0; e < edges[g].size(); ++e) {
            const double Lc = L[g][3 * e], L1 = L[g][3 * e + 1], L2 = L[g][3 * e + 2];
            const int best = (L2 > L1) ? 2 : 1;
            const double gain = (best == 1 ? L1 : L2) - Lc;
            if (gain > NNI_MIN_GAIN) cands[g].push_back({edges[g][e].first, edges[g][e].second, best, gain, Tn[g][3 * e + best], (int)cands[g].size()});
        }
        if (cands[g].empty()) continue;
        std::sort(cands[g].begin(), cands[g].end(), [](const Cand &a, const Cand &b) { return a.gain != b.gain ? a.gain > b.gain : a.order < b.order; });
        Tree &T = genes[g].tree;
        backup[g] = T;
        std::vector<char> used(T.nnodes(), 0);
        for (auto &c : cands[g]) {
            if (used[c.u] || used[c.v]) continue;
            used[c.u] = used[c.v] = 1;
            nni_apply(T, c.u, c.v, c.alt, c.t); applied[g]++;
        }
        invalidate_all(g); stageA[g] = 1;
    }
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
        nni_apply(genes[g].tree, c.u, c.v, c.alt, c.t); applied[g] = 1;
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

int Batch::search(bool nni, int spr_radius, bool opt_alpha_flag, double eps, double *lnl_out) {
    (void)spr_radius;
    const int n = (int)genes.size();
    std::vector<double> lnl(n, 0.0);
    if (int rc = optimize(opt_alpha_flag, 0.1, lnl.data())) return rc;
    std::vector<char> active(n, nni ? 1 : 0);
    for (int outer = 0; outer < 20; ++outer) {
        bool any = false; for (char a : active) any |= a;
        if (!any) break;
        std::vector<int> moves(n, 0), applied;
        std::vector<char> ract(active);
        for (int round = 0; round < 100; ++round) {
            bool anyr = false; for (char a : ract) anyr |= a;
            if (!anyr) break;
            if (int rc = nni_round(ract, lnl, applied)) return rc;
            for (int g = 0; g < n; ++g) if (ract[g]) { if (applied[g] == 0) ract[g] = 0; else moves[g] += applied[g]; }
        }
        if (int rc = optimize(opt_alpha_flag, 0.1, lnl.data(), &active)) return rc;
        for (int g = 0; g < n; ++g) if (active[g] && moves[g] == 0) active[g] = 0;
    }
    if (int rc = optimize(opt_alpha_flag, eps, lnl.data())) return rc;
    for (int g = 0; g < n; ++g) lnl_out[g] = lnl[g];
    return 0;
}

}  // namespace pml
