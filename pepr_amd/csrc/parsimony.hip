// parsimony.hip -- maximum-parsimony start trees on the device (`raxmlHPC -f d -y`,
// reference call site RAxMLRunner.java:215-251).
//
// Fitch state sets are one uint32 per (node direction, pattern): bit s = amino acid s possible.
// All genes of a batch advance in lock step; every device step is ONE launch of k_fitch, which
// walks a dependency-ordered list of set operations per (gene, group) for a block of 256
// patterns.  Patterns are independent, so a thread only ever re-reads what it wrote itself and the
// walk needs no barrier.  Pure integer, HBM/L2-bound work: 4-16 B per pattern per operation.
//
//   SET    out = F(l, r)                       F(l,r) = (l & r) ? (l & r) : (l | r)
//   COUNT  score += w * [l & r == 0]           (tree length: one per inner node + the root edge)
//   COSTX  score += w * [F(l, r) & x == 0]     (cost of hanging set x on the edge whose two sides are l, r)
//   PATH   out = F(l, r); score += w * [F(out, y) & x == 0]   (SPR: carry the pruned tree's message one
//                                               edge further and price the regraft on that edge)
// Scores are reduced inside a wavefront (DPP) and stored per wave (no atomics); k_fitch_reduce sums
// the waves.  The algorithm (addition order, enumeration orders, tie breaks) is the one spelled out in
// oracle/pml_oracle.c so results are bit-identical to the oracle.
#include <algorithm>
#include <cstring>
#include <functional>

#include "engine.hpp"

namespace pml {

struct FOp { int kind, out, l, r, y, x, score, pad; };       // vector ids inside the gene's pool
enum { F_SET = 0, F_COUNT = 1, F_COSTX = 2, F_PATH = 3, F_SETCOUNT = 4 };
struct FRun {
    const FOp *ops; unsigned *pool; const int *w; int *partial;   // partial[score * nwaves + wave]
    int nops, mpad, nwaves, pad;
};
struct FRed { const int *partial; int *out; int nscores, nwaves; };

typedef const unsigned __attribute__((address_space(1))) *gcu32;
typedef unsigned __attribute__((address_space(1))) *gu32;

__device__ __forceinline__ unsigned fitch(unsigned l, unsigned r) { const unsigned x = l & r; return x ? x : (l | r); }

// sum over the 64 lanes, result uniform
__device__ __forceinline__ int wave_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);   // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);   // row_mirror: every lane holds its row's sum
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) +
           __builtin_amdgcn_readlane(v, 48);
}

__global__ __launch_bounds__(256) void k_fitch(const FRun *__restrict__ runs) {
    const FRun run = runs[blockIdx.y];
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= run.mpad) return;
    const bool live = p < run.mpad;
    const size_t m = (size_t)run.mpad;
    gu32 pool = (gu32)run.pool;
    const int w = live ? run.w[p] : 0;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool lane0 = (threadIdx.x & 63) == 0;
    for (int i = 0; i < run.nops; ++i) {
        const FOp op = run.ops[i];                           // uniform -> scalar loads
        unsigned l = 0xFFFFF, r = 0xFFFFF;
        if (live) { l = pool[op.l * m + p]; r = pool[op.r * m + p]; }
        int add = 0;
        if (op.kind == F_SET || op.kind == F_SETCOUNT) {
            if (live) pool[op.out * m + p] = fitch(l, r);
            if (op.kind == F_SET) continue;
            add = (l & r) ? 0 : w;
        } else if (op.kind == F_COUNT) {
            add = (l & r) ? 0 : w;
        } else if (op.kind == F_COSTX) {
            const unsigned x = live ? pool[op.x * m + p] : 0xFFFFF;
            add = (fitch(l, r) & x) ? 0 : w;
        } else {                                              // F_PATH
            const unsigned o = fitch(l, r);
            unsigned y = 0xFFFFF, x = 0xFFFFF;
            if (live) { pool[op.out * m + p] = o; y = pool[op.y * m + p]; x = pool[op.x * m + p]; }
            add = (fitch(o, y) & x) ? 0 : w;
        }
        const int s = wave_sum(add);
        if (lane0) run.partial[(size_t)op.score * run.nwaves + wave] = s;
    }
}

__global__ __launch_bounds__(256) void k_fitch_reduce(const FRed *__restrict__ reds) {
    const FRed rd = reds[blockIdx.y];
    for (int s = blockIdx.x * 256 + threadIdx.x; s < rd.nscores; s += gridDim.x * 256) {
        int acc = 0;
        for (int k = 0; k < rd.nwaves; ++k) acc += rd.partial[(size_t)s * rd.nwaves + k];
        rd.out[s] = acc;
    }
}

#define PCHK(expr)                                                                                 \
    do { hipError_t e_ = (expr);                                                                   \
         if (e_ != hipSuccess) return ctx->fail(-4, std::string("parsimony: ") + hipGetErrorString(e_)); } while (0)

static unsigned code_mask(int c) {
    if (c < 20) return 1u << c;
    if (c == 20) return (1u << 2) | (1u << 3);     // B = N | D
    if (c == 21) return (1u << 5) | (1u << 6);     // Z = Q | E
    return 0xFFFFFu;
}
static uint64_t splitmix(uint64_t &s) {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

namespace {
struct PGene {
    EncodedAlignment aln; Tree tree; std::vector<int> order;
    int nvec = 0, ngroups = 1, nwaves = 0;
    unsigned *d_pool = nullptr; int *d_w = nullptr;
    int vtip(int i) const { return i; }
    int vmsg(int v, int k) const { return aln.ntax + (v - aln.ntax) * 3 + k; }
    int vpath(int grp, int depth, int radius) const { return aln.ntax + 3 * (aln.ntax - 2) + grp * (radius + 2) + depth; }
    int vside(int v, int to) const { return v < aln.ntax ? v : vmsg(v, tree.slot(v, to)); }   // S(v -> to)
    // ops that refresh every directed message of the present tree (post-order towards `root_tip`, then
    // pre-order away from it); with `count`, the post-order half also scores the tree length
    void refresh_ops(int root_tip, std::vector<FOp> &ops, int *nscore) const {
        const int r = tree.nbr[root_tip][0];
        std::function<void(int, int)> post = [&](int v, int from) {
            if (v < aln.ntax) return;
            const int k = tree.slot(v, from), x = tree.nbr[v][(k + 1) % 3], y = tree.nbr[v][(k + 2) % 3];
            post(x, v); post(y, v);
            FOp o{nscore ? F_SETCOUNT : F_SET, vmsg(v, k), vside(x, v), vside(y, v), 0, 0, nscore ? (*nscore)++ : 0, 0};
            ops.push_back(o);
        };
        std::function<void(int, int)> pre = [&](int v, int from) {      // messages v -> children (away from `from`)
            if (v < aln.ntax) return;
            const int k = tree.slot(v, from);
            for (int j = 1; j <= 2; ++j) {
                const int h = tree.nbr[v][(k + j) % 3], o = tree.nbr[v][(k + 3 - j) % 3];
                ops.push_back(FOp{F_SET, vmsg(v, (k + j) % 3), vside(from, v), vside(o, v), 0, 0, 0, 0});
                pre(h, v);
            }
        };
        post(r, root_tip);
        if (nscore) ops.push_back(FOp{F_COUNT, 0, vside(r, root_tip), vtip(root_tip), 0, 0, (*nscore)++, 0});
        pre(r, root_tip);
    }
};
struct Stage {
    Ctx *ctx; void *d = nullptr; size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (d) hipFree(d);
        cap = bytes + bytes / 2; d = nullptr;
        PCHK(hipMalloc(&d, cap));
        return 0;
    }
    ~Stage() { if (d) hipFree(d); }
};
}  // namespace

// one device step: runs[i] walks ops_of[i]; returns the reduced scores of every run
static int fitch_step(Ctx *ctx, Stage &st, Stage &sc, const std::vector<PGene *> &owner, const std::vector<std::vector<FOp>> &ops_of,
                      const std::vector<int> &nscores, std::vector<std::vector<int>> &scores) {
    const size_t nr = owner.size();
    if (!nr) return 0;
    size_t nops = 0, npart = 0, nsc = 0; int max_mpad = 0, max_sc = 0;
    for (size_t i = 0; i < nr; ++i) { nops += ops_of[i].size(); npart += (size_t)nscores[i] * owner[i]->nwaves; nsc += nscores[i];
                                      max_mpad = std::max(max_mpad, owner[i]->aln.mpad); max_sc = std::max(max_sc, nscores[i]); }
    const size_t o_runs = 0, o_reds = o_runs + nr * sizeof(FRun), o_ops = o_reds + nr * sizeof(FRed), total = o_ops + nops * sizeof(FOp);
    if (int rc = st.ensure(total)) return rc;
    if (int rc = sc.ensure((npart + nsc) * sizeof(int))) return rc;
    std::vector<char> h(total);
    FRun *runs = (FRun *)(h.data() + o_runs); FRed *reds = (FRed *)(h.data() + o_reds); FOp *ops = (FOp *)(h.data() + o_ops);
    int *d_part = (int *)sc.d, *d_out = d_part + npart;
    size_t po = 0, pp = 0, ps = 0;
    for (size_t i = 0; i < nr; ++i) {
        std::memcpy(ops + po, ops_of[i].data(), ops_of[i].size() * sizeof(FOp));
        runs[i] = FRun{(const FOp *)((char *)st.d + o_ops) + po, owner[i]->d_pool, owner[i]->d_w, d_part + pp, (int)ops_of[i].size(), owner[i]->aln.mpad, owner[i]->nwaves, 0};
        reds[i] = FRed{d_part + pp, d_out + ps, nscores[i], owner[i]->nwaves};
        po += ops_of[i].size(); pp += (size_t)nscores[i] * owner[i]->nwaves; ps += nscores[i];
    }
    PCHK(hipMemcpyAsync(st.d, h.data(), total, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_fitch, dim3((max_mpad + 255) / 256, (unsigned)nr), dim3(256), 0, ctx->stream, (const FRun *)((char *)st.d + o_runs));
    if (nsc) hipLaunchKernelGGL(k_fitch_reduce, dim3(std::min(64, (max_sc + 255) / 256), (unsigned)nr), dim3(256), 0, ctx->stream, (const FRed *)((char *)st.d + o_reds));
    PCHK(hipGetLastError());
    std::vector<int> flat(nsc);
    if (nsc) PCHK(hipMemcpyAsync(flat.data(), d_out, nsc * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    if (int rc_ = ctx->sync(ctx->stream)) return rc_;
    scores.assign(nr, {});
    ps = 0;
    for (size_t i = 0; i < nr; ++i) { scores[i].assign(flat.begin() + ps, flat.begin() + ps + nscores[i]); ps += nscores[i]; }
    return 0;
}

// Parsimony trees of n genes: randomised stepwise addition (seed 0 = input order) + SPR hill climbing
// within `radius` edges (0 = none).  trees_out[g] has branch lengths 0.1; lengths_out[g] = weighted Fitch length.
int parsimony_batch(Ctx *ctx, int n, const pml_alignment_view *alns, unsigned seed, int radius,
                    std::vector<Tree> &trees_out, std::vector<EncodedAlignment> &alns_out, std::vector<long long> &lengths_out, std::vector<int> &moves_out) {
    std::vector<PGene> G(n);
    Stage st{ctx}, sc{ctx};
    size_t total_blocks = 0;
    for (int g = 0; g < n; ++g) {
        std::string err;
        if (!G[g].aln.encode(alns[g].ntax, alns[g].nsites, alns[g].names, alns[g].rows, err)) return ctx->fail(-2, "gene " + std::to_string(g) + ": " + err);
        if (G[g].aln.ntax < 3) return ctx->fail(-2, "gene " + std::to_string(g) + ": parsimony needs at least 3 taxa");
        total_blocks += (G[g].aln.mpad + 255) / 256;
    }
    // groups: independent prunes of one gene spread over several workgroups when the batch alone cannot fill 256 CUs
    const int want_groups = radius > 0 ? (int)std::max<size_t>(1, std::min<size_t>(64, (2048 + total_blocks - 1) / std::max<size_t>(total_blocks, 1))) : 1;
    struct Free { std::vector<PGene> &G; ~Free() { for (auto &g : G) { if (g.d_pool) hipFree(g.d_pool); if (g.d_w) hipFree(g.d_w); } } } freer{G};
    for (int g = 0; g < n; ++g) {
        PGene &pg = G[g]; const int nt = pg.aln.ntax, mp = pg.aln.mpad;
        pg.ngroups = want_groups; pg.nwaves = ((mp + 255) / 256) * 4;
        pg.nvec = nt + 3 * (nt - 2) + pg.ngroups * (radius + 2);
        PCHK(hipMalloc((void **)&pg.d_pool, (size_t)pg.nvec * mp * sizeof(unsigned)));
        PCHK(hipMalloc((void **)&pg.d_w, (size_t)mp * sizeof(int)));
        std::vector<unsigned> tips((size_t)nt * mp); std::vector<int> w(mp);
        for (size_t i = 0; i < tips.size(); ++i) tips[i] = code_mask(pg.aln.codes[i]);
        for (int p = 0; p < mp; ++p) w[p] = (int)pg.aln.weight[p];
        PCHK(hipMemcpy(pg.d_pool, tips.data(), tips.size() * sizeof(unsigned), hipMemcpyHostToDevice));
        PCHK(hipMemcpy(pg.d_w, w.data(), w.size() * sizeof(int), hipMemcpyHostToDevice));
        pg.order.resize(nt);
        for (int i = 0; i < nt; ++i) pg.order[i] = i;
        if (seed) { uint64_t s = seed; for (int i = nt - 1; i >= 1; --i) std::swap(pg.order[i], pg.order[(int)(splitmix(s) % (uint64_t)(i + 1))]); }
        Tree &t = pg.tree; t.ntax = nt;
        t.nbr.assign(2 * nt - 2, {-1, -1, -1}); t.len.assign(2 * nt - 2, {0.1, 0.1, 0.1});
        for (int k = 0; k < 3; ++k) { t.nbr[nt][k] = pg.order[k]; t.nbr[pg.order[k]][0] = nt; }
    }
    std::vector<PGene *> owner; std::vector<std::vector<FOp>> ops_of; std::vector<int> nscores; std::vector<std::vector<int>> scores;
    // ---- stepwise addition: one launch per step for all genes still growing ----
    int max_tax = 0; for (auto &g : G) max_tax = std::max(max_tax, g.aln.ntax);
    for (int s = 3; s < max_tax; ++s) {
        owner.clear(); ops_of.clear(); nscores.clear();
        std::vector<std::vector<std::pair<int, int>>> edges;
        for (auto &pg : G) {
            if (s >= pg.aln.ntax) continue;
            const Tree &t = pg.tree; const int x = pg.order[s];
            std::vector<FOp> ops; pg.refresh_ops(pg.order[0], ops, nullptr);
            std::vector<std::pair<int, int>> ed;
            for (int v = 0; v < t.nnodes(); ++v) {
                if (t.nbr[v][0] < 0) continue;
                for (int k = 0; k < 3; ++k) { const int u = t.nbr[v][k]; if (u < 0 || u < v) continue;
                    ops.push_back(FOp{F_COSTX, 0, pg.vside(v, u), pg.vside(u, v), 0, pg.vtip(x), (int)ed.size(), 0}); ed.push_back({v, u}); }
            }
            owner.push_back(&pg); ops_of.push_back(std::move(ops)); nscores.push_back((int)ed.size()); edges.push_back(std::move(ed));
        }
        if (int rc = fitch_step(ctx, st, sc, owner, ops_of, nscores, scores)) return rc;
        for (size_t i = 0; i < owner.size(); ++i) {
            PGene &pg = *owner[i]; Tree &t = pg.tree; const int nt = pg.aln.ntax, x = pg.order[s], w = nt + s - 2;
            int best = 0;
            for (int e = 1; e < (int)scores[i].size(); ++e) if (scores[i][e] < scores[i][best]) best = e;
            const int bv = edges[i][best].first, bu = edges[i][best].second;
            t.nbr[bv][t.slot(bv, bu)] = w; t.nbr[bu][t.slot(bu, bv)] = w;
            t.nbr[w] = {bv, bu, x}; t.nbr[x][0] = w;
        }
    }
    // ---- SPR hill climbing: per round one refresh launch + one scoring launch over (gene, group) ----
    moves_out.assign(n, 0);
    std::vector<char> active(n, 0);
    for (int g = 0; g < n; ++g) active[g] = radius > 0 && G[g].aln.ntax > 4;
    struct Cand { int v, k, g, h, base; };       // score index -> move; base = index of the prune's own cost
    for (int round = 0;; ++round) {
        owner.clear(); ops_of.clear(); nscores.clear();
        for (int g = 0; g < n; ++g) if (active[g] && round < 20 * G[g].aln.ntax) { std::vector<FOp> ops; G[g].refresh_ops(0, ops, nullptr); owner.push_back(&G[g]); ops_of.push_back(std::move(ops)); nscores.push_back(0); } else active[g] = 0;
        if (owner.empty()) break;
        if (int rc = fitch_step(ctx, st, sc, owner, ops_of, nscores, scores)) return rc;
        // scoring runs.  Budget the launch: genes are taken until ~8M ops are staged, then launched.
        std::vector<int> gene_of_owner;
        for (int g = 0; g < n; ++g) if (active[g]) gene_of_owner.push_back(g);
        size_t cursor = 0;
        while (cursor < gene_of_owner.size()) {
            owner.clear(); ops_of.clear(); nscores.clear();
            std::vector<std::vector<Cand>> cands; std::vector<int> run_gene;
            size_t staged = 0;
            for (; cursor < gene_of_owner.size() && (staged == 0 || staged < (8u << 20)); ++cursor) {
                const int g = gene_of_owner[cursor]; PGene &pg = G[g]; const Tree &t = pg.tree; const int nt = pg.aln.ntax;
                std::vector<std::vector<FOp>> gops(pg.ngroups); std::vector<std::vector<Cand>> gc(pg.ngroups);
                int prune_no = 0;
                for (int v = nt; v < t.nnodes(); ++v) for (int k = 0; k < 3; ++k) {
                    const int sub = t.nbr[v][k], x = t.nbr[v][(k + 1) % 3], y = t.nbr[v][(k + 2) % 3];
                    if (x < nt && y < nt) continue;
                    const int grp = prune_no++ % pg.ngroups;
                    auto &ops = gops[grp]; auto &cs = gc[grp];
                    const int P = pg.vside(sub, v), base = (int)cs.size();
                    ops.push_back(FOp{F_COSTX, 0, pg.vside(x, v), pg.vside(y, v), 0, P, base, 0}); cs.push_back(Cand{v, k, -1, -1, base});
                    std::function<void(int, int, int, int)> explore = [&](int M0, int gnode, int from, int depth) {
                        if (gnode < nt) return;
                        const int kf = t.slot(gnode, from);
                        for (int j = 1; j <= 2; ++j) {
                            const int h = t.nbr[gnode][(kf + j) % 3], o = t.nbr[gnode][(kf + 3 - j) % 3];
                            const int M1 = pg.vpath(grp, depth, radius);
                            ops.push_back(FOp{F_PATH, M1, M0, pg.vside(o, gnode), pg.vside(h, gnode), P, (int)cs.size(), 0});
                            cs.push_back(Cand{v, k, gnode, h, base});
                            if (depth < radius) explore(M1, h, gnode, depth + 1);
                        }
                    };
                    explore(pg.vside(y, v), x, v, 1);
                    explore(pg.vside(x, v), y, v, 1);
                }
                for (int grp = 0; grp < pg.ngroups; ++grp) {
                    if (gops[grp].empty()) continue;
                    staged += gops[grp].size();
                    owner.push_back(&pg); nscores.push_back((int)gc[grp].size()); ops_of.push_back(std::move(gops[grp])); cands.push_back(std::move(gc[grp])); run_gene.push_back(g);
                }
            }
            if (int rc = fitch_step(ctx, st, sc, owner, ops_of, nscores, scores)) return rc;
            // best move per gene: largest gain, first in the oracle's enumeration order (prune order, then DFS order).
            // groups interleave prunes round-robin, so order candidates by (prune number, position) = (v, k, index).
            struct Best { long long gain = 0; int v = -1, k = 0, g = 0, h = 0; long long key = 0; };
            std::vector<Best> best(n);
            for (size_t i = 0; i < owner.size(); ++i) {
                const int g = run_gene[i]; Best &b = best[g];
                for (size_t c = 0; c < cands[i].size(); ++c) {
                    const Cand &cd = cands[i][c]; if (cd.g < 0) continue;
                    const long long gain = (long long)scores[i][cd.base] - scores[i][c];
                    const long long key = ((long long)(cd.v * 3 + cd.k) << 32) | (long long)c;
                    if (gain > b.gain || (gain == b.gain && gain > 0 && key < b.key)) { b.gain = gain; b.v = cd.v; b.k = cd.k; b.g = cd.g; b.h = cd.h; b.key = key; }
                }
            }
            for (size_t i = 0; i < owner.size(); ++i) {
                const int g = run_gene[i]; if (i && run_gene[i - 1] == g) continue;
                Best &b = best[g];
                if (b.gain <= 0) { active[g] = 0; continue; }
                Tree &t = G[g].tree; const int a = t.nbr[b.v][(b.k + 1) % 3], bb = t.nbr[b.v][(b.k + 2) % 3];
                t.nbr[a][t.slot(a, b.v)] = bb; t.nbr[bb][t.slot(bb, b.v)] = a;
                t.nbr[b.g][t.slot(b.g, b.h)] = b.v; t.nbr[b.h][t.slot(b.h, b.g)] = b.v;
                t.nbr[b.v][(b.k + 1) % 3] = b.g; t.nbr[b.v][(b.k + 2) % 3] = b.h;
                moves_out[g]++;
            }
        }
    }
    // ---- final lengths ----
    owner.clear(); ops_of.clear(); nscores.clear();
    for (auto &pg : G) { std::vector<FOp> ops; int ns = 0; pg.refresh_ops(0, ops, &ns); owner.push_back(&pg); ops_of.push_back(std::move(ops)); nscores.push_back(ns); }
    if (int rc = fitch_step(ctx, st, sc, owner, ops_of, nscores, scores)) return rc;
    trees_out.resize(n); alns_out.resize(n); lengths_out.assign(n, 0);
    for (int g = 0; g < n; ++g) { for (int v : scores[g]) lengths_out[g] += v; trees_out[g] = G[g].tree; alns_out[g] = std::move(G[g].aln); }
    return 0;
}

}  // namespace pml
