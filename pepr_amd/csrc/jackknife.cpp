// jackknife.cpp -- gene-wise jackknife driver: the reference's data-parallel loop
// (PhylogenomicPipeline2.java:994-1126, 1227-1275, 1587-1633) as one native call.
// Concatenation follows MSAConcatenator.concatenate (MSAConcatenator.java:78-189): taxa = sorted
// union of the genes' taxon names, a gene that lacks a taxon contributes '?' columns.
// Gene alignments are encoded once into HBM; the full alignment and every replicate are gathered from them on
// the device (GeneStore / Batch::create_replicates) and the replicates are searched as ONE batch; the support counting is TreeSupportDecorator.addSupportValues (:86-163).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <set>

#include "../../include/peprml.h"
#include "api_types.hpp"

using namespace pml;

namespace {
struct Concat { std::vector<std::string> names, rows; };

bool concatenate(int ngenes, const pml_alignment *genes, const std::vector<int> &sel, Concat &out, std::string &err) {
    std::set<std::string> uni;
    for (int g : sel) {
        if (g < 0 || g >= ngenes) { err = "gene index out of range"; return false; }
        const pml_alignment &A = genes[g];
        if (!A.names || !A.rows || A.ntax <= 0) { err = "bad alignment"; return false; }
        for (int i = 0; i < A.ntax; ++i) uni.insert(A.names[i]);
    }
    out.names.assign(uni.begin(), uni.end());           // std::set order == Arrays.sort for ASCII names
    out.rows.assign(out.names.size(), std::string());
    for (int g : sel) {
        const pml_alignment &A = genes[g];
        std::map<std::string, int> idx;
        for (int i = 0; i < A.ntax; ++i) idx[A.names[i]] = i;
        for (size_t t = 0; t < out.names.size(); ++t) {
            auto it = idx.find(out.names[t]);
            if (it == idx.end()) out.rows[t].append((size_t)A.nsites, '?');
            else {
                if ((int)strnlen(A.rows[it->second], (size_t)A.nsites) < A.nsites) { err = "row shorter than nsites"; return false; }
                out.rows[t].append(A.rows[it->second], (size_t)A.nsites);
            }
        }
    }
    return true;
}
// replicate r = `subset` genes drawn without replacement (partial Fisher-Yates on one mt19937_64 stream), sorted
std::vector<std::vector<int>> draw_subsets(int ngenes, int reps, int subset, unsigned long long seed) {
    std::mt19937_64 rng(seed);
    std::vector<std::vector<int>> rep((size_t)reps);
    std::vector<int> all(ngenes); for (int i = 0; i < ngenes; ++i) all[i] = i;
    for (int r = 0; r < reps; ++r) {
        std::vector<int> pool(all);
        for (int i = 0; i < subset; ++i) { const size_t j = i + (size_t)(rng() % (uint64_t)(ngenes - i)); std::swap(pool[i], pool[j]); }
        rep[r].assign(pool.begin(), pool.begin() + subset);
        std::sort(rep[r].begin(), rep[r].end());
    }
    return rep;
}
char *dup_cstr(const std::string &s) { char *p = (char *)std::malloc(s.size() + 1); if (p) std::memcpy(p, s.c_str(), s.size() + 1); return p; }
}  // namespace

extern "C" int pml_concatenate(int ngenes, const pml_alignment *genes, int nsel, const int *sel, char **fasta_out) {
    if (!genes || ngenes <= 0 || !fasta_out) return PML_EINVAL;
    *fasta_out = nullptr;
    try {
        std::vector<int> s;
        if (sel) s.assign(sel, sel + nsel); else { s.resize(ngenes); for (int i = 0; i < ngenes; ++i) s[i] = i; }
        Concat c; std::string err;
        if (!concatenate(ngenes, genes, s, c, err)) return PML_EINVAL;
        std::string txt;
        for (size_t i = 0; i < c.names.size(); ++i) { txt += '>'; txt += c.names[i]; txt += '\n'; txt += c.rows[i]; txt += '\n'; }
        *fasta_out = dup_cstr(txt);
    } catch (const std::exception &) { return PML_ENOMEM; }
    return *fasta_out ? PML_OK : PML_ENOMEM;
}

extern "C" int pml_jackknife_draw(int ngenes, int reps, int subset_size, unsigned long long seed, int *sel_out) {
    if (ngenes <= 0 || reps < 0 || !sel_out) return PML_EINVAL;
    int subset = subset_size > 0 ? subset_size : ngenes / 2;
    subset = std::max(1, std::min(subset, ngenes));
    try {
        const auto rep = draw_subsets(ngenes, reps, subset, seed);
        for (int r = 0; r < reps; ++r) std::copy(rep[r].begin(), rep[r].end(), sel_out + (size_t)r * subset);
    } catch (const std::exception &) { return PML_ENOMEM; }
    return subset;
}

// Test hook for SURVEY 8f-3: the code matrix + weights k_gather builds on the device for one gene selection, read back.
extern "C" int pml_debug_gather(pml_ctx *ctx, int ngenes, const pml_alignment *genes, int nsel, const int *sel,
                                int *ntax_out, int *npat_out, int *mpad_out, unsigned char **codes_out, double **weights_out,
                                char **names_out) {
    if (!ctx || !genes || ngenes <= 0 || !ntax_out || !npat_out || !mpad_out || !codes_out || !weights_out || !names_out) return PML_EINVAL;
    *codes_out = nullptr; *weights_out = nullptr; *names_out = nullptr;
    std::lock_guard<std::mutex> lk(ctx->c.mu);
    pml_drop_worker_caches(ctx);
    try {
        for (int g = 0; g < ngenes; ++g) if (!genes[g].names || !genes[g].rows || genes[g].ntax <= 0) return ctx->c.fail(PML_EINVAL, "bad alignment");
        std::vector<int> s;
        if (sel) s.assign(sel, sel + nsel); else { s.resize(ngenes); for (int i = 0; i < ngenes; ++i) s[i] = i; }
        GeneStore store;
        struct Drop { GeneStore &s; ~Drop() { s.destroy(); } } drop{store};
        if (int rc = store.create(&ctx->c, ngenes, reinterpret_cast<const pml_alignment_view *>(genes))) return rc;
        Batch b;
        struct DropB { Batch &b; ~DropB() { b.destroy(); } } dropb{b};
        if (int rc = b.create_replicates(&ctx->c, store, {s}, 0, 4, 1.0)) return rc;
        const Gene &G = b.genes[0];
        const size_t nt = (size_t)G.aln.ntax, mp = (size_t)G.aln.mpad;
        unsigned char *codes = (unsigned char *)std::malloc(nt * mp); double *w = (double *)std::malloc(mp * sizeof(double));
        if (!codes || !w) { std::free(codes); std::free(w); return ctx->c.fail(PML_ENOMEM, "host allocation failed"); }
        if (hipMemcpy(codes, G.d_codes, nt * mp, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(w, G.d_weight, mp * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) {
            std::free(codes); std::free(w); return ctx->c.fail(PML_EDEVICE, "read-back of the gathered matrix failed");
        }
        std::string names; for (auto &n : G.aln.names) { names += n; names += '\n'; }
        *ntax_out = (int)nt; *npat_out = G.aln.npat; *mpad_out = (int)mp; *codes_out = codes; *weights_out = w; *names_out = dup_cstr(names);
    } catch (const std::bad_alloc &) { return ctx->c.fail(PML_ENOMEM, "host allocation failed"); }
    catch (const std::exception &e) { return ctx->c.fail(PML_EINVAL, e.what()); }
    return PML_OK;
}

extern "C" int pml_jackknife(pml_ctx *ctx, int ngenes, const pml_alignment *genes, const pml_model *model,
                             const pml_jackknife_opts *opts, pml_result *main_out, char **support_out) {
    if (!ctx || !genes || ngenes <= 0 || !main_out) return PML_EINVAL;
    pml_fpguard fpg;
    std::memset(main_out, 0, sizeof *main_out);
    if (support_out) *support_out = nullptr;
    const int reps = opts ? opts->reps : 100;
    int subset = (opts && opts->subset_size > 0) ? opts->subset_size : ngenes / 2;
    subset = std::max(1, std::min(subset, ngenes));
    const double eps = (opts && opts->epsilon > 0) ? opts->epsilon : 1e-3;
    const int spr_full = opts ? opts->spr_radius_full : 5;
    if (reps < 0) return PML_EINVAL;
    const int sworld = (opts && opts->shard_world > 1) ? opts->shard_world : 1, srank = sworld > 1 ? opts->shard_rank : 0;
    if (srank < 0 || srank >= sworld) return PML_EINVAL;
    std::lock_guard<std::mutex> lk(ctx->c.mu);
    pml_drop_worker_caches(ctx);
    try {
        std::vector<int> all(ngenes); for (int i = 0; i < ngenes; ++i) all[i] = i;
        for (int g = 0; g < ngenes; ++g) if (!genes[g].names || !genes[g].rows || genes[g].ntax <= 0) return ctx->c.fail(PML_EINVAL, "bad alignment");
        // every gene is encoded once and stays in HBM; the full alignment and each replicate are index lists over
        // that store, materialised by k_gather -- no concatenated text, no per-replicate encode
        GeneStore store;
        struct Drop { GeneStore &s; ~Drop() { s.destroy(); } } drop{store};
        static_assert(sizeof(pml_alignment) == sizeof(pml_alignment_view), "alignment view layout");
        if (int rc = store.create(&ctx->c, ngenes, reinterpret_cast<const pml_alignment_view *>(genes))) return rc;
        // replicates: seeded draw without replacement (reference: RandomSetUtils.getRandomSet, unseeded)
        std::vector<std::vector<int>> rep = draw_subsets(ngenes, reps, subset, opts ? opts->seed : 0);
        if (sworld > 1) {                       // this rank's replicates (all ranks drew the same lists)
            std::vector<std::vector<int>> mine;
            for (int r = srank; r < reps; r += sworld) mine.push_back(rep[r]);
            rep.swap(mine);
        }
        const int reps_here = (int)rep.size();
        const int ncat = model ? model->ncat : 4, pm = model ? model->pi_mode : 0;
        const double alpha = model ? model->alpha : 1.0;
        // full tree
        Tree main_tree; std::vector<std::string> main_names; double main_lnl = 0, main_alpha = alpha; int main_npat = 0, main_nsites = 0;
        if (srank == 0) {
            Batch b; int rc = b.create_replicates(&ctx->c, store, {all}, pm, ncat, alpha);
            if (!rc) rc = b.search(true, spr_full, true, eps, &main_lnl);
            if (rc) { b.destroy(); return rc; }
            main_tree = b.genes[0].tree; main_names = b.genes[0].aln.names; main_alpha = b.genes[0].alpha; main_npat = b.genes[0].aln.npat; main_nsites = b.genes[0].aln.nsites;
            b.destroy();
        }
        // support trees: one device batch, or consecutive sub-batches when the replicates' CLV arenas do not fit in free
        // HBM together (a replicate of 100 taxa x 10^5 patterns needs ~19 GB in search mode)
        std::vector<Tree> sup((size_t)reps_here);
        std::string sup_txt;
        if (reps_here > 0) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)1 << 40;
            size_t budget = (size_t)(0.85 * (double)(free_b + ctx->c.arena_cache_bytes));
            if (const char *e = std::getenv("PML_HBM_BUDGET_MB")) budget = (size_t)std::atoll(e) << 20;   // test hook
            auto rep_bytes = [&](const std::vector<int> &sel) {
                size_t pats = 0; std::set<std::string> taxa;
                for (int g : sel) { pats += (size_t)store.items[g].aln.npat; taxa.insert(store.items[g].aln.names.begin(), store.items[g].aln.names.end()); }
                const size_t mp = (pats + 31) / 32 * 32, nt = std::max<size_t>(taxa.size(), 3), slots = 3 * (nt - 2) + NSCRATCH + MAXTAIL;
                return slots * CLV_ROWS * ((mp + 127) / 128 * 128) * 8 + slots * mp * 4 + nt * mp + 64 * mp + (1 << 16);
            };
            for (int begin = 0; begin < reps_here;) {
                size_t used = 0; int end = begin;
                while (end < reps_here) { const size_t need = rep_bytes(rep[end]); if (end > begin && used + need > budget) break; used += need; ++end; }
                std::vector<std::vector<int>> part(rep.begin() + begin, rep.begin() + end);
                Batch b; int rc = b.create_replicates(&ctx->c, store, part, pm, ncat, alpha);
                std::vector<double> l(end - begin);
                if (!rc) rc = b.search(true, 0, true, eps, l.data());
                if (rc) { b.destroy(); return rc; }
                for (int r = begin; r < end; ++r) {
                    const Gene &G = b.genes[r - begin];
                    const std::string nw = G.tree.newick(G.aln.names, 6);
                    sup_txt += nw; sup_txt += '\n';
                    // a replicate may lack taxa that occur only in unselected genes: such trees cannot
                    // contain the main tree's bipartitions and are counted as not supporting
                    if (srank != 0) continue;
                    if (G.aln.names == main_names) sup[r] = G.tree;
                    else { std::string e2; Tree t; if (Tree::parse(nw.c_str(), main_names, t, e2)) sup[r] = t; else sup[r] = Tree(); }
                }
                b.destroy();
                begin = end;
            }
        }
        std::vector<Tree> usable;
        for (auto &t : sup) if (t.ntax == main_tree.ntax) usable.push_back(t);
        const std::string out = srank == 0 ? main_tree.newick_labeled(main_names, 6, support_counts(main_tree, usable)) : std::string();
        main_out->lnl = main_lnl; main_out->alpha = main_alpha; main_out->tree_length = main_tree.length();
        main_out->npatterns = main_npat; main_out->nsites = main_nsites;
        if (srank == 0) main_out->newick = dup_cstr(out);
        if (support_out) *support_out = dup_cstr(sup_txt);
        if (srank == 0 && !main_out->newick) return ctx->c.fail(PML_ENOMEM, "host allocation failed");
    } catch (const std::bad_alloc &) { return ctx->c.fail(PML_ENOMEM, "host allocation failed"); }
    catch (const std::exception &e) { return ctx->c.fail(PML_EINVAL, e.what()); }
    return PML_OK;
}

// Non-parametric bootstrap (`raxmlHPC -f a -x seed -N reps`, RAxMLRunner.java:115-132 with reps > 0): best ML tree
// of the alignment (NNI + lazy SPR) and `reps` trees of column-resampled alignments (NNI), searched as ONE device
// batch; the best tree carries the percentage of replicates containing each bipartition (RAxML_bipartitions.<n>).
// RAxML's "rapid" heuristics (CAT approximation, tree reuse between replicates) are not restated: each replicate
// gets a full search, which is what those heuristics approximate.
extern "C" int pml_bootstrap(pml_ctx *ctx, const pml_alignment *aln, const pml_model *model, int reps, unsigned long long seed,
                             int spr_radius_best, double epsilon, pml_result *best_out, char **replicate_newicks_out) {
    if (!ctx || !aln || !best_out || reps < 0 || aln->ntax < 3 || aln->nsites < 1) return PML_EINVAL;
    pml_fpguard fpg;
    std::memset(best_out, 0, sizeof *best_out);
    if (replicate_newicks_out) *replicate_newicks_out = nullptr;
    const double eps = epsilon > 0 ? epsilon : 1e-3;
    std::lock_guard<std::mutex> lk(ctx->c.mu);
    pml_drop_worker_caches(ctx);
    try {
        const int ncat = model ? model->ncat : 4, pm = model ? model->pi_mode : 0;
        const double alpha = model ? model->alpha : 1.0;
        const int n = aln->ntax, L = aln->nsites;
        Tree best; std::vector<std::string> names; double lnl = 0, al = alpha; int npat = 0;
        {
            pml_alignment_view v{n, L, aln->names, aln->rows};
            Batch b; int rc = b.create(&ctx->c, 1, &v, nullptr, pm, ncat, alpha, false);
            if (!rc) rc = b.search(true, spr_radius_best, true, eps, &lnl);
            if (rc) { b.destroy(); return rc; }
            best = b.genes[0].tree; names = b.genes[0].aln.names; al = b.genes[0].alpha; npat = b.genes[0].aln.npat;
            b.destroy();
        }
        std::vector<Tree> trees((size_t)reps); std::string txt;
        if (reps > 0) {
            std::mt19937_64 rng(seed);
            std::vector<std::vector<std::string>> rows((size_t)reps, std::vector<std::string>((size_t)n, std::string((size_t)L, '-')));
            std::vector<int> col((size_t)L);
            for (int r = 0; r < reps; ++r) {
                for (int s = 0; s < L; ++s) col[s] = (int)(rng() % (uint64_t)L);
                for (int i = 0; i < n; ++i) { const char *src = aln->rows[i]; std::string &dst = rows[r][i]; for (int s = 0; s < L; ++s) dst[s] = src[col[s]]; }
            }
            std::vector<std::vector<const char *>> rps((size_t)reps);
            std::vector<pml_alignment_view> vs((size_t)reps);
            for (int r = 0; r < reps; ++r) { for (auto &x : rows[r]) rps[r].push_back(x.c_str()); vs[r] = pml_alignment_view{n, L, aln->names, rps[r].data()}; }
            // one device batch, or consecutive sub-batches when the replicates do not fit in free HBM together
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) free_b = (size_t)1 << 40;
            size_t budget = (size_t)(0.85 * (double)(free_b + ctx->c.arena_cache_bytes));
            if (const char *e = std::getenv("PML_HBM_BUDGET_MB")) budget = (size_t)std::atoll(e) << 20;   // test hook
            const size_t mp = ((size_t)L + 31) / 32 * 32, slots = 3 * ((size_t)n - 2) + NSCRATCH + MAXTAIL;
            const size_t per_rep = slots * CLV_ROWS * ((mp + 127) / 128 * 128) * 8 + slots * mp * 4 + (size_t)n * mp + 64 * mp + (1 << 16);
            const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)reps, budget / per_rep));
            for (int begin = 0; begin < reps; begin += chunk) {
                const int m = std::min(chunk, reps - begin);
                Batch b; int rc = b.create(&ctx->c, m, vs.data() + begin, nullptr, pm, ncat, alpha, false);
                std::vector<double> l((size_t)m);
                if (!rc) rc = b.search(true, 0, true, eps, l.data());
                if (rc) { b.destroy(); return rc; }
                for (int r = 0; r < m; ++r) { trees[begin + r] = b.genes[r].tree; txt += b.genes[r].tree.newick(names, 6); txt += '\n'; }
                b.destroy();
            }
        }
        auto counts = support_counts(best, trees);
        if (reps > 0) for (auto &row : counts) for (int &c : row) if (c >= 0) c = (int)(0.5 + 100.0 * c / reps);   // percent, as RAxML_bipartitions
        best_out->lnl = lnl; best_out->alpha = al; best_out->tree_length = best.length(); best_out->npatterns = npat; best_out->nsites = L;
        best_out->newick = dup_cstr(reps > 0 ? best.newick_labeled(names, 6, counts) : best.newick(names, 6));
        if (replicate_newicks_out) *replicate_newicks_out = dup_cstr(txt);
        if (!best_out->newick) return ctx->c.fail(PML_ENOMEM, "host allocation failed");
    } catch (const std::bad_alloc &) { return ctx->c.fail(PML_ENOMEM, "host allocation failed"); }
    catch (const std::exception &e) { return ctx->c.fail(PML_EINVAL, e.what()); }
    return PML_OK;
}
