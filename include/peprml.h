/*
 * peprml.h -- C ABI of libpeprml.so, the MI355X-native maximum-likelihood tree engine that
 * replaces the external-process calls on PEPR's tree-building path.
 *
 * Every entry point names the reference interface it replaces (paths relative to the PEPR
 * repository, src/edu/vt/vbi/ci/ abbreviated as .../):
 *
 *   pml_score      <- .../pepr/tree/RAxMLRunner.java:162-213   runRaxmlPerSiteLL():
 *                        `raxmlHPC -f g -m PROTGAMMAWAG -z trees -s aln` -> RAxML_perSiteLLs.<run>
 *   pml_optimize   <- .../pepr/tree/FastTreeRunner.java:142-199 getRaxmlBranchLengths() and
 *                     .../pepr/tree/RAxMLRunner.java:253-272:  `raxmlHPC -f e -t tree` -> RAxML_result.<run>
 *   pml_search     <- .../pepr/tree/RAxMLRunner.java:79-152     run():  `raxmlHPC -f d -m PROTGAMMAWAG`
 *                     .../pepr/tree/FastTreeRunner.java:38-135  run():  `FastTree_WAG -gamma -nosupport`
 *   pml_*_batch    <- .../pepr/tree/pipeline/PhylogenomicPipeline2.java:1587-1633
 *                        GeneSubsetTreeRunnable.run(): the data-parallel loop of independent tree builds
 *   pml_rf_distance<- .../pepr/tree/AdvancedTree.java:1460-1491 (Robinson-Foulds used for acceptance)
 *   pml_jackknife  <- .../pepr/tree/pipeline/PhylogenomicPipeline2.java:994-1126
 *                        buildConcatenatedTreeWithGeneWiseJackKnifeSupport(): full tree + N support trees on
 *                        random gene subsets (:959-977, 1227-1275, 1587-1633) + support counts, ONE call
 *   pml_parsimony  <- .../pepr/tree/RAxMLRunner.java:215-251  runRaxmlParsimonyWithBranchLengths():
 *                        `raxmlHPC -f d -y` -> RAxML_parsimonyTree.<run> (topology only); the caller then
 *                        runs pml_optimize on it, as the reference runs `-f e -t` (:253-272)
 *   pml_bootstrap  <- .../pepr/tree/RAxMLRunner.java:115-132,302-318 (`-f a -x -N`, bootstrapReps > 0)
 *   pml_sh_support <- .../pepr/tree/FastTreeRunner.java:67-70 (`FastTree_WAG -gamma` without -nosupport)
 *   pml_gamma20    <- .../pepr/tree/FastTreeRunner.java:67-70 (`-gamma`: the "Gamma(20) LogLk ... alpha ... rescaling
 *                        lengths" step that ends every FastTree_WAG run PEPR makes; its printed tree is what PEPR parses)
 *   pml_concatenate<- .../pepr/alignment/MSAConcatenator.java:78-189 (sorted taxon union, '?' padding)
 *   pml_refine_next<- .../pepr/tree/PhylogeneticTreeRefiner.java:298-359 + AdvancedTree.java:1061-1098
 *   pml_support_tree<- .../pepr/tree/TreeSupportDecorator.java:86-163 addSupportValues(): integer
 *                        bipartition counts of the support trees written as node labels of the main tree
 *
 * Conventions: caller owns inputs (nothing is retained after return);
 * the library allocates results, the caller releases them with pml_result_free(); no files, no
 * cwd dependence, no stdout/stderr output; every function returns 0 or a negative PML_E* code and
 * never aborts or throws.  A context may be used from several threads: batch calls serialise on it,
 * concurrent single-gene calls are coalesced into one device batch (pml_coalescing_stats).
 * The alignment rows are exactly SequenceAlignment.alignedSequenceChars
 * (.../pepr/alignment/SequenceAlignment.java:61): one char row per taxon, 20 amino-acid letters,
 * '-' gap, '?' absent gene, anything else unknown (treated like a gap; B/Z are N|D and Q|E).
 *
 * There is NO CPU fallback: without a HIP device pml_create() fails with PML_ENODEVICE.
 */
#ifndef PEPRML_H
#define PEPRML_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PML_OK          0
#define PML_EINVAL     -1   /* bad argument */
#define PML_EPARSE     -2   /* Newick / alignment parse error (message in pml_last_error) */
#define PML_ENODEVICE  -3   /* no usable HIP device */
#define PML_ENOMEM     -4   /* host or device allocation failed */
#define PML_EDEVICE    -5   /* HIP runtime error */
#define PML_ENOTFOUND  -6

typedef struct pml_ctx pml_ctx;
typedef struct pml_batch pml_batch;

typedef struct {
    int device;              /* HIP device ordinal (rank-local GPU) */
    int profile;             /* 1: record HIP events around kernels (pml_kernel_stats) */
    size_t arena_bytes;      /* > 0: reserve this much HBM at pml_create for batch arenas (a fresh allocation is
                              * zero-filled by the driver at ~40 GB/s); 0 = allocate per batch, keep the last one */
} pml_config;

typedef struct {
    int ntax;
    int nsites;
    const char *const *names;   /* ntax taxon names (Newick leaf labels) */
    const char *const *rows;    /* ntax rows of nsites chars */
} pml_alignment;

/* PML_PI_EMPIRICAL = RAxML's "F" models (PROTGAMMAWAGF, one of the names -matrix_eval passes, PhylogenomicPipeline2.java:260-284;
 * RAxMLRunner's own default is an F model, RAxMLRunner.java:46): WAG exchangeabilities with frequencies counted from each
 * gene's alignment (eight sweeps of proportional counting, ambiguity codes spread over their states, floor 0.001) -- a
 * per-gene eigen-system.  Built for score / optimize / search / per-site calls and resident batches; the jackknife's
 * device-gathered replicates refuse it. */
enum { PML_PI_RAXML_3DP = 0, PML_PI_WAG_FULL = 1, PML_PI_EMPIRICAL = 2 };

typedef struct {
    int ncat;                /* Gamma categories (4 = RAxML PROTGAMMA; 1 = no rate heterogeneity) */
    double alpha;            /* Gamma shape (start value when optimised) */
    int pi_mode;             /* PML_PI_RAXML_3DP (RAxML 7.2.5 PROTGAMMAWAG), PML_PI_WAG_FULL (FastTree_WAG) or PML_PI_EMPIRICAL (PROTGAMMAWAGF) */
} pml_model;

typedef struct {
    int optimize_alpha;      /* 1: Brent on alpha */
    int nni;                 /* 1: NNI hill climbing */
    int spr_radius;          /* >0: SPR rounds with this rearrangement radius */
    double epsilon;          /* stop when a round gains less than this many lnL units */
    unsigned seed;           /* 0: NJ start tree (deterministic); != 0: randomised stepwise-addition parsimony start with
                              * this seed, as `raxmlHPC -f d -p seed` starts (genes that come with a start tree keep it) */
    /* topological constraints, FastTree's -constraints semantics as PEPR produces them
     * (FastTreeRunner.getFastTreeConstraintsForTree, FastTreeRunner.java:243-273): a 0/1/- matrix,
     * one row per named taxon, one column per constrained split ('-' = taxon free in that column).
     * The result tree displays every non-trivial column; taxa absent from the matrix are free.
     * nconstraints = 0: unconstrained. */
    int nconstraints;
    int constraint_ntax;
    const char *const *constraint_names;
    const char *const *constraint_rows;
} pml_search_opts;

typedef struct {
    int status;              /* PML_OK or error for this gene */
    double lnl;              /* log likelihood */
    double alpha;            /* Gamma shape used / found */
    double tree_length;      /* sum of branch lengths */
    int npatterns;           /* alignment patterns after compression */
    int nsites;
    char *newick;            /* resulting tree, NUL-terminated, RAxML-style unrooted (may be NULL) */
    double *site_lnl;        /* per-site lnL in alignment column order (only if requested) */
} pml_result;

#define PML_WANT_SITE_LNL 1   /* flags for pml_score */

/* lifecycle */
int  pml_create(const pml_config *cfg, pml_ctx **out);
void pml_destroy(pml_ctx *ctx);
const char *pml_strerror(int code);
const char *pml_last_error(pml_ctx *ctx);     /* detail of the last failure on this context */
const char *pml_version(void);

/* one-shot calls (one gene) */
int pml_score(pml_ctx *ctx, const pml_alignment *aln, const char *newick, const pml_model *model,
              int flags, pml_result *out);
int pml_optimize(pml_ctx *ctx, const pml_alignment *aln, const char *newick, const pml_model *model,
                 const pml_search_opts *opts, pml_result *out);
int pml_search(pml_ctx *ctx, const pml_alignment *aln, const char *start_newick /* NULL = NJ */,
               const pml_model *model, const pml_search_opts *opts, pml_result *out);

/* gene-batched calls: n independent (alignment, tree) units evaluated together on the device */
int pml_score_batch(pml_ctx *ctx, int n, const pml_alignment *alns, const char *const *newicks,
                    const pml_model *model, int flags, pml_result *out);
int pml_optimize_batch(pml_ctx *ctx, int n, const pml_alignment *alns, const char *const *newicks,
                       const pml_model *model, const pml_search_opts *opts, pml_result *out);
int pml_search_batch(pml_ctx *ctx, int n, const pml_alignment *alns, const char *const *start_newicks,
                     const pml_model *model, const pml_search_opts *opts, pml_result *out);
void pml_result_free(pml_result *r);

/* resident batches: encode + upload once, then evaluate repeatedly with inputs in HBM */
int  pml_batch_create(pml_ctx *ctx, int n, const pml_alignment *alns, const char *const *newicks,
                      const pml_model *model, pml_batch **out);
void pml_batch_destroy(pml_batch *b);
int  pml_batch_size(const pml_batch *b);
int  pml_batch_npatterns(const pml_batch *b, int gene);
/* full post-order CLV pass + root evaluation for every gene (all CLVs recomputed) */
int  pml_batch_score(pml_batch *b, double *lnl_out /* n */);
/* the same pass with every CLV WRITTEN to HBM (a child its parent consumes next is still read from registers): the whole-tree
 * traversal a search runs after a topology or alpha change, where later partial traversals read the CLVs back; same lnL bits */
int  pml_batch_score_stored(pml_batch *b, double *lnl_out /* n */);
int  pml_batch_site_lnl(pml_batch *b, int gene, double *site_lnl /* nsites */);
int  pml_batch_set_alpha(pml_batch *b, int gene /* -1 = all */, double alpha);
int  pml_batch_optimize(pml_batch *b, const pml_search_opts *opts, double *lnl_out, double *alpha_out);
int  pml_batch_search(pml_batch *b, const pml_search_opts *opts, double *lnl_out, double *alpha_out);
int  pml_batch_newick(pml_batch *b, int gene, int digits, char **out /* free with pml_free */);
/* d lnL/dt, d2 lnL/dt2 for the branch above each gene's taxon 0 (test hook for the Newton kernel) */
int  pml_batch_root_derivs(pml_batch *b, double *lnl, double *d1, double *d2);
void pml_free(void *p);

/* tree utilities (host) */
int pml_rf_distance(const char *newick_a, const char *newick_b, int *rf_out);
/* main tree with, on every internal branch, the number of support trees containing its bipartition
 * as an integer node label `)87:0.1`; *out is freed with pml_free */
int pml_support_tree(const char *main_newick, int ntrees, const char *const *support_newicks, int digits, char **out);

/* gene-wise jackknife (the data-parallel loop of the reference, one call):
 *   full tree  = search on the concatenation of ALL genes (NNI + SPR radius spr_radius_full),
 *   supports   = `reps` searches (NJ + NNI), each on the concatenation of `subset_size` genes drawn
 *                without replacement (0 = ngenes/2, PhylogenomicPipeline2.java:1599-1617),
 *   result     = full tree whose internal branches carry the number of support trees containing them.
 * Genes may cover different taxon subsets: the concatenation uses the sorted union of taxon names
 * and pads absent genes with '?' (MSAConcatenator.java:118-120,164-170).  The reference draws the
 * subsets from an unseeded java.util.Random; here the draw is seeded (deterministic). */
typedef struct {
    int reps;                    /* support trees (PEPR default 100) */
    int subset_size;             /* genes per replicate; 0 = ngenes / 2 */
    unsigned long long seed;
    int spr_radius_full;         /* SPR radius for the full tree (0 = NNI only) */
    double epsilon;              /* search epsilon (0 = 1e-3) */
    /* multi-GPU: every rank passes the same genes/seed; this call searches only replicates r with
     * r % shard_world == shard_rank and the full tree only on shard_rank 0 (elsewhere main_out->newick is
     * NULL and carries no supports); the caller gathers the support trees and decorates with
     * pml_support_tree (pepr_amd/distributed.py: jackknife()).  0,0 or world <= 1 = everything here. */
    int shard_rank, shard_world;
} pml_jackknife_opts;
int pml_jackknife(pml_ctx *ctx, int ngenes, const pml_alignment *genes, const pml_model *model,
                  const pml_jackknife_opts *opts, pml_result *main_out /* newick carries the supports */,
                  char **support_newicks_out /* optional: reps lines, '\n'-separated; pml_free */);
/* host-only: the gene subsets pml_jackknife draws for (ngenes, reps, subset_size, seed) -- replicate r = sel_out[r*k .. r*k+k),
 * ascending gene indices, k = the return value (subset_size, or ngenes/2 when 0; < 0 = error).  The reference draws them with
 * RandomSetUtils.getRandomSet (.../pepr/util/RandomSetUtils.java:9-35, unseeded java.util.Random); callers that need to know
 * which genes a support tree was built from (reports, the oracle-side parity test) get the seeded draw here. */
int pml_jackknife_draw(int ngenes, int reps, int subset_size, unsigned long long seed, int *sel_out /* reps x k */);
/* test hook for the device-side concatenation (MSAConcatenator.java:78-189): encodes the genes into HBM, gathers
 * the selection `sel` (NULL = all) on the device exactly as pml_jackknife does for a replicate, and reads the result back:
 * codes_out[ntax x mpad] (0..19 = ARNDCQEGHILKMFPSTWYV, 20 = B, 21 = Z, 22 = gap/unknown), weights_out[mpad] (0 = padding),
 * names_out = the sorted taxon union, one per line.  The three buffers are released with pml_free. */
int pml_debug_gather(pml_ctx *ctx, int ngenes, const pml_alignment *genes, int nsel, const int *sel, int *ntax_out,
                     int *npat_out, int *mpad_out, unsigned char **codes_out, double **weights_out, char **names_out);
/* host-only: the refinement loop's support queries on a rooted Newick with support labels (")95:0.1" or
 * ":0.1[95]"; missing = 100, fractions are x100) -- PhylogeneticTreeRefiner.java:298-359 getNextIndexToRefine,
 * AdvancedTree.java:1061-1098 getMeanDescendantSupportValues.  *ingroup_out = comma-joined sorted leaf names of
 * the next clade to refine, NULL if none; done[] = clades already refined in the same format (the caller's
 * refinedSubsets).  mean_out (optional) = floor(mean descendant support) per node in order of appearance. */
int pml_refine_next(const char *supported_newick, int cutoff, int ndone, const char *const *done,
                    char **ingroup_out, int *nnodes_out, int **mean_out /* pml_free */);
/* SH-like local supports, FastTree's default output when -nosupport is absent (FastTreeRunner.java:67-70: PEPR drops
 * -nosupport when bootstrapReps > 0; AdvancedTree.getBranchSupports :484-506 reads the 0-1 labels x100).  For every
 * internal split of the given tree (lengths and model->alpha as given): the split's arrangement against its two NNI
 * alternatives, `nboot` (FastTree: 1000) resamples of the alignment columns on the device, support = share of
 * resamples that do not overturn the observed advantage (FastTree 2.1 SHSupport / Guindon et al. 2010).
 * out[i].newick carries the supports as inner labels with 3 decimals; lnl = lnL of the tree. */
int pml_sh_support(pml_ctx *ctx, const pml_alignment *aln, const char *newick, const pml_model *model,
                   int nboot, unsigned long long seed, pml_result *out);
int pml_sh_support_batch(pml_ctx *ctx, int n, const pml_alignment *alns, const char *const *newicks,
                         const pml_model *model, int nboot, unsigned long long seed, pml_result *out);
/* FastTree's `-gamma` likelihood (FastTreeRunner.java:67-70 passes -gamma on every call; SURVEY 8a-11 vi): the given tree's
 * per-site likelihoods at FastTree's 20 fixed rates 0.05 * 400^(k/19) (five device traversals of four rates), re-weighted
 * by a discretised Gamma(alpha) with mean `rescale`; alpha and rescale are fitted on that table.  out[i].lnl = Gamma20 lnL,
 * out[i].alpha = its alpha, out[i].newick = the tree with every length multiplied by rescale_out[i] (what FastTree prints),
 * out[i].tree_length of that tree.  model->pi_mode selects the frequencies (FastTree_WAG: PML_PI_WAG_FULL); model->alpha
 * and ncat are not used. */
int pml_gamma20(pml_ctx *ctx, const pml_alignment *aln, const char *newick, const pml_model *model, pml_result *out,
                double *rescale_out);
int pml_gamma20_batch(pml_ctx *ctx, int n, const pml_alignment *alns, const char *const *newicks, const pml_model *model,
                      pml_result *out, double *rescale_out /* n, may be NULL */);
/* Maximum-parsimony trees (Fitch lengths on the device): randomised stepwise addition (seed 0 =
 * input order) then SPR hill climbing within spr_radius edges (0 = none; RAxML uses 20).  out[i].newick is
 * topology only; out[i].lnl / alpha / tree_length are 0; mp_length[i] (optional) = weighted Fitch length. */
typedef struct { unsigned seed; int spr_radius; } pml_parsimony_opts;
int pml_parsimony(pml_ctx *ctx, const pml_alignment *aln, const pml_parsimony_opts *opts, pml_result *out, long long *mp_length);
int pml_parsimony_batch(pml_ctx *ctx, int n, const pml_alignment *alns, const pml_parsimony_opts *opts,
                        pml_result *out, long long *mp_length);
/* Non-parametric bootstrap (`raxmlHPC -f a -x seed -N reps`, RAxMLRunner.java:115-132): best ML tree (NNI + SPR)
 * labelled with the percentage of `reps` column-resampled replicate trees (one device batch) containing each
 * bipartition, as RAxML_bipartitions.<run> carries them (read at RAxMLRunner.java:302-318). */
int pml_bootstrap(pml_ctx *ctx, const pml_alignment *aln, const pml_model *model, int reps, unsigned long long seed,
                  int spr_radius_best, double epsilon, pml_result *best_out,
                  char **replicate_newicks_out /* optional: reps lines; pml_free */);
/* host-only: FASTA text (">taxon\nSEQ\n" per taxon, SequenceAlignment.java:405-416) of the
 * concatenation of the selected genes (sel == NULL: all), taxa = sorted union, '?' padding */
int pml_concatenate(int ngenes, const pml_alignment *genes, int nsel, const int *sel, char **fasta_out);

/* Concurrent pml_score / pml_optimize / pml_search calls on one context (PEPR's tree_threads workers) are
 * coalesced into device batches; this reports how many batches ran and how many single calls they carried. */
int pml_coalescing_stats(pml_ctx *ctx, long long *batches, long long *requests);

/* The branch-length Newton kernel splits a request over several workgroups that exchange partial sums.  Its forward progress
 * does not depend on dispatch order (slices are claimed by ticket), and its waits are bounded in wall-clock time: when a wait
 * gives up (a co-tenant of the GPU kept a request's slices apart for > 2 s), the affected requests -- in a chained smoothing
 * pass, the whole pass -- are re-issued through a no-exchange form of the same kernel whose results are bit-identical, and the
 * call succeeds.  This reports how often that happened on the context: give-up events, requests re-issued, launches of the
 * no-exchange form (all 0 on a GPU the process has to itself). */
int pml_newton_fallbacks(pml_ctx *ctx, long long *giveups, long long *reissued, long long *seq_launches);

/* Every computing entry point runs its host-side arithmetic under the DEFAULT floating-point control state (round to
 * nearest, no flush-to-zero / denormals-are-zero) whatever the calling thread -- a JVM worker, a Python thread -- came in
 * with, and restores the caller's state on return: results do not depend on the caller's MXCSR.  Diagnostic: the distinct
 * control states callers entered with (returns their number; values[] receives up to cap of them). */
int pml_debug_fpenv(unsigned *values, int cap);

/* profiling: HIP-event time of device kernels since the last reset (cfg.profile = 1) */
enum { PML_K_PMAT = 0, PML_K_NEWVIEW = 1, PML_K_EVALUATE = 2, PML_K_SUMTABLE = 3, PML_K_NEWTON = 4,
       PML_K_REDUCE = 5,
       PML_K_HOST_BUILD = 6 /* CPU ms building descriptors */, PML_K_HOST_WAIT = 7 /* CPU ms in stream sync */, PML_K_COUNT = 8 };
int pml_kernel_stats(pml_ctx *ctx, int kernel, long long *launches, double *total_ms,
                     double *algo_bytes /* algorithmic bytes moved, SURVEY 8d figures */);
/* algorithmic flops of the launches counted by pml_kernel_stats, SURVEY 8d's per-operation figures (newview inner-inner 6480,
 * tip-inner 3280, tip-tip 80, evaluate 3360 flop per pattern; a sumtable = an inner-inner contraction) */
int pml_kernel_flops(pml_ctx *ctx, int kernel, double *algo_flops);
int pml_kernel_stats_reset(pml_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
