/*
 * pml_oracle.h -- CPU oracle (TEST INFRASTRUCTURE, not product code).
 *
 * Plain-C float64 restatement of the likelihood arithmetic that PEPR obtains from the
 * external programs it spawns (reference call sites:
 *   src/edu/vt/vbi/ci/pepr/tree/RAxMLRunner.java:115-147   raxmlHPC -f d|e|g -m PROTGAMMAWAG
 *   src/edu/vt/vbi/ci/pepr/tree/FastTreeRunner.java:67-94   FastTree_WAG -gamma -nosupport).
 * The arithmetic itself lives in third-party programs whose source is NOT in the reference
 * repository (RAxML 7.2.5, Stamatakis 2009; FastTree 2.1.1, Price et al. 2010); this file
 * restates their *published* algorithm: Felsenstein pruning (Felsenstein 1981) under the WAG
 * model (Whelan & Goldman 2001) with discrete-Gamma rate heterogeneity (Yang 1994, mean of
 * K equal-probability bins), Newton-Raphson branch-length optimisation, Brent alpha
 * optimisation and NNI / SPR hill climbing; Fitch (1971) parsimony with stepwise addition + SPR (`raxmlHPC -y`);
 * SH-like local supports (FastTree `SHSupport`, Guindon et al. 2010).
 *
 * PARITY UNPINNED against the reference binaries: the reference holds no golden vectors or
 * tests for this path and its bundled prebuilt executables may not be
 * executed in this build pipeline.  What IS pinned (tests/test_oracle_*.py):
 *   - WAG constants against the data table the reference ships (tests/golden/wag_constants.json),
 *   - RAxML 7.2.5 conventions recorded in SURVEY.md section 8c (pi rounded to 3 decimals with
 *     pi(I)=0.049, K=4 mean-rate Gamma, all-ones tip vectors for gap/?/X),
 *   - the pruning recursion against brute-force enumeration and scipy.linalg.expm,
 *   - Gamma quantiles/rates against scipy.special and Yang (1994) table values.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code.
 */
#ifndef PML_ORACLE_H
#define PML_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

#define PO_NS 20          /* amino-acid states, order ARNDCQEGHILKMFPSTWYV */
#define PO_MAXCAT 32
#define PO_NCODES 23      /* 20 states + B + Z + gap/unknown */

enum { PO_PI_RAXML3DP = 0, PO_PI_FULL = 1 };

typedef struct {
    double pi[PO_NS];
    double Q[PO_NS][PO_NS];      /* normalised rate matrix, rows sum to 0, 1 subst/site */
    double eval[PO_NS];          /* eigenvalues of Q */
    double U[PO_NS][PO_NS];      /* P(t) = U diag(exp(eval t)) Uinv */
    double Uinv[PO_NS][PO_NS];
} po_model;

typedef struct {
    int ntax, nsites, npat;
    char **names;
    unsigned char *codes;        /* [ntax][npat] code 0..22 */
    int *weight;                 /* [npat] */
    int *site2pat;               /* [nsites] */
} po_aln;

typedef struct {
    int ntax;
    int nnodes;                  /* 2*ntax-2: tips 0..ntax-1, inner ntax..2ntax-3 */
    int (*nbr)[3];               /* neighbour ids (-1 unused) */
    double (*len)[3];            /* branch length to that neighbour */
} po_tree;

/* ---- model ---- */
void po_wag_tables(double S[PO_NS][PO_NS], double pi_full[PO_NS], double pi_3dp[PO_NS]);
void po_model_init(po_model *m, int pi_mode);
/* PROTGAMMAWAGF: WAG exchangeabilities with the given (empirical) frequencies; po_empirical_freqs counts them from an alignment */
void po_model_init_freqs(po_model *m, const double *pi20);
void po_pmatrix(const po_model *m, double t, double P[PO_NS][PO_NS]);
double po_lngamma(double x);
double po_incgamma(double a, double x);             /* regularised lower P(a,x) */
double po_gamma_quantile(double p, double a);       /* x with P(a,x)=p, scale 1 */
void po_gamma_rates(double alpha, int K, int median, double *rates);
unsigned po_code_mask(int code);
int po_char_code(int c);

/* ---- alignment ---- */
po_aln *po_aln_create(int ntax, int nsites, const char *const *names, const char *const *rows,
                      int compress);
void po_aln_free(po_aln *a);
void po_empirical_freqs(const po_aln *a, double *pi20);

/* ---- tree ---- */
po_tree *po_tree_parse(const char *newick, const po_aln *a, char *err, int errlen);
po_tree *po_tree_copy(const po_tree *t);
void po_tree_free(po_tree *t);
/* RAxML-style unrooted newick (trifurcation at the neighbour of taxon 0), %.*f lengths */
char *po_tree_newick(const po_tree *t, const po_aln *a, int digits);
int po_tree_rf(const po_tree *a, const po_tree *b);  /* Robinson-Foulds (symmetric difference / 2) */
double po_tree_length(const po_tree *t);

/* ---- likelihood ---- */
typedef struct po_engine po_engine;
po_engine *po_engine_create(const po_aln *a, const po_model *m, int ncat, double alpha);
void po_engine_free(po_engine *e);
void po_engine_set_alpha(po_engine *e, double alpha);
double po_engine_alpha(const po_engine *e);
/* full lnL of a tree; if pat_lnl != NULL receives per-pattern ln likelihoods (unweighted) */
double po_engine_lnl(po_engine *e, const po_tree *t, double *pat_lnl);
/* per-site (alignment column order) lnL; returns total */
double po_engine_site_lnl(po_engine *e, const po_tree *t, double *site_lnl);
/* optimise all branch lengths (+alpha if opt_alpha) until lnL gain < eps; returns lnL */
double po_engine_optimize(po_engine *e, po_tree *t, int opt_alpha, double eps);
/* d lnL/dt and d2 lnL/dt2 of branch (u,v) at its current length */
void po_engine_branch_derivs(po_engine *e, const po_tree *t, int u, int v, double *lnl,
                             double *d1, double *d2);
/* tree search: NJ start (or `start` if non-NULL), NNI (+SPR if spr_radius>0); returns lnL */
double po_engine_search(po_engine *e, po_tree **t_inout, int spr_radius, double eps);
po_tree *po_nj_tree(const po_aln *a);
/* topological constraints (FastTree -constraints matrix: names, rows of '0' '1' '-', one column per split; HARD here -- see
 * pml_oracle.c): po_engine_search and its NNI / SPR rounds honour them; po_nj_tree_constrained = the start tree used when
 * the given / NJ start tree violates one */
void po_engine_set_constraints(po_engine *e, int ncons, int ntax_c, const char *const *names, const char *const *rows);
po_tree *po_nj_tree_constrained(const po_engine *e);
int po_engine_tree_displays(const po_engine *e, const po_tree *t);
/* SH-like local supports (FastTree SHSupport): support[] in internal-edge order (u ascending, slot ascending, v > u inner);
 * returns the number of edges written */
int po_engine_sh_support(po_engine *e, const po_tree *t, int nboot, unsigned long long seed, double *support);

/* ---- FastTree's `-gamma` step: Gamma20 lnL of a given tree, alpha and length rescale fitted on the per-pattern x rate table
 * (spec in pml_oracle.c); table_out (optional) = npat x 20 per-pattern ln likelihoods, pattern-major ---- */
void po_g20_rates(double *rates20);
void po_g20_weights(double alpha, double mult, double *w20);
double po_gamma20(const po_aln *a, const po_model *m, const po_tree *t, double *alpha_out, double *rescale_out, double *table_out);

/* ---- parsimony (`raxmlHPC -y` start tree; spec in pml_oracle.c) ---- */
long long po_parsimony_length(const po_aln *a, const po_tree *t);     /* weighted Fitch length */
po_tree *po_parsimony_tree(const po_aln *a, unsigned seed, int radius, long long *length_out, int *moves_out);

/* brute force (tiny trees only): sums over all inner-state assignments; independent of pruning */
double po_bruteforce_lnl(const po_aln *a, const po_model *m, int ncat, double alpha,
                         const po_tree *t);

#ifdef __cplusplus
}
#endif
#endif
