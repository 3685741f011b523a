/*
 * pml_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see pml_oracle.h header).
 *
 * PARITY UNPINNED vs the reference's bundled FastTree_WAG / raxmlHPC executables (they may not
 * be executed here and the reference holds no golden vectors); pinned by independent checks
 * listed in pml_oracle.h.  Reference call sites this restates the arithmetic behind:
 *   RAxMLRunner.java:115-147 (-f d / -f e / -f g, -m PROTGAMMAWAG), FastTreeRunner.java:67-94.
 */
#include "pml_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

#define PO_TMIN 1.0e-6     /* RAxML 7.2.5: zmax = 1-1e-6  -> t >= ~1e-6 */
#define PO_TMAX 34.5       /* RAxML 7.2.5: zmin = 1e-15   -> t <= 34.5 */
#define PO_ALPHA_MIN 0.02
#define PO_ALPHA_MAX 1000.0
#define PO_SCALE_THRESH 1.1579208923731620e-77  /* 2^-256 (RAxML "minlikelihood") */
#define PO_SCALE_MULT   8.6361685550944446e+76  /* 2^255? no: set at init to 2^256 */
#define PO_LOG_2_256    177.44567822334599      /* 256 ln 2 */

/* ------------------------------------------------------------------------------------------
 * WAG constants (Whelan & Goldman 2001, PAML wag.dat lower triangle; order ARNDCQEGHILKMFPSTWYV)
 * ---------------------------------------------------------------------------------------- */
static const double WAG_LOWER[190] = {
0.551571,
0.509848, 0.635346,
0.738998, 0.147304, 5.429420,
1.027040, 0.528191, 0.265256, 0.0302949,
0.908598, 3.035500, 1.543640, 0.616783, 0.0988179,
1.582850, 0.439157, 0.947198, 6.174160, 0.021352, 5.469470,
1.416720, 0.584665, 1.125560, 0.865584, 0.306674, 0.330052, 0.567717,
0.316954, 2.137150, 3.956290, 0.930676, 0.248972, 4.294110, 0.570025, 0.249410,
0.193335, 0.186979, 0.554236, 0.039437, 0.170135, 0.113917, 0.127395, 0.0304501, 0.138190,
0.397915, 0.497671, 0.131528, 0.0848047, 0.384287, 0.869489, 0.154263, 0.0613037, 0.499462, 3.170970,
0.906265, 5.351420, 3.012010, 0.479855, 0.0740339, 3.894900, 2.584430, 0.373558, 0.890432, 0.323832, 0.257555,
0.893496, 0.683162, 0.198221, 0.103754, 0.390482, 1.545260, 0.315124, 0.174100, 0.404141, 4.257460, 4.854020, 0.934276,
0.210494, 0.102711, 0.0961621, 0.0467304, 0.398020, 0.0999208, 0.0811339, 0.049931, 0.679371, 1.059470, 2.115170, 0.088836, 1.190630,
1.438550, 0.679489, 0.195081, 0.423984, 0.109404, 0.933372, 0.682355, 0.243570, 0.696198, 0.0999288, 0.415844, 0.556896, 0.171329, 0.161444,
3.370790, 1.224190, 3.974230, 1.071760, 1.407660, 1.028870, 0.704939, 1.341820, 0.740169, 0.319440, 0.344739, 0.967130, 0.493905, 0.545931, 1.613280,
2.121110, 0.554413, 2.030060, 0.374866, 0.512984, 0.857928, 0.822765, 0.225833, 0.473307, 1.458160, 0.326622, 1.386980, 1.516120, 0.171903, 0.795384, 4.378020,
0.113133, 1.163920, 0.0719167, 0.129767, 0.717070, 0.215737, 0.156557, 0.336983, 0.262569, 0.212483, 0.665309, 0.137505, 0.515706, 1.529640, 0.139405, 0.523742, 0.110864,
0.240735, 0.381533, 1.086000, 0.325711, 0.543833, 0.227710, 0.196303, 0.103604, 3.873440, 0.420170, 0.398618, 0.133264, 0.428437, 6.454280, 0.216046, 0.786993, 0.291148, 2.485390,
2.006010, 0.251849, 0.196246, 0.152335, 1.002140, 0.301281, 0.588731, 0.187247, 0.118358, 7.821300, 1.800340, 0.305434, 2.058450, 0.649892, 0.314887, 0.232739, 1.388230, 0.365369, 0.314730
};
static const double WAG_PI_FULL[20] = {
0.0866279, 0.043972, 0.0390894, 0.0570451, 0.0193078, 0.0367281, 0.0580589, 0.0832518, 0.0244313,
0.048466, 0.086209, 0.0620286, 0.0195027, 0.0384319, 0.0457631, 0.0695179, 0.0610127, 0.0143859,
0.0352742, 0.0708956 };
/* RAxML 7.2.5 PROTGAMMAWAG base frequencies (3 decimals, sum 1.000; pi(I)=0.049) */
static const double WAG_PI_3DP[20] = {
0.087, 0.044, 0.039, 0.057, 0.019, 0.037, 0.058, 0.083, 0.024, 0.049, 0.086, 0.062, 0.020, 0.038,
0.046, 0.070, 0.061, 0.014, 0.035, 0.071 };

void po_wag_tables(double S[PO_NS][PO_NS], double pi_full[PO_NS], double pi_3dp[PO_NS]) {
    int i, j, k = 0;
    for (i = 0; i < 20; i++) S[i][i] = 0.0;
    for (i = 1; i < 20; i++)
        for (j = 0; j < i; j++) { S[i][j] = S[j][i] = WAG_LOWER[k++]; }
    for (i = 0; i < 20; i++) { pi_full[i] = WAG_PI_FULL[i]; pi_3dp[i] = WAG_PI_3DP[i]; }
}

/* cyclic Jacobi eigen-decomposition of a symmetric 20x20 matrix: A = V diag(d) V^T */
static void jacobi20(double A[PO_NS][PO_NS], double d[PO_NS], double V[PO_NS][PO_NS]) {
    int n = PO_NS, i, j, p, q, sweep;
    for (i = 0; i < n; i++) for (j = 0; j < n; j++) V[i][j] = (i == j);
    for (sweep = 0; sweep < 100; sweep++) {
        double off = 0.0;
        for (p = 0; p < n; p++) for (q = p + 1; q < n; q++) off += A[p][q] * A[p][q];
        if (off < 1e-40) break;
        for (p = 0; p < n; p++)
            for (q = p + 1; q < n; q++) {
                if (fabs(A[p][q]) < 1e-300) continue;
                double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                for (i = 0; i < n; i++) {
                    double aip = A[i][p], aiq = A[i][q];
                    A[i][p] = c * aip - s * aiq; A[i][q] = s * aip + c * aiq;
                }
                for (i = 0; i < n; i++) {
                    double api = A[p][i], aqi = A[q][i];
                    A[p][i] = c * api - s * aqi; A[q][i] = s * api + c * aqi;
                }
                for (i = 0; i < n; i++) {
                    double vip = V[i][p], viq = V[i][q];
                    V[i][p] = c * vip - s * viq; V[i][q] = s * vip + c * viq;
                }
            }
    }
    for (i = 0; i < n; i++) d[i] = A[i][i];
    /* sort descending (eigenvalue 0 first) for determinism */
    for (i = 0; i < n; i++) {
        int best = i;
        for (j = i + 1; j < n; j++) if (d[j] > d[best]) best = j;
        if (best != i) {
            double tmp = d[i]; d[i] = d[best]; d[best] = tmp;
            for (j = 0; j < n; j++) { tmp = V[j][i]; V[j][i] = V[j][best]; V[j][best] = tmp; }
        }
    }
}

/* PROTGAMMAWAGF (RAxMLRunner.java:46 default PROTGAMMALGF, PhylogenomicPipeline2.java:260-284 -matrix_eval names): the "F"
 * models replace the matrix's own frequencies by EMPIRICAL ones counted from the alignment.  Restated from RAxML's published
 * scheme (parity unpinned: no source, no fixture in the reference): start from 1/20 each; eight sweeps in which every
 * character of every taxon spreads its pattern weight over the states it allows in proportion to the current frequencies
 * (an unambiguous residue counts 1 for its state, B = N|D and Z = Q|E share theirs, gap / ? / X spread over all 20, i.e. add
 * nothing but a multiple of the current vector); normalise after every sweep; finally states rarer than 0.001 are lifted to
 * 0.001 and the others scaled down so that the sum stays 1 (repeated until none is below). */
void po_empirical_freqs(const po_aln *a, double *pi) {
    double f[20], acc[20];
    for (int l = 0; l < 20; l++) f[l] = 0.05;
    for (int sweep = 0; sweep < 8; sweep++) {
        for (int l = 0; l < 20; l++) acc[l] = 0.0;
        for (int i = 0; i < a->ntax; i++) for (int p = 0; p < a->npat; p++) {
            const unsigned mk = po_code_mask(a->codes[(size_t)i * a->npat + p]);
            double sum = 0.0;
            for (int l = 0; l < 20; l++) if ((mk >> l) & 1) sum += f[l];
            const double wj = (double)a->weight[p] / sum;
            for (int l = 0; l < 20; l++) if ((mk >> l) & 1) acc[l] += wj * f[l];
        }
        double tot = 0.0;
        for (int l = 0; l < 20; l++) tot += acc[l];
        for (int l = 0; l < 20; l++) f[l] = acc[l] / tot;
    }
    for (int round = 0; round < 100; round++) {
        double lift = 0.0, big = 0.0; int low = 0;
        for (int l = 0; l < 20; l++) { if (f[l] < 0.001) { lift += 0.001 - f[l]; low++; } else big += f[l]; }
        if (!low) break;
        for (int l = 0; l < 20; l++) f[l] = f[l] < 0.001 ? 0.001 : f[l] * (1.0 - lift / big);
    }
    for (int l = 0; l < 20; l++) pi[l] = f[l];
}

static void model_from_pi(po_model *m, double S[PO_NS][PO_NS]);
void po_model_init_freqs(po_model *m, const double *pi) {
    double S[PO_NS][PO_NS], pf[PO_NS], p3[PO_NS];
    po_wag_tables(S, pf, p3);
    double sum = 0;
    for (int i = 0; i < 20; i++) { m->pi[i] = pi[i]; sum += pi[i]; }
    for (int i = 0; i < 20; i++) m->pi[i] /= sum;
    model_from_pi(m, S);
}
void po_model_init(po_model *m, int pi_mode) {
    double S[PO_NS][PO_NS], pf[PO_NS], p3[PO_NS];
    int i;
    po_wag_tables(S, pf, p3);
    double sum = 0;
    for (i = 0; i < 20; i++) { m->pi[i] = (pi_mode == PO_PI_FULL) ? pf[i] : p3[i]; sum += m->pi[i]; }
    for (i = 0; i < 20; i++) m->pi[i] /= sum;
    model_from_pi(m, S);
}
/* Q = S diag(pi) normalised to one substitution per site, symmetrised eigen-decomposition (m->pi is set and sums to 1) */
static void model_from_pi(po_model *m, double S[PO_NS][PO_NS]) {
    double B[PO_NS][PO_NS], V[PO_NS][PO_NS];
    int i, j;
    double mu = 0;
    for (i = 0; i < 20; i++) {
        double row = 0;
        for (j = 0; j < 20; j++) if (j != i) { m->Q[i][j] = S[i][j] * m->pi[j]; row += m->Q[i][j]; }
        m->Q[i][i] = -row; mu += m->pi[i] * row;
    }
    for (i = 0; i < 20; i++) for (j = 0; j < 20; j++) m->Q[i][j] /= mu;
    for (i = 0; i < 20; i++) for (j = 0; j < 20; j++)
        B[i][j] = sqrt(m->pi[i]) * m->Q[i][j] / sqrt(m->pi[j]);
    for (i = 0; i < 20; i++) for (j = i + 1; j < 20; j++) { B[i][j] = B[j][i] = 0.5 * (B[i][j] + B[j][i]); }
    jacobi20(B, m->eval, V);
    for (i = 0; i < 20; i++) for (j = 0; j < 20; j++) {
        m->U[i][j] = V[i][j] / sqrt(m->pi[i]);
        m->Uinv[j][i] = V[i][j] * sqrt(m->pi[i]);
    }
}

void po_pmatrix(const po_model *m, double t, double P[PO_NS][PO_NS]) {
    double e[PO_NS]; int i, j, k;
    for (k = 0; k < 20; k++) e[k] = exp(m->eval[k] * t);
    for (i = 0; i < 20; i++) for (j = 0; j < 20; j++) {
        double s = 0;
        for (k = 0; k < 20; k++) s += m->U[i][k] * e[k] * m->Uinv[k][j];
        P[i][j] = s < 0 ? 0.0 : s;
    }
}

/* ------------------------------------------------------------------------------------------
 * discrete Gamma (Yang 1994)
 * ---------------------------------------------------------------------------------------- */
double po_lngamma(double x) { return lgamma(x); }

double po_incgamma(double a, double x) {
    if (x <= 0) return 0.0;
    double gln = lgamma(a);
    if (x < a + 1.0) {                      /* series */
        double ap = a, sum = 1.0 / a, del = sum; int n;
        for (n = 0; n < 100000; n++) { ap += 1.0; del *= x / ap; sum += del; if (fabs(del) < fabs(sum) * 1e-17) break; }
        return sum * exp(-x + a * log(x) - gln);
    } else {                                /* continued fraction (modified Lentz) */
        double tiny = 1e-300, b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d; int i;
        for (i = 1; i < 100000; i++) {
            double an = -i * (i - a); b += 2.0;
            d = an * d + b; if (fabs(d) < tiny) d = tiny;
            c = b + an / c; if (fabs(c) < tiny) c = tiny;
            d = 1.0 / d; double del = d * c; h *= del;
            if (fabs(del - 1.0) < 1e-16) break;
        }
        return 1.0 - exp(-x + a * log(x) - gln) * h;
    }
}

double po_gamma_quantile(double p, double a) {
    /* bisection on log x: robust for alpha down to 0.02 */
    double lo = -1600.0, hi = log(a + 40.0 * sqrt(a) + 400.0); int i;
    for (i = 0; i < 400; i++) {
        double mid = 0.5 * (lo + hi);
        if (po_incgamma(a, exp(mid)) < p) lo = mid; else hi = mid;
        if (hi - lo < 1e-15 * fmax(1.0, fabs(mid))) break;
    }
    return exp(0.5 * (lo + hi));
}

void po_gamma_rates(double alpha, int K, int median, double *rates) {
    int i;
    if (K == 1) { rates[0] = 1.0; return; }
    if (median) {
        double s = 0;
        for (i = 0; i < K; i++) { rates[i] = po_gamma_quantile((2.0 * i + 1.0) / (2.0 * K), alpha) / alpha; s += rates[i]; }
        for (i = 0; i < K; i++) rates[i] *= K / s;
        return;
    }
    double prev = 0.0;
    for (i = 0; i < K; i++) {
        double cur = (i == K - 1) ? 1.0 : po_incgamma(alpha + 1.0, po_gamma_quantile((i + 1.0) / K, alpha));
        rates[i] = (cur - prev) * K; prev = cur;
    }
}

/* ------------------------------------------------------------------------------------------
 * alignment
 * ---------------------------------------------------------------------------------------- */
int po_char_code(int c) {
    static const char *aa = "ARNDCQEGHILKMFPSTWYV";
    c = toupper(c);
    const char *p = (c != 0) ? strchr(aa, c) : NULL;
    if (p) return (int)(p - aa);
    if (c == 'B') return 20;
    if (c == 'Z') return 21;
    return 22;
}
unsigned po_code_mask(int code) {
    if (code < 20) return 1u << code;
    if (code == 20) return (1u << 2) | (1u << 3);   /* B = N or D */
    if (code == 21) return (1u << 5) | (1u << 6);   /* Z = Q or E */
    return 0xFFFFFu;
}

po_aln *po_aln_create(int ntax, int nsites, const char *const *names, const char *const *rows, int compress) {
    po_aln *a = (po_aln *)calloc(1, sizeof(po_aln));
    int i, s;
    a->ntax = ntax; a->nsites = nsites;
    a->names = (char **)calloc(ntax, sizeof(char *));
    for (i = 0; i < ntax; i++) a->names[i] = strdup(names[i]);
    unsigned char *col = (unsigned char *)malloc((size_t)ntax * nsites);   /* [site][tax] */
    for (i = 0; i < ntax; i++) for (s = 0; s < nsites; s++) col[(size_t)s * ntax + i] = (unsigned char)po_char_code(rows[i][s]);
    a->site2pat = (int *)malloc(sizeof(int) * (nsites > 0 ? nsites : 1));
    int *first = (int *)malloc(sizeof(int) * (nsites > 0 ? nsites : 1));   /* pattern -> first site */
    a->weight = (int *)calloc(nsites > 0 ? nsites : 1, sizeof(int));
    int hsize = 1; while (hsize < 2 * nsites + 16) hsize <<= 1;
    int *table = (int *)malloc(sizeof(int) * hsize);
    for (i = 0; i < hsize; i++) table[i] = -1;
    int npat = 0;
    for (s = 0; s < nsites; s++) {
        const unsigned char *c = col + (size_t)s * ntax;
        int pat = -1;
        if (compress) {
            unsigned long long h = 1469598103934665603ULL;
            for (i = 0; i < ntax; i++) { h ^= c[i]; h *= 1099511628211ULL; }
            int slot = (int)(h & (unsigned)(hsize - 1));
            while (table[slot] >= 0) {
                if (memcmp(col + (size_t)first[table[slot]] * ntax, c, ntax) == 0) { pat = table[slot]; break; }
                slot = (slot + 1) & (hsize - 1);
            }
            if (pat < 0) { table[slot] = npat; }
        }
        if (pat < 0) { pat = npat; first[npat++] = s; }
        a->site2pat[s] = pat; a->weight[pat]++;
    }
    a->npat = npat;
    a->codes = (unsigned char *)malloc((size_t)ntax * (npat > 0 ? npat : 1));
    for (i = 0; i < ntax; i++) for (s = 0; s < npat; s++) a->codes[(size_t)i * npat + s] = col[(size_t)first[s] * ntax + i];
    free(col); free(first); free(table);
    return a;
}
void po_aln_free(po_aln *a) {
    if (!a) return;
    for (int i = 0; i < a->ntax; i++) free(a->names[i]);
    free(a->names); free(a->codes); free(a->weight); free(a->site2pat); free(a);
}

/* ------------------------------------------------------------------------------------------
 * tree
 * ---------------------------------------------------------------------------------------- */
typedef struct pnode { int parent, nchild, cap; int *child; double len; int haslen; int tip; } pnode;
typedef struct { const char *s; int pos; pnode *n; int nn, cap; const po_aln *a; char *err; int errlen; int fail; int *seen; } pstate;

static int p_new(pstate *st) {
    if (st->nn == st->cap) { st->cap = st->cap ? 2 * st->cap : 64; st->n = (pnode *)realloc(st->n, sizeof(pnode) * st->cap); }
    pnode *x = &st->n[st->nn]; memset(x, 0, sizeof(*x)); x->parent = -1; x->tip = -1; x->len = 0.1;
    return st->nn++;
}
static void p_addchild(pstate *st, int p, int c) {
    pnode *x = &st->n[p];
    if (x->nchild == x->cap) { x->cap = x->cap ? 2 * x->cap : 4; x->child = (int *)realloc(x->child, sizeof(int) * x->cap); }
    x->child[x->nchild++] = c; st->n[c].parent = p;
}
static void p_fail(pstate *st, const char *msg) { if (!st->fail && st->err) snprintf(st->err, st->errlen, "%s (at char %d)", msg, st->pos); st->fail = 1; }
static void p_ws(pstate *st) {
    for (;;) {
        while (st->s[st->pos] && isspace((unsigned char)st->s[st->pos])) st->pos++;
        if (st->s[st->pos] == '[') { while (st->s[st->pos] && st->s[st->pos] != ']') st->pos++; if (st->s[st->pos]) st->pos++; }
        else break;
    }
}
static int p_subtree(pstate *st);
static void p_label_len(pstate *st, int id, int isleaf) {
    p_ws(st);
    int b = st->pos;
    if (st->s[st->pos] == '\'') { st->pos++; b = st->pos; while (st->s[st->pos] && st->s[st->pos] != '\'') st->pos++; }
    else while (st->s[st->pos] && !strchr(",():;[", st->s[st->pos]) && !isspace((unsigned char)st->s[st->pos])) st->pos++;
    int e = st->pos;
    if (st->s[st->pos] == '\'') st->pos++;
    if (isleaf) {
        int i, found = -1;
        for (i = 0; i < st->a->ntax; i++) if ((int)strlen(st->a->names[i]) == e - b && strncmp(st->a->names[i], st->s + b, e - b) == 0) { found = i; break; }
        if (found < 0) { p_fail(st, "leaf name not in alignment"); return; }
        if (st->seen[found]) { p_fail(st, "duplicate leaf name"); return; }
        st->seen[found] = 1; st->n[id].tip = found;
    }
    p_ws(st);
    if (st->s[st->pos] == ':') {
        st->pos++; p_ws(st);
        char *end; double v = strtod(st->s + st->pos, &end);
        if (end == st->s + st->pos) { p_fail(st, "bad branch length"); return; }
        st->pos = (int)(end - st->s); st->n[id].len = v; st->n[id].haslen = 1;
    }
    p_ws(st);
}
static int p_subtree(pstate *st) {
    if (st->fail) return -1;
    p_ws(st);
    int id = p_new(st);
    if (st->s[st->pos] == '(') {
        st->pos++;
        for (;;) {
            int c = p_subtree(st); if (st->fail) return -1;
            p_addchild(st, id, c); p_ws(st);
            if (st->s[st->pos] == ',') { st->pos++; continue; }
            if (st->s[st->pos] == ')') { st->pos++; break; }
            p_fail(st, "expected , or )"); return -1;
        }
        p_label_len(st, id, 0);
    } else p_label_len(st, id, 1);
    return id;
}

static void t_connect(po_tree *t, int a, int b, double len) {
    int k;
    for (k = 0; k < 3 && t->nbr[a][k] >= 0; k++) ;
    t->nbr[a][k] = b; t->len[a][k] = len;
    for (k = 0; k < 3 && t->nbr[b][k] >= 0; k++) ;
    t->nbr[b][k] = a; t->len[b][k] = len;
}
static po_tree *t_alloc(int ntax) {
    po_tree *t = (po_tree *)calloc(1, sizeof(po_tree));
    t->ntax = ntax; t->nnodes = 2 * ntax - 2;
    t->nbr = (int (*)[3])malloc(sizeof(int[3]) * t->nnodes);
    t->len = (double (*)[3])calloc(t->nnodes, sizeof(double[3]));
    for (int i = 0; i < t->nnodes; i++) t->nbr[i][0] = t->nbr[i][1] = t->nbr[i][2] = -1;
    return t;
}

/* convert parsed rooted multifurcating structure (node r, attached to parent id `up` with length
 * `len`) into the binary unrooted array form; polytomies resolved with PO_TMIN branches */
static int t_build(pstate *st, po_tree *t, int r, int *next_inner) {
    pnode *x = &st->n[r];
    while (x->nchild == 1) {                 /* collapse unary nodes */
        int c = x->child[0]; st->n[c].len += x->len * (x->parent >= 0); r = c; x = &st->n[r];
    }
    if (x->nchild == 0) return x->tip;
    int id = (*next_inner)++;
    /* resolve to two children */
    int nc = x->nchild, i;
    int *ids = (int *)malloc(sizeof(int) * nc); double *ls = (double *)malloc(sizeof(double) * nc);
    for (i = 0; i < nc; i++) {
        int c = x->child[i]; double l = st->n[c].len; int cc = c;
        while (st->n[cc].nchild == 1) { cc = st->n[cc].child[0]; l += st->n[cc].len; }
        ls[i] = l; ids[i] = t_build(st, t, c, next_inner);
        x = &st->n[r];
    }
    int cur = ids[0]; double curl = ls[0];
    for (i = 1; i < nc - 1; i++) {           /* ladderise extra children */
        int nid = (*next_inner)++;
        t_connect(t, nid, cur, curl); t_connect(t, nid, ids[i], ls[i]);
        cur = nid; curl = PO_TMIN;
    }
    t_connect(t, id, cur, curl); t_connect(t, id, ids[nc - 1], ls[nc - 1]);
    free(ids); free(ls);
    return id;
}

po_tree *po_tree_parse(const char *newick, const po_aln *a, char *err, int errlen) {
    pstate st; memset(&st, 0, sizeof(st));
    st.s = newick; st.a = a; st.err = err; st.errlen = errlen;
    st.seen = (int *)calloc(a->ntax, sizeof(int));
    if (err && errlen) err[0] = 0;
    int root = p_subtree(&st);
    po_tree *t = NULL;
    int i;
    if (!st.fail) {
        for (i = 0; i < a->ntax; i++) if (!st.seen[i]) { p_fail(&st, "taxon missing from tree"); break; }
    }
    if (!st.fail && a->ntax < 3) p_fail(&st, "need at least 3 taxa");
    if (!st.fail) {
        while (st.n[root].nchild == 1) root = st.n[root].child[0];
        t = t_alloc(a->ntax);
        int next_inner = a->ntax;
        pnode *r = &st.n[root];
        if (r->nchild == 2) {
            /* rooted: join the two root children by one branch */
            int c0 = r->child[0], c1 = r->child[1];
            double l = 0; int cc = c0; l += st.n[cc].len; while (st.n[cc].nchild == 1) { cc = st.n[cc].child[0]; l += st.n[cc].len; }
            cc = c1; l += st.n[cc].len; while (st.n[cc].nchild == 1) { cc = st.n[cc].child[0]; l += st.n[cc].len; }
            int a0 = t_build(&st, t, c0, &next_inner);
            int a1 = t_build(&st, t, c1, &next_inner);
            t_connect(t, a0, a1, l);
        } else {
            /* >=3 children: root becomes an inner node with 3 neighbours */
            int nc = r->nchild; int id = next_inner++;
            int *ids = (int *)malloc(sizeof(int) * nc); double *ls = (double *)malloc(sizeof(double) * nc);
            for (i = 0; i < nc; i++) {
                int c = st.n[root].child[i]; double l = st.n[c].len; int cc = c;
                while (st.n[cc].nchild == 1) { cc = st.n[cc].child[0]; l += st.n[cc].len; }
                ls[i] = l; ids[i] = t_build(&st, t, c, &next_inner);
            }
            int cur = ids[0]; double curl = ls[0];
            for (i = 1; i < nc - 2; i++) {
                int nid = next_inner++;
                t_connect(t, nid, cur, curl); t_connect(t, nid, ids[i], ls[i]);
                cur = nid; curl = PO_TMIN;
            }
            t_connect(t, id, cur, curl); t_connect(t, id, ids[nc - 2], ls[nc - 2]); t_connect(t, id, ids[nc - 1], ls[nc - 1]);
            free(ids); free(ls);
        }
        if (next_inner != t->nnodes) { p_fail(&st, "internal: node count mismatch"); po_tree_free(t); t = NULL; }
    }
    for (i = 0; i < st.nn; i++) free(st.n[i].child);
    free(st.n); free(st.seen);
    if (t) for (i = 0; i < t->nnodes; i++) for (int k = 0; k < 3; k++) if (t->nbr[i][k] >= 0) {
        if (!(t->len[i][k] >= 0)) t->len[i][k] = 0; if (t->len[i][k] > PO_TMAX) t->len[i][k] = PO_TMAX;
    }
    return t;
}
po_tree *po_tree_copy(const po_tree *s) {
    po_tree *t = t_alloc(s->ntax);
    memcpy(t->nbr, s->nbr, sizeof(int[3]) * s->nnodes); memcpy(t->len, s->len, sizeof(double[3]) * s->nnodes);
    return t;
}
void po_tree_free(po_tree *t) { if (!t) return; free(t->nbr); free(t->len); free(t); }
double po_tree_length(const po_tree *t) {
    double s = 0; for (int i = 0; i < t->nnodes; i++) for (int k = 0; k < 3; k++) if (t->nbr[i][k] > i) s += t->len[i][k];
    return s;
}

typedef struct { char *s; size_t n, cap; } sbuf;
static void sb_add(sbuf *b, const char *x) {
    size_t l = strlen(x);
    if (b->n + l + 1 > b->cap) { b->cap = 2 * (b->n + l + 1); b->s = (char *)realloc(b->s, b->cap); }
    memcpy(b->s + b->n, x, l + 1); b->n += l;
}
static void nw_rec(const po_tree *t, const po_aln *a, int v, int from, double len, int digits, sbuf *b) {
    char tmp[64];
    if (v < t->ntax) sb_add(b, a->names[v]);
    else {
        int first = 1; sb_add(b, "(");
        for (int k = 0; k < 3; k++) { int w = t->nbr[v][k]; if (w < 0 || w == from) continue; if (!first) sb_add(b, ","); first = 0; nw_rec(t, a, w, v, t->len[v][k], digits, b); }
        sb_add(b, ")");
    }
    snprintf(tmp, sizeof tmp, ":%.*f", digits, len); sb_add(b, tmp);
}
char *po_tree_newick(const po_tree *t, const po_aln *a, int digits) {
    sbuf b = {0, 0, 0}; char tmp[64];
    int r = t->nbr[0][0];                   /* inner neighbour of taxon 0 (ntax>=3) */
    sb_add(&b, "("); sb_add(&b, a->names[0]); snprintf(tmp, sizeof tmp, ":%.*f", digits, t->len[0][0]); sb_add(&b, tmp);
    for (int k = 0; k < 3; k++) { int w = t->nbr[r][k]; if (w < 0 || w == 0) continue; sb_add(&b, ","); nw_rec(t, a, w, r, t->len[r][k], digits, &b); }
    sb_add(&b, ");");
    return b.s;
}

/* bipartitions: bitset of taxa on the side NOT containing taxon 0 */
static void bip_rec(const po_tree *t, int v, int from, unsigned long long *sets, int words, int *count, unsigned long long *out) {
    memset(out, 0, sizeof(unsigned long long) * words);
    if (v < t->ntax) { out[v >> 6] |= 1ULL << (v & 63); return; }
    unsigned long long *tmp = (unsigned long long *)malloc(sizeof(unsigned long long) * words);
    for (int k = 0; k < 3; k++) { int w = t->nbr[v][k]; if (w < 0 || w == from) continue; bip_rec(t, w, v, sets, words, count, tmp); for (int i = 0; i < words; i++) out[i] |= tmp[i]; }
    free(tmp);
    if (from >= t->ntax || from < 0) { /* internal edge (v,from) with from inner */ }
    if (from >= t->ntax) { memcpy(sets + (size_t)(*count) * words, out, sizeof(unsigned long long) * words); (*count)++; }
}
static int g_words;
static int bip_cmp(const void *a, const void *b) { return memcmp(a, b, sizeof(unsigned long long) * g_words); }
static unsigned long long *tree_bips(const po_tree *t, int *count) {
    int words = (t->ntax + 63) / 64;
    unsigned long long *sets = (unsigned long long *)calloc((size_t)(t->nnodes) * words, sizeof(unsigned long long));
    unsigned long long *out = (unsigned long long *)malloc(sizeof(unsigned long long) * words);
    *count = 0;
    int r = t->nbr[0][0];
    for (int k = 0; k < 3; k++) { int w = t->nbr[r][k]; if (w < 0 || w == 0) continue; bip_rec(t, w, r, sets, words, count, out); }
    free(out);
    g_words = words; qsort(sets, *count, sizeof(unsigned long long) * words, bip_cmp);
    return sets;
}
int po_tree_rf(const po_tree *a, const po_tree *b) {
    int na, nb, words = (a->ntax + 63) / 64;
    unsigned long long *sa = tree_bips(a, &na), *sb = tree_bips(b, &nb);
    int i = 0, j = 0, common = 0; g_words = words;
    while (i < na && j < nb) {
        int c = bip_cmp(sa + (size_t)i * words, sb + (size_t)j * words);
        if (c == 0) { common++; i++; j++; } else if (c < 0) i++; else j++;
    }
    free(sa); free(sb);
    return (na + nb - 2 * common) / 2;
}

/* ------------------------------------------------------------------------------------------
 * likelihood engine: directional CLVs  D(v -> nbr k) with lazy validity
 * ---------------------------------------------------------------------------------------- */
struct po_engine {
    const po_aln *a; const po_model *m;
    int K; double alpha; double rates[PO_MAXCAT];
    int ntax, npat, nnodes;
    double **clv;      /* [ (v-ntax)*3 + k ] -> npat*K*20 doubles, layout [pat][cat][state] */
    int **scl;         /* per-pattern cumulative scaling counts */
    unsigned char *valid;
    const po_tree *bound;   /* tree the cache refers to */
    double tipvec[PO_NCODES][PO_NS];
    double *sumtab;    /* npat*K*20 */
    double ntol;       /* Newton stop: |dt| < ntol */
    unsigned char *dirty, *dirty_next;   /* [node*3+slot]: branch needs re-optimisation (both directions set) */
    long n_newview, n_evaluate, n_deriv;
    /* topological constraints (FastTree -constraints, FastTreeRunner.java:54-64,243-273): split i = taxa cone[i] | taxa czero[i] */
    int ncons, cwords; unsigned long long *cone, *czero;
};

po_engine *po_engine_create(const po_aln *a, const po_model *m, int ncat, double alpha) {
    po_engine *e = (po_engine *)calloc(1, sizeof(po_engine));
    e->a = a; e->m = m; e->K = ncat; e->ntax = a->ntax; e->npat = a->npat; e->nnodes = 2 * a->ntax - 2;
    int nd = (e->nnodes - e->ntax) * 3;
    e->clv = (double **)calloc(nd, sizeof(double *)); e->scl = (int **)calloc(nd, sizeof(int *));
    e->valid = (unsigned char *)calloc(nd, 1);
    for (int c = 0; c < PO_NCODES; c++) { unsigned mk = po_code_mask(c); for (int s = 0; s < 20; s++) e->tipvec[c][s] = (mk >> s) & 1 ? 1.0 : 0.0; }
    e->sumtab = (double *)malloc(sizeof(double) * (size_t)(e->npat > 0 ? e->npat : 1) * ncat * 20);
    e->ntol = 1e-8;
    e->dirty = (unsigned char *)calloc((size_t)e->nnodes * 3, 1); e->dirty_next = (unsigned char *)calloc((size_t)e->nnodes * 3, 1);
    po_engine_set_alpha(e, alpha);
    return e;
}
void po_engine_free(po_engine *e) {
    if (!e) return;
    int nd = (e->nnodes - e->ntax) * 3;
    for (int i = 0; i < nd; i++) { free(e->clv[i]); free(e->scl[i]); }
    free(e->clv); free(e->scl); free(e->valid); free(e->sumtab); free(e->dirty); free(e->dirty_next); free(e->cone); free(e->czero); free(e);
}
static void eng_invalidate_all(po_engine *e) { memset(e->valid, 0, (size_t)(e->nnodes - e->ntax) * 3); }
void po_engine_set_alpha(po_engine *e, double alpha) {
    if (alpha < PO_ALPHA_MIN) alpha = PO_ALPHA_MIN; if (alpha > PO_ALPHA_MAX) alpha = PO_ALPHA_MAX;
    e->alpha = alpha; po_gamma_rates(alpha, e->K, 0, e->rates); eng_invalidate_all(e);
}
double po_engine_alpha(const po_engine *e) { return e->alpha; }

static int slot_of(const po_tree *t, int v, int w) { for (int k = 0; k < 3; k++) if (t->nbr[v][k] == w) return k; return -1; }

/* after the length of (a,b) changed (or topology changed around it): everything that "sees"
 * the branch from outside becomes stale */
static void eng_invalidate_from(po_engine *e, const po_tree *t, int v, int from) {
    if (v < e->ntax) return;
    for (int k = 0; k < 3; k++) {
        int w = t->nbr[v][k]; if (w == from || w < 0) continue;
        e->valid[(v - e->ntax) * 3 + k] = 0;
        eng_invalidate_from(e, t, w, v);
    }
}
static void eng_branch_changed(po_engine *e, const po_tree *t, int a, int b) {
    eng_invalidate_from(e, t, a, b); eng_invalidate_from(e, t, b, a);
}

/* x[cat][s] = sum_j P_cat[s][j] * child[cat][j]  for one pattern */
static void eng_newview(po_engine *e, const po_tree *t, int v, int k);

typedef struct { const double *clv; const int *scl; const unsigned char *codes; } side;
static side eng_side(po_engine *e, const po_tree *t, int v, int to) {
    /* the message from v towards `to` */
    side s = {0, 0, 0};
    if (v < e->ntax) { s.codes = e->a->codes + (size_t)v * e->npat; return s; }
    int k = slot_of(t, v, to);
    if (!e->valid[(v - e->ntax) * 3 + k]) eng_newview(e, t, v, k);
    s.clv = e->clv[(v - e->ntax) * 3 + k]; s.scl = e->scl[(v - e->ntax) * 3 + k];
    return s;
}
static void eng_pmats(const po_engine *e, double t, double P[][PO_NS][PO_NS]) {
    for (int c = 0; c < e->K; c++) po_pmatrix(e->m, t * e->rates[c], P[c]);
}
/* generic CLV combination: out = (P(bl0).L) * (P(bl1).R) with the 2^256 rescue */
static void nv_core(po_engine *e, side L, side R, double bl0, double bl1, double *out, int *osc) {
    int K = e->K, np = e->npat;
    double (*PL)[PO_NS][PO_NS] = (double (*)[PO_NS][PO_NS])malloc(sizeof(double[PO_NS][PO_NS]) * K);
    double (*PR)[PO_NS][PO_NS] = (double (*)[PO_NS][PO_NS])malloc(sizeof(double[PO_NS][PO_NS]) * K);
    eng_pmats(e, bl0, PL); eng_pmats(e, bl1, PR);
    const double two256 = ldexp(1.0, 256), thresh = ldexp(1.0, -256);
    for (int p = 0; p < np; p++) {
        double mx = 0.0;
        for (int c = 0; c < K; c++) {
            const double *xl = L.clv ? L.clv + ((size_t)p * K + c) * 20 : e->tipvec[L.codes[p]];
            const double *xr = R.clv ? R.clv + ((size_t)p * K + c) * 20 : e->tipvec[R.codes[p]];
            double *o = out + ((size_t)p * K + c) * 20;
            for (int s = 0; s < 20; s++) {
                double a = 0, b = 0;
                for (int j = 0; j < 20; j++) { a += PL[c][s][j] * xl[j]; b += PR[c][s][j] * xr[j]; }
                o[s] = a * b; if (o[s] > mx) mx = o[s];
            }
        }
        int sc = (L.scl ? L.scl[p] : 0) + (R.scl ? R.scl[p] : 0);
        if (mx < thresh) { double *o = out + (size_t)p * K * 20; for (int q = 0; q < K * 20; q++) o[q] *= two256; sc++; }
        osc[p] = sc;
    }
    free(PL); free(PR);
    e->n_newview++;
}
static void eng_newview(po_engine *e, const po_tree *t, int v, int k) {
    int idx = (v - e->ntax) * 3 + k, K = e->K, np = e->npat;
    int ch[2], ci = 0; double bl[2];
    for (int q = 0; q < 3; q++) if (q != k) { ch[ci] = t->nbr[v][q]; bl[ci] = t->len[v][q]; ci++; }
    side L = eng_side(e, t, ch[0], v), R = eng_side(e, t, ch[1], v);
    if (!e->clv[idx]) { e->clv[idx] = (double *)malloc(sizeof(double) * (size_t)np * K * 20); e->scl[idx] = (int *)malloc(sizeof(int) * np); }
    nv_core(e, L, R, bl[0], bl[1], e->clv[idx], e->scl[idx]);
    e->valid[idx] = 1;
}
static void eng_bind(po_engine *e, const po_tree *t) { if (e->bound != t) { e->bound = t; eng_invalidate_all(e); } }

/* lnL evaluated on branch (u,v) */
static double eng_evaluate(po_engine *e, const po_tree *t, int u, int v, double *pat_lnl) {
    int K = e->K, np = e->npat;
    side A = eng_side(e, t, u, v), B = eng_side(e, t, v, u);
    double bl = t->len[u][slot_of(t, u, v)];
    double (*P)[PO_NS][PO_NS] = (double (*)[PO_NS][PO_NS])malloc(sizeof(double[PO_NS][PO_NS]) * K);
    eng_pmats(e, bl, P);
    double total = 0;
    for (int p = 0; p < np; p++) {
        double site = 0;
        for (int c = 0; c < K; c++) {
            const double *xa = A.clv ? A.clv + ((size_t)p * K + c) * 20 : e->tipvec[A.codes[p]];
            const double *xb = B.clv ? B.clv + ((size_t)p * K + c) * 20 : e->tipvec[B.codes[p]];
            double cat = 0;
            for (int s = 0; s < 20; s++) { double y = 0; for (int j = 0; j < 20; j++) y += P[c][s][j] * xb[j]; cat += e->m->pi[s] * xa[s] * y; }
            site += cat;
        }
        site /= K;
        int sc = (A.scl ? A.scl[p] : 0) + (B.scl ? B.scl[p] : 0);
        double l = log(site) - sc * PO_LOG_2_256;
        if (pat_lnl) pat_lnl[p] = l;
        total += e->a->weight[p] * l;
    }
    free(P); e->n_evaluate++;
    return total;
}
double po_engine_lnl(po_engine *e, const po_tree *t, double *pat_lnl) {
    eng_bind(e, t);
    return eng_evaluate(e, t, 0, t->nbr[0][0], pat_lnl);
}
double po_engine_site_lnl(po_engine *e, const po_tree *t, double *site_lnl) {
    double *pl = (double *)malloc(sizeof(double) * (e->npat > 0 ? e->npat : 1));
    double tot = po_engine_lnl(e, t, pl);
    for (int s = 0; s < e->a->nsites; s++) site_lnl[s] = pl[e->a->site2pat[s]];
    free(pl); return tot;
}

/* sumtable for two sides: S[p][c][i] = (sum_s pi_s A[s] U[s][i]) * (sum_j Uinv[i][j] B[j]) */
static void sumtable_core(po_engine *e, side A, side B, int *scale_out) {
    int K = e->K, np = e->npat;
    for (int p = 0; p < np; p++) {
        for (int c = 0; c < K; c++) {
            const double *xa = A.clv ? A.clv + ((size_t)p * K + c) * 20 : e->tipvec[A.codes[p]];
            const double *xb = B.clv ? B.clv + ((size_t)p * K + c) * 20 : e->tipvec[B.codes[p]];
            double *o = e->sumtab + ((size_t)p * K + c) * 20;
            for (int i = 0; i < 20; i++) {
                double l = 0, r = 0;
                for (int s = 0; s < 20; s++) { l += e->m->pi[s] * xa[s] * e->m->U[s][i]; r += e->m->Uinv[i][s] * xb[s]; }
                o[i] = l * r;
            }
        }
        if (scale_out) scale_out[p] = (A.scl ? A.scl[p] : 0) + (B.scl ? B.scl[p] : 0);
    }
}
static void eng_sumtable(po_engine *e, const po_tree *t, int u, int v, int *scale_out) {
    side A = eng_side(e, t, u, v), B = eng_side(e, t, v, u);
    sumtable_core(e, A, B, scale_out);
}
/* lnL (without scaling constant), d1, d2 at branch length tt from the sumtable */
static void eng_core_derivs(po_engine *e, double tt, const int *scale, double *lnl, double *d1, double *d2) {
    int K = e->K, np = e->npat;
    double ex[PO_MAXCAT][PO_NS], g1[PO_MAXCAT][PO_NS], g2[PO_MAXCAT][PO_NS];
    for (int c = 0; c < K; c++) for (int i = 0; i < 20; i++) {
        double lr = e->m->eval[i] * e->rates[c];
        ex[c][i] = exp(lr * tt); g1[c][i] = lr * ex[c][i]; g2[c][i] = lr * lr * ex[c][i];
    }
    double L = 0, D1 = 0, D2 = 0;
    for (int p = 0; p < np; p++) {
        double f = 0, f1 = 0, f2 = 0;
        const double *s = e->sumtab + (size_t)p * K * 20;
        for (int c = 0; c < K; c++) for (int i = 0; i < 20; i++) { double x = s[c * 20 + i]; f += x * ex[c][i]; f1 += x * g1[c][i]; f2 += x * g2[c][i]; }
        double w = e->a->weight[p], r1 = f1 / f;
        L += w * (log(f / K) - (scale ? scale[p] * PO_LOG_2_256 : 0.0));
        D1 += w * r1; D2 += w * (f2 / f - r1 * r1);
    }
    *lnl = L; *d1 = D1; *d2 = D2; e->n_deriv++;
}
void po_engine_branch_derivs(po_engine *e, const po_tree *t, int u, int v, double *lnl, double *d1, double *d2) {
    eng_bind(e, t);
    int *sc = (int *)malloc(sizeof(int) * (e->npat > 0 ? e->npat : 1));
    eng_sumtable(e, t, u, v, sc);
    eng_core_derivs(e, t->len[u][slot_of(t, u, v)], sc, lnl, d1, d2);
    free(sc);
}

/* Newton-Raphson on one branch with step control; returns new length.
 * Spec (mirrored by the HIP engine, see DESIGN.md "branch Newton"):
 *   t0 = clamp(t); up to 32 iterations: (L,d1,d2) at t; if d2<0 step=-d1/d2 else step = d1>0 ? t : -t/2;
 *   tn = clamp(t+step, TMIN, TMAX); backtrack (halve step, <=8x) while L(tn) < L(t) - 1e-9;
 *   stop when |tn-t| < e->ntol */
static double eng_newton_branch(po_engine *e, double t0, double *lnl_out) {
    double t = t0 < PO_TMIN ? PO_TMIN : (t0 > PO_TMAX ? PO_TMAX : t0);
    double L, d1, d2;
    eng_core_derivs(e, t, NULL, &L, &d1, &d2);
    for (int it = 0; it < 32; it++) {
        double step = (d2 < 0) ? -d1 / d2 : (d1 > 0 ? t : -0.5 * t);
        double tn = t + step, Ln, n1, n2; int bt = 0;
        if (fabs(step) < e->ntol && d2 < 0) {        /* converged: take the (sub-tolerance) step unevaluated */
            t = tn < PO_TMIN ? PO_TMIN : (tn > PO_TMAX ? PO_TMAX : tn);
            break;
        }
        for (;;) {
            if (tn < PO_TMIN) tn = PO_TMIN; if (tn > PO_TMAX) tn = PO_TMAX;
            eng_core_derivs(e, tn, NULL, &Ln, &n1, &n2);
            if (Ln >= L - 1e-9 || bt >= 8) break;
            bt++; tn = 0.5 * (tn + t);
        }
        if (Ln < L - 1e-9) break;            /* could not improve: keep t */
        double dt = fabs(tn - t);
        t = tn; L = Ln; d1 = n1; d2 = n2;
        if (dt < e->ntol) break;
    }
    if (lnl_out) *lnl_out = L;
    return t;
}
static void tree_set_len(po_tree *t, int u, int v, double l) { t->len[u][slot_of(t, u, v)] = l; t->len[v][slot_of(t, v, u)] = l; }

/* dirty-branch bookkeeping: a smoothing pass re-optimises only branches flagged dirty; a branch
 * whose length moved by more than thr flags itself and every branch sharing a node with it for
 * the NEXT pass (DESIGN.md "smoothing pass") */
static void mark_node(unsigned char *f, const po_tree *t, int v) {
    for (int k = 0; k < 3; k++) { int w = t->nbr[v][k]; if (w < 0) continue; f[v * 3 + k] = 1; f[w * 3 + slot_of(t, w, v)] = 1; }
}
static void mark_all(po_engine *e) { memset(e->dirty, 1, (size_t)e->nnodes * 3); }
static void eng_smooth_rec(po_engine *e, po_tree *t, int v, int from, double *maxdelta, double thr) {
    for (int k = 0; k < 3; k++) {
        int w = t->nbr[v][k]; if (w < 0 || w == from) continue;
        if (e->dirty[v * 3 + k]) {
            eng_sumtable(e, t, v, w, NULL);
            double old = t->len[v][k], nl = eng_newton_branch(e, old, NULL), dl = fabs(nl - old);
            if (dl > *maxdelta) *maxdelta = dl;
            if (nl != old) { tree_set_len(t, v, w, nl); eng_branch_changed(e, t, v, w); }
            if (dl > thr) { mark_node(e->dirty_next, t, v); mark_node(e->dirty_next, t, w); }
        }
        if (w >= e->ntax) eng_smooth_rec(e, t, w, v, maxdelta, thr);
    }
}
/* one smoothing pass over the dirty branches, DFS from taxon 0; returns max |dt| */
static double eng_smooth(po_engine *e, po_tree *t, double thr) {
    double md = 0;
    memset(e->dirty_next, 0, (size_t)e->nnodes * 3);
    eng_smooth_rec(e, t, 0, -1, &md, thr);
    memcpy(e->dirty, e->dirty_next, (size_t)e->nnodes * 3);
    return md;
}

/* Brent maximisation of lnL over alpha on log scale */
static double eng_alpha_obj(po_engine *e, po_tree *t, double la) { po_engine_set_alpha(e, exp(la)); return -po_engine_lnl(e, t, NULL); }
/* Brent on log(alpha) with tolerance tol, bracketed like RAxML brackets its model parameters: a window of
 * +-ln 4 around the current value (clipped to [ALPHA_MIN, ALPHA_MAX]); if the minimum ends at an edge of the
 * window that is not a global limit, the search continues from there in a window twice as wide */
static double eng_opt_alpha(po_engine *e, po_tree *t, double tol) {
    const double gold = 0.3819660112501051;
    const double LMIN = log(PO_ALPHA_MIN), LMAX = log(PO_ALPHA_MAX);
    double x = log(e->alpha), fx = eng_alpha_obj(e, t, x), W = log(4.0);
    for (int win = 0; win < 8; win++) {
        double a = x - W > LMIN ? x - W : LMIN, b = x + W < LMAX ? x + W : LMAX;
        const double lo = a, hi = b;
        double w = x, v = x, fw = fx, fv = fx, d = 0, ee = 0;
        for (int it = 0; it < 60; it++) {
            double xm = 0.5 * (a + b), tol1 = tol, tol2 = 2 * tol1;      /* absolute in log(alpha) = relative in alpha */
            if (fabs(x - xm) <= tol2 - 0.5 * (b - a)) break;
            int golden = 1; double u;
            if (fabs(ee) > tol1) {
                double r = (x - w) * (fx - fv), q = (x - v) * (fx - fw), p = (x - v) * q - (x - w) * r;
                q = 2 * (q - r); if (q > 0) p = -p; q = fabs(q);
                double etemp = ee; ee = d;
                if (!(fabs(p) >= fabs(0.5 * q * etemp) || p <= q * (a - x) || p >= q * (b - x))) {
                    d = p / q; u = x + d; if (u - a < tol2 || b - u < tol2) d = (xm - x >= 0) ? tol1 : -tol1; golden = 0;
                }
            }
            if (golden) { ee = (x >= xm) ? a - x : b - x; d = gold * ee; }
            u = (fabs(d) >= tol1) ? x + d : x + (d >= 0 ? tol1 : -tol1);
            double fu = eng_alpha_obj(e, t, u);
            if (fu <= fx) { if (u >= x) a = x; else b = x; v = w; fv = fw; w = x; fw = fx; x = u; fx = fu; }
            else { if (u < x) a = u; else b = u; if (fu <= fw || w == x) { v = w; fv = fw; w = u; fw = fu; } else if (fu <= fv || v == x || v == w) { v = u; fv = fu; } }
        }
        const double edge = 4 * tol;
        if ((x - lo < edge && lo > LMIN) || (hi - x < edge && hi < LMAX)) { W *= 2; continue; }
        break;
    }
    po_engine_set_alpha(e, exp(x));
    return -fx;
}

double po_engine_optimize(po_engine *e, po_tree *t, int opt_alpha, double eps) {
    eng_bind(e, t);
    for (int i = 0; i < t->nnodes; i++) for (int k = 0; k < 3; k++) if (t->nbr[i][k] >= 0 && t->len[i][k] < PO_TMIN) t->len[i][k] = PO_TMIN;
    eng_invalidate_all(e);
    double lnl = po_engine_lnl(e, t, NULL);
    /* precision follows eps: passes stop when max |dt| < thr = clamp(eps/100, 1e-6, 1e-3), Newton stops
     * at thr/100; <= 8 passes per round for eps >= 0.05 (the search's intermediate optimisations),
     * <= 16 otherwise; every round starts with all branches dirty */
    const int maxpass = eps >= 0.05 ? 8 : 16;
    double thr = eps * 0.01; if (thr < 1e-6) thr = 1e-6; if (thr > 1e-3) thr = 1e-3;
    const double save = e->ntol;
    e->ntol = thr * 0.01;
    for (int round = 0; round < 100; round++) {
        mark_all(e);
        /* geometric pass budget 1, 2, 4, ... maxpass: while alpha is still moving a lot, branch
         * lengths are not polished to thr (they shift again with the next alpha) */
        int budget = opt_alpha ? (1 << (round < 5 ? round : 5)) : maxpass; if (budget > maxpass) budget = maxpass;
        for (int pass = 0; pass < budget; pass++) { if (eng_smooth(e, t, thr) < thr) break; }
        double nl = opt_alpha ? eng_opt_alpha(e, t, eps >= 0.05 ? 1e-2 : 1e-4) : po_engine_lnl(e, t, NULL);
        double gain = nl - lnl; lnl = nl;
        if (gain < eps) break;
    }
    e->ntol = save;
    return lnl;
}

/* ------------------------------------------------------------------------------------------
 * FastTree's `-gamma` likelihood ("Gamma(20) LogLk ... alpha ... rescaling lengths by ..."), the last step of every
 * FastTree_WAG run PEPR makes (FastTreeRunner.java:67-70 always passes -gamma).  The program is FastTree 2.1.1 (Price,
 * Dehal & Arkin 2010), shipped only as a binary; this restates its published procedure (FastTree 2.1 GammaLogLk /
 * RescaleGammaLogLk) with the discretisation SURVEY.md Appendix B re-derived from the binary's -log output:
 *   - the tree's per-site likelihoods are taken at 20 FIXED rates r_k = 0.05 * 400^(k/19), one rate at a time;
 *   - rate k gets the weight P(mult*hi_k) - P(mult*lo_k) of a Gamma(shape alpha, mean 1) distribution, bins cut at the
 *     arithmetic midpoints of adjacent rates (first from 0, last to infinity);
 *   - lnL(alpha, mult) = sum_sites ln sum_k w_k L_site,k; alpha and mult are optimised alternately, each by a bounded
 *     one-dimensional Brent search (here: on log alpha / log mult in [0.01, 10], tolerance 1e-3), for at most 10 rounds,
 *     stopping when a round gains < 1e-3;  rescale = 1/mult multiplies the printed branch lengths.
 * PARITY UNPINNED: the survey's probe numbers (alpha 2.930, rescale 1.021, -4623.926 on its 8 x 300 toy) cannot be
 * re-derived here because that alignment was not kept and the binary may not be run.
 * ---------------------------------------------------------------------------------------- */
void po_g20_rates(double *r) { for (int k = 0; k < 20; k++) r[k] = 0.05 * pow(400.0, k / 19.0); }
void po_g20_weights(double alpha, double mult, double *w) {
    double r[20], prev = 0.0; po_g20_rates(r);
    for (int k = 0; k < 20; k++) {
        double cur = (k == 19) ? 1.0 : po_incgamma(alpha, mult * 0.5 * (r[k] + r[k + 1]) * alpha);
        w[k] = cur - prev; prev = cur;
    }
}
typedef struct { const double *tab; const int *wt; int npat; } g20_data;
static double g20_neglnl(const g20_data *d, double la, double lm) {
    double w[20]; po_g20_weights(exp(la), exp(lm), w);
    double tot = 0;
    for (int p = 0; p < d->npat; p++) {
        const double *t = d->tab + (size_t)p * 20;
        double mx = t[0]; for (int k = 1; k < 20; k++) if (t[k] > mx) mx = t[k];
        double s = 0; for (int k = 0; k < 20; k++) s += w[k] * exp(t[k] - mx);
        tot += d->wt[p] * (mx + log(s));
    }
    return -tot;
}
/* bounded Brent minimisation (same iteration as eng_opt_alpha / the engine's host-side Brent), absolute tolerance tol */
static double g20_brent(const g20_data *d, int which, double *la, double *lm, double fx0, double lo, double hi, double tol) {
    const double gold = 0.3819660112501051;
    double a = lo, b = hi, x = which ? *lm : *la, w = x, v = x, fx = fx0, fw = fx0, fv = fx0, dd = 0, ee = 0;
    for (int it = 0; it < 60; it++) {
        double xm = 0.5 * (a + b), tol1 = tol, tol2 = 2 * tol1, u;
        if (fabs(x - xm) <= tol2 - 0.5 * (b - a)) break;
        int golden = 1;
        if (fabs(ee) > tol1) {
            double r = (x - w) * (fx - fv), q = (x - v) * (fx - fw), p = (x - v) * q - (x - w) * r;
            q = 2 * (q - r); if (q > 0) p = -p; q = fabs(q);
            double etemp = ee; ee = dd;
            if (!(fabs(p) >= fabs(0.5 * q * etemp) || p <= q * (a - x) || p >= q * (b - x))) {
                dd = p / q; u = x + dd; if (u - a < tol2 || b - u < tol2) dd = (xm - x >= 0) ? tol1 : -tol1; golden = 0;
            }
        }
        if (golden) { ee = (x >= xm) ? a - x : b - x; dd = gold * ee; }
        u = (fabs(dd) >= tol1) ? x + dd : x + (dd >= 0 ? tol1 : -tol1);
        double fu = which ? g20_neglnl(d, *la, u) : g20_neglnl(d, u, *lm);
        if (fu <= fx) { if (u >= x) a = x; else b = x; v = w; fv = fw; w = x; fw = fx; x = u; fx = fu; }
        else { if (u < x) a = u; else b = u; if (fu <= fw || w == x) { v = w; fv = fw; w = u; fw = fu; } else if (fu <= fv || v == x || v == w) { v = u; fv = fu; } }
    }
    if (which) *lm = x; else *la = x;
    return fx;
}
double po_gamma20(const po_aln *a, const po_model *m, const po_tree *t, double *alpha_out, double *rescale_out, double *table_out) {
    const int np = a->npat;
    double rates[20]; po_g20_rates(rates);
    double *tab = (double *)malloc(sizeof(double) * (size_t)(np > 0 ? np : 1) * 20), *pat = (double *)malloc(sizeof(double) * (size_t)(np > 0 ? np : 1));
    for (int k = 0; k < 20; k++) {             /* likelihood at rate r_k = likelihood of the tree with every length x r_k at rate 1 */
        po_tree *tk = po_tree_copy(t);
        for (int i = 0; i < tk->nnodes; i++) for (int q = 0; q < 3; q++) if (tk->nbr[i][q] >= 0) tk->len[i][q] *= rates[k];
        po_engine *e = po_engine_create(a, m, 1, 1.0);
        po_engine_lnl(e, tk, pat);
        for (int p = 0; p < np; p++) tab[(size_t)p * 20 + k] = pat[p];
        po_engine_free(e); po_tree_free(tk);
    }
    g20_data d = { tab, a->weight, np };
    double la = 0.0, lm = 0.0, f = g20_neglnl(&d, la, lm);
    const double LO = log(0.01), HI = log(10.0);
    for (int round = 0; round < 10; round++) {
        const double start = f;
        f = g20_brent(&d, 0, &la, &lm, f, LO, HI, 1e-3);
        f = g20_brent(&d, 1, &la, &lm, f, LO, HI, 1e-3);
        if (!(f < start - 1e-3)) break;
    }
    if (table_out) memcpy(table_out, tab, sizeof(double) * (size_t)np * 20);
    free(tab); free(pat);
    *alpha_out = exp(la); *rescale_out = exp(-lm);
    return -f;
}

/* ------------------------------------------------------------------------------------------
 * brute force
 * ---------------------------------------------------------------------------------------- */
double po_bruteforce_lnl(const po_aln *a, const po_model *m, int K, double alpha, const po_tree *t) {
    int ninner = t->nnodes - t->ntax, ne = 0;
    if (ninner > 4) return NAN;
    double rates[PO_MAXCAT]; po_gamma_rates(alpha, K, 0, rates);
    int eu[16], ev[16]; double el[16];
    for (int i = 0; i < t->nnodes; i++) for (int k = 0; k < 3; k++) if (t->nbr[i][k] > i) { eu[ne] = i; ev[ne] = t->nbr[i][k]; el[ne] = t->len[i][k]; ne++; }
    long nassign = 1; for (int i = 0; i < ninner; i++) nassign *= 20;
    double total = 0;
    double (*P)[PO_NS][PO_NS] = (double (*)[PO_NS][PO_NS])malloc(sizeof(double[PO_NS][PO_NS]) * ne);
    for (int p = 0; p < a->npat; p++) {
        double site = 0;
        for (int c = 0; c < K; c++) {
            for (int q = 0; q < ne; q++) po_pmatrix(m, el[q] * rates[c], P[q]);
            double cat = 0;
            for (long as = 0; as < nassign; as++) {
                int st[8]; long x = as; for (int i = 0; i < ninner; i++) { st[i] = (int)(x % 20); x /= 20; }
                double pr = m->pi[st[0]];     /* root = first inner node */
                /* orient edges away from inner node ntax by BFS order: since P reversible, use
                 * pi_root * prod over edges P[parent][child] with orientation found by DFS */
                int stack[16], par[16], sp = 0; stack[sp] = t->ntax; par[sp] = -1; sp++;
                while (sp > 0 && pr > 0) {
                    sp--; int v = stack[sp], pv = par[sp];
                    for (int k = 0; k < 3; k++) {
                        int w = t->nbr[v][k]; if (w < 0 || w == pv) continue;
                        int q; for (q = 0; q < ne; q++) if ((eu[q] == v && ev[q] == w) || (eu[q] == w && ev[q] == v)) break;
                        if (w >= t->ntax) { pr *= P[q][st[v - t->ntax]][st[w - t->ntax]]; stack[sp] = w; par[sp] = v; sp++; }
                        else { unsigned mk = po_code_mask(a->codes[(size_t)w * a->npat + p]); double s = 0; for (int j = 0; j < 20; j++) if ((mk >> j) & 1) s += P[q][st[v - t->ntax]][j]; pr *= s; }
                    }
                }
                cat += pr;
            }
            site += cat / K;
        }
        total += a->weight[p] * log(site);
    }
    free(P);
    return total;
}

/* ------------------------------------------------------------------------------------------
 * start tree: neighbour joining on Kimura-corrected protein distances
 *   d = -ln max(1 - p - 0.2 p^2, 0.05) over positions where both residues are unambiguous
 *   (3.0 when nothing is comparable); first minimum of the Q criterion wins ties.
 * ---------------------------------------------------------------------------------------- */
/* ------------------------------------------------------------------------------------------
 * Topological constraints.  FastTree's -constraints file (PEPR writes it from a constraint tree, one 0/1 column per
 * node: FastTreeRunner.java:243-273) names splits the result should display; taxa marked '-' or absent from the matrix
 * are free in that column.  FastTree treats violations as a penalty (soft); the engine and this restatement treat them
 * as HARD: a column with >= 2 taxa on each side is a split every tree of the search must
 * be compatible with.  A leaf set X (one side of a tree edge) is compatible with a constraint (A | B) iff X misses A, or
 * misses B, or contains all of A, or contains all of B.  Shared spec (engine: host.cpp split_compatible / leaf_sets,
 * search.cpp): NJ joins only clusters whose union is compatible (if no pair is, the unconstrained minimum is taken); an
 * NNI alternative is a candidate only if the one new split it creates is compatible; an SPR regraft edge (g,h) -- and
 * everything behind it -- is skipped when leaves(h side) + leaves(pruned subtree) is incompatible; a start tree that
 * violates a constraint is replaced by the constrained NJ tree.
 * ---------------------------------------------------------------------------------------- */
void po_engine_set_constraints(po_engine *e, int ncons, int ntax_c, const char *const *names, const char *const *rows) {
    free(e->cone); free(e->czero); e->cone = e->czero = NULL; e->ncons = 0;
    const int n = e->ntax, words = (n + 63) / 64;
    e->cwords = words;
    if (ncons <= 0 || ntax_c <= 0) return;
    int *map = (int *)malloc(sizeof(int) * ntax_c);
    for (int i = 0; i < ntax_c; i++) { map[i] = -1; for (int t = 0; t < n; t++) if (!strcmp(e->a->names[t], names[i])) { map[i] = t; break; } }
    e->cone = (unsigned long long *)calloc((size_t)ncons * words, sizeof(unsigned long long));
    e->czero = (unsigned long long *)calloc((size_t)ncons * words, sizeof(unsigned long long));
    for (int c = 0; c < ncons; c++) {
        unsigned long long *one = e->cone + (size_t)e->ncons * words, *zero = e->czero + (size_t)e->ncons * words;
        memset(one, 0, sizeof(unsigned long long) * words); memset(zero, 0, sizeof(unsigned long long) * words);
        int n1 = 0, n0 = 0;
        for (int i = 0; i < ntax_c; i++) {
            int t = map[i]; if (t < 0) continue;
            if (rows[i][c] == '1') { one[t >> 6] |= 1ULL << (t & 63); n1++; }
            else if (rows[i][c] == '0') { zero[t >> 6] |= 1ULL << (t & 63); n0++; }
        }
        if (n1 >= 2 && n0 >= 2) e->ncons++;        /* smaller sides are trivially displayed */
    }
    free(map);
}
static int cons_compatible(const po_engine *e, const unsigned long long *X) {
    const int words = e->cwords;
    for (int c = 0; c < e->ncons; c++) {
        const unsigned long long *one = e->cone + (size_t)c * words, *zero = e->czero + (size_t)c * words;
        int hit1 = 0, hit0 = 0, all1 = 1, all0 = 1;
        for (int i = 0; i < words; i++) {
            if (X[i] & one[i]) hit1 = 1;
            if (X[i] & zero[i]) hit0 = 1;
            if ((X[i] & one[i]) != one[i]) all1 = 0;
            if ((X[i] & zero[i]) != zero[i]) all0 = 0;
        }
        if (!(!hit1 || !hit0 || all1 || all0)) return 0;
    }
    return 1;
}
/* leaves behind node v when the edge towards `from` is cut: L[(v - n)*3 + slot(v, from)] (words each), inner v only */
static void leafset_rec(const po_tree *t, int v, int from, unsigned long long *L, int words) {
    const int n = t->ntax;
    unsigned long long *S = L + ((size_t)(v - n) * 3 + slot_of(t, v, from)) * words;
    memset(S, 0, sizeof(unsigned long long) * words);
    for (int k = 0; k < 3; k++) {
        int w = t->nbr[v][k]; if (w == from) continue;
        if (w < n) S[w >> 6] |= 1ULL << (w & 63);
        else { leafset_rec(t, w, v, L, words); const unsigned long long *c = L + ((size_t)(w - n) * 3 + slot_of(t, w, v)) * words; for (int q = 0; q < words; q++) S[q] |= c[q]; }
    }
}
static unsigned long long *tree_leaf_sets(const po_tree *t) {
    const int n = t->ntax, words = (n + 63) / 64;
    unsigned long long *L = (unsigned long long *)calloc((size_t)3 * (n - 2) * words, sizeof(unsigned long long));
    for (int v = n; v < t->nnodes; v++) for (int k = 0; k < 3; k++) leafset_rec(t, v, t->nbr[v][k], L, words);   /* O(n^2): oracle sizes only */
    return L;
}
static void set_of(const po_tree *t, const unsigned long long *L, int words, int node, int toward, unsigned long long *out) {
    const int n = t->ntax;
    if (node < n) { memset(out, 0, sizeof(unsigned long long) * words); out[node >> 6] |= 1ULL << (node & 63); }
    else memcpy(out, L + ((size_t)(node - n) * 3 + slot_of(t, node, toward)) * words, sizeof(unsigned long long) * words);
}
static int tree_displays(const po_engine *e, const po_tree *t) {
    if (e->ncons == 0) return 1;
    const int n = t->ntax, words = e->cwords;
    unsigned long long *L = tree_leaf_sets(t);
    int ok = 1;
    for (int i = 0; i < 3 * (n - 2) && ok; i++) ok = cons_compatible(e, L + (size_t)i * words);
    free(L);
    return ok;
}
int po_engine_tree_displays(const po_engine *e, const po_tree *t) { return tree_displays(e, t); }

static po_tree *nj_build(const po_aln *a, const po_engine *ce);
po_tree *po_nj_tree(const po_aln *a) { return nj_build(a, NULL); }
po_tree *po_nj_tree_constrained(const po_engine *e) { return nj_build(e->a, e); }
static po_tree *nj_build(const po_aln *a, const po_engine *ce) {
    int n = a->ntax, N = 2 * n - 2, np = a->npat, i, j, p;
    const int constrained = ce && ce->ncons > 0, words = (n + 63) / 64;
    unsigned long long *cl = NULL, *X = NULL;          /* leaf set per node (constrained mode) */
    if (constrained) {
        cl = (unsigned long long *)calloc((size_t)N * words, sizeof(unsigned long long)); X = (unsigned long long *)calloc(words, sizeof(unsigned long long));
        for (i = 0; i < n; i++) cl[(size_t)i * words + (i >> 6)] |= 1ULL << (i & 63);
    }
    double *D = (double *)calloc((size_t)N * N, sizeof(double));
    for (i = 0; i < n; i++)
        for (j = i + 1; j < n; j++) {
            double cmp = 0, diff = 0;
            const unsigned char *ci = a->codes + (size_t)i * np, *cj = a->codes + (size_t)j * np;
            for (p = 0; p < np; p++) if (ci[p] < 20 && cj[p] < 20) { cmp += a->weight[p]; if (ci[p] != cj[p]) diff += a->weight[p]; }
            double d = 3.0;
            if (cmp > 0) { double pd = diff / cmp, x = 1.0 - pd - 0.2 * pd * pd; d = -log(x > 0.05 ? x : 0.05); }
            D[(size_t)i * N + j] = D[(size_t)j * N + i] = d;
        }
    po_tree *t = t_alloc(n);
    int *act = (int *)malloc(sizeof(int) * n), m = n, next = n;
    double *r = (double *)calloc(N, sizeof(double));
    for (i = 0; i < n; i++) act[i] = i;
#define NJCLAMP(x) ((x) < PO_TMIN ? PO_TMIN : ((x) > PO_TMAX ? PO_TMAX : (x)))
    while (m > 3) {
        for (i = 0; i < m; i++) { double s = 0; for (j = 0; j < m; j++) s += D[(size_t)act[i] * N + act[j]]; r[act[i]] = s; }
        double best = 1e300; int bi = -1, bj = -1;
        for (int pass = 0; pass < 2 && bi < 0; pass++)      /* pass 1 (only if no compatible pair exists): unconstrained */
        for (i = 0; i < m; i++) for (j = i + 1; j < m; j++) {
            int x = act[i], y = act[j];
            double q = (m - 2) * D[(size_t)x * N + y] - r[x] - r[y];
            if (q < best) {
                if (constrained && pass == 0) {
                    for (int w = 0; w < words; w++) X[w] = cl[(size_t)x * words + w] | cl[(size_t)y * words + w];
                    if (!cons_compatible(ce, X)) continue;
                }
                best = q; bi = i; bj = j;
            }
        }
        int x = act[bi], y = act[bj], u = next++;
        if (constrained) for (int w = 0; w < words; w++) cl[(size_t)u * words + w] = cl[(size_t)x * words + w] | cl[(size_t)y * words + w];
        double dxy = D[(size_t)x * N + y], lx = 0.5 * dxy + (r[x] - r[y]) / (2.0 * (m - 2));
        t_connect(t, u, x, NJCLAMP(lx)); t_connect(t, u, y, NJCLAMP(dxy - lx));
        for (i = 0; i < m; i++) { int z = act[i]; if (z == x || z == y) continue; double d = 0.5 * (D[(size_t)x * N + z] + D[(size_t)y * N + z] - dxy); D[(size_t)u * N + z] = D[(size_t)z * N + u] = d; }
        act[bi] = u; for (i = bj; i + 1 < m; i++) act[i] = act[i + 1]; m--;
    }
    {
        int x = act[0], y = act[1], z = act[2], u = next++;
        double dxy = D[(size_t)x * N + y], dxz = D[(size_t)x * N + z], dyz = D[(size_t)y * N + z];
        t_connect(t, u, x, NJCLAMP(0.5 * (dxy + dxz - dyz))); t_connect(t, u, y, NJCLAMP(0.5 * (dxy + dyz - dxz))); t_connect(t, u, z, NJCLAMP(0.5 * (dxz + dyz - dxy)));
    }
    free(D); free(act); free(r); free(cl); free(X);
    return t;
}

/* ------------------------------------------------------------------------------------------
 * topology search (spec mirrored by pepr_amd/csrc/search.cpp, DESIGN.md "Search")
 * ---------------------------------------------------------------------------------------- */
#define NNI_MIN_GAIN 0.01
typedef struct { int u, v, alt; double gain, t; int order; } nni_cand;
static int cand_cmp(const void *a, const void *b) {
    const nni_cand *x = (const nni_cand *)a, *y = (const nni_cand *)b;
    if (x->gain > y->gain) return -1; if (x->gain < y->gain) return 1;
    return x->order - y->order;
}
static void others(const po_tree *t, int v, int excl, int out[2], double len[2]) {
    int ci = 0; for (int q = 0; q < 3; q++) if (t->nbr[v][q] != excl) { out[ci] = t->nbr[v][q]; len[ci] = t->len[v][q]; ci++; }
}
/* swap subtree x (neighbour of u) with subtree y (neighbour of v); slots keep their position */
static void tree_swap(po_tree *t, int u, int x, int v, int y) {
    int ku = slot_of(t, u, x), kv = slot_of(t, v, y), kx = slot_of(t, x, u), ky = slot_of(t, y, v);
    double lx = t->len[u][ku], ly = t->len[v][kv];
    t->nbr[u][ku] = y; t->len[u][ku] = ly; t->nbr[v][kv] = x; t->len[v][kv] = lx;
    t->nbr[x][kx] = v; t->nbr[y][ky] = u;
}
static void nni_apply(po_engine *e, po_tree *t, int u, int v, int alt, double tnew) {
    int a[2], c[2]; double la[2], lc[2];
    others(t, u, v, a, la); others(t, v, u, c, lc);
    tree_swap(t, u, a[1], v, alt == 1 ? c[0] : c[1]);
    tree_set_len(t, u, v, tnew);
    mark_node(e->dirty, t, u); mark_node(e->dirty, t, v);      /* the five branches of the quartet */
}
static void tree_assign(po_tree *dst, const po_tree *src) {
    memcpy(dst->nbr, src->nbr, sizeof(int[3]) * src->nnodes); memcpy(dst->len, src->len, sizeof(double[3]) * src->nnodes);
}
static double light_smooth(po_engine *e, po_tree *t) {
    for (int pass = 0; pass < 2; pass++) if (eng_smooth(e, t, 1e-3) < 1e-3) break;     /* dirty = around the moves */
    return po_engine_lnl(e, t, NULL);
}
/* one NNI round; returns number of applied moves, updates *lnl */
static int nni_round(po_engine *e, po_tree *t, double *lnl) {
    int n = e->ntax, np = e->npat, K = e->K, ncand = 0, i;
    nni_cand *cands = (nni_cand *)malloc(sizeof(nni_cand) * (size_t)(n > 3 ? n - 3 : 1));
    double *X = (double *)malloc(sizeof(double) * (size_t)np * K * 20), *Y = (double *)malloc(sizeof(double) * (size_t)np * K * 20);
    int *xs = (int *)malloc(sizeof(int) * np), *ys = (int *)malloc(sizeof(int) * np), *sc = (int *)malloc(sizeof(int) * np);
    unsigned long long *leafs = e->ncons > 0 ? tree_leaf_sets(t) : NULL;
    for (int u = n; u < t->nnodes; u++) for (int k = 0; k < 3; k++) {
        int v = t->nbr[u][k]; if (v < n || v < u) continue;
        int a[2], c[2]; double la[2], lc[2];
        others(t, u, v, a, la); others(t, v, u, c, lc);
        double t0 = t->len[u][k], Lc, L[3], T[3];
        eng_sumtable(e, t, u, v, sc);
        (void)eng_newton_branch(e, t0, &Lc);
        for (int alt = 1; alt <= 2; alt++) {
            int y = (alt == 1) ? 0 : 1;
            side sa = eng_side(e, t, a[0], u), sb = eng_side(e, t, a[1], u), sy = eng_side(e, t, c[y], v), sz = eng_side(e, t, c[1 - y], v);
            nv_core(e, sa, sy, la[0], lc[y], X, xs);          /* new u: (a0, swapped-in child) */
            nv_core(e, sb, sz, la[1], lc[1 - y], Y, ys);      /* new v: (a1, remaining child)  */
            side SX = {X, xs, 0}, SY = {Y, ys, 0};
            sumtable_core(e, SX, SY, sc);
            T[alt] = eng_newton_branch(e, t0, &L[alt]);
            /* scaling constants differ between configurations: compare full lnL */
            { double corr = 0; for (int p = 0; p < np; p++) corr += e->a->weight[p] * sc[p]; L[alt] -= corr * PO_LOG_2_256; }
        }
        { int *s0 = (int *)malloc(sizeof(int) * np); eng_sumtable(e, t, u, v, s0); double corr = 0; for (int p = 0; p < np; p++) corr += e->a->weight[p] * s0[p]; Lc -= corr * PO_LOG_2_256; free(s0); }
        if (e->ncons > 0) {            /* an alternative whose new split violates a constraint is not a candidate */
            const int words = e->cwords;
            unsigned long long *XA = (unsigned long long *)malloc(sizeof(unsigned long long) * words), *XB = (unsigned long long *)malloc(sizeof(unsigned long long) * words);
            for (int alt = 1; alt <= 2; alt++) {
                set_of(t, leafs, words, a[0], u, XA); set_of(t, leafs, words, c[alt - 1], v, XB);
                for (int w = 0; w < words; w++) XA[w] |= XB[w];
                if (!cons_compatible(e, XA)) L[alt] = -1e300;
            }
            free(XA); free(XB);
        }
        int best = (L[2] > L[1]) ? 2 : 1;
        double gain = L[best] - Lc;
        if (gain > NNI_MIN_GAIN) { cands[ncand].u = u; cands[ncand].v = v; cands[ncand].alt = best; cands[ncand].gain = gain; cands[ncand].t = T[best]; cands[ncand].order = ncand; ncand++; }
    }
    free(X); free(Y); free(xs); free(ys); free(sc); free(leafs);
    int applied = 0;
    if (ncand > 0) {
        qsort(cands, ncand, sizeof(nni_cand), cand_cmp);
        char *used = (char *)calloc(t->nnodes, 1);
        po_tree *backup = po_tree_copy(t);
        double lnl0 = *lnl;
        memset(e->dirty, 0, (size_t)e->nnodes * 3);
        for (i = 0; i < ncand; i++) {
            if (used[cands[i].u] || used[cands[i].v]) continue;
            used[cands[i].u] = used[cands[i].v] = 1;
            nni_apply(e, t, cands[i].u, cands[i].v, cands[i].alt, cands[i].t); applied++;
        }
        eng_invalidate_all(e);
        double l1 = light_smooth(e, t);
        if (!(l1 > lnl0 + 1e-6)) {                     /* combined moves did not help: best one only */
            tree_assign(t, backup); eng_invalidate_all(e);
            memset(e->dirty, 0, (size_t)e->nnodes * 3);
            nni_apply(e, t, cands[0].u, cands[0].v, cands[0].alt, cands[0].t); applied = 1;
            l1 = light_smooth(e, t);
            if (!(l1 > lnl0 + 1e-6)) { tree_assign(t, backup); eng_invalidate_all(e); applied = 0; l1 = lnl0; }
        }
        *lnl = l1;
        po_tree_free(backup); free(used);
    }
    free(cands);
    return applied;
}

/* ---- SH-like local supports (FastTree 2.1 `SHSupport`, the default output of `FastTree_WAG -gamma` without
 * -nosupport, reference call site FastTreeRunner.java:67-70; Guindon et al. 2010).  For every internal edge in
 * nni_round's order: per-pattern lnL of the current arrangement and of its two NNI alternatives, each with its central
 * branch Newton-optimised (tolerance 1e-8) and nothing else re-optimised; nboot resamples of the alignment columns
 * drawn by the counter hash col(r, j) = mix64((seed+1)*0x9E3779B97F4A7C15 + r*nsites + j) % nsites; sums are centred
 * on the original totals; a resample supports the split when (best - second best) < observed delta = L0 - max(L1, L2);
 * delta <= 0 gives support 0.  PARITY UNPINNED vs the FastTree binary (its RNG stream cannot be reproduced). ---- */
static unsigned long long mix64(unsigned long long z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    return z;
}
static void sumtab_pat_lnl(po_engine *e, double tt, const int *scale, double *out) {
    int K = e->K, np = e->npat;
    double ex[PO_MAXCAT][PO_NS];
    for (int c = 0; c < K; c++) for (int i = 0; i < 20; i++) ex[c][i] = exp(e->m->eval[i] * e->rates[c] * tt);
    for (int p = 0; p < np; p++) {
        const double *s = e->sumtab + (size_t)p * K * 20; double f = 0;
        for (int c = 0; c < K; c++) for (int i = 0; i < 20; i++) f += s[c * 20 + i] * ex[c][i];
        out[p] = log(f / K) - scale[p] * PO_LOG_2_256;
    }
}
int po_engine_sh_support(po_engine *e, const po_tree *t, int nboot, unsigned long long seed, double *support) {
    int n = e->ntax, np = e->npat, K = e->K, ns = e->a->nsites, nedge = 0;
    const double save = e->ntol; e->ntol = 1e-8;
    (void)po_engine_lnl(e, t, NULL);
    double *X = (double *)malloc(sizeof(double) * (size_t)np * K * 20), *Y = (double *)malloc(sizeof(double) * (size_t)np * K * 20);
    int *xs = (int *)malloc(sizeof(int) * np), *ys = (int *)malloc(sizeof(int) * np), *sc = (int *)malloc(sizeof(int) * np);
    double *l[3]; for (int i = 0; i < 3; i++) l[i] = (double *)malloc(sizeof(double) * np);
    for (int u = n; u < t->nnodes; u++) for (int k = 0; k < 3; k++) {
        int v = t->nbr[u][k]; if (v < n || v < u) continue;
        int a[2], c[2]; double la[2], lc[2];
        others(t, u, v, a, la); others(t, v, u, c, lc);
        double t0 = t->len[u][k], tt;
        eng_sumtable(e, t, u, v, sc); tt = eng_newton_branch(e, t0, NULL); sumtab_pat_lnl(e, tt, sc, l[0]);
        for (int alt = 1; alt <= 2; alt++) {
            int y = (alt == 1) ? 0 : 1;
            side sa = eng_side(e, t, a[0], u), sb = eng_side(e, t, a[1], u), sy = eng_side(e, t, c[y], v), sz = eng_side(e, t, c[1 - y], v);
            nv_core(e, sa, sy, la[0], lc[y], X, xs);
            nv_core(e, sb, sz, la[1], lc[1 - y], Y, ys);
            side SX = {X, xs, 0}, SY = {Y, ys, 0};
            sumtable_core(e, SX, SY, sc); tt = eng_newton_branch(e, t0, NULL); sumtab_pat_lnl(e, tt, sc, l[alt]);
        }
        double orig[3] = {0, 0, 0};
        for (int j = 0; j < ns; j++) { int p = e->a->site2pat[j]; for (int i = 0; i < 3; i++) orig[i] += l[i][p]; }
        double delta = orig[0] - (orig[1] > orig[2] ? orig[1] : orig[2]);
        int cnt = 0;
        if (delta > 0) {
            unsigned long long base = (seed + 1ull) * 0x9E3779B97F4A7C15ull;
            for (int b = 0; b < nboot; b++) {
                double s[3] = {0, 0, 0};
                unsigned long long k0 = base + (unsigned long long)b * (unsigned long long)ns;
                for (int j = 0; j < ns; j++) { int p = e->a->site2pat[(int)(mix64(k0 + (unsigned long long)j) % (unsigned long long)ns)]; s[0] += l[0][p]; s[1] += l[1][p]; s[2] += l[2][p]; }
                for (int i = 0; i < 3; i++) s[i] -= orig[i];
                double best = s[0] > s[1] ? s[0] : s[1]; if (s[2] > best) best = s[2];
                double second;
                if (best == s[0]) second = s[1] > s[2] ? s[1] : s[2];
                else if (best == s[1]) second = s[0] > s[2] ? s[0] : s[2];
                else second = s[0] > s[1] ? s[0] : s[1];
                if (best - second < delta) cnt++;
            }
        }
        support[nedge++] = nboot > 0 ? (double)cnt / nboot : 0.0;
    }
    free(X); free(Y); free(xs); free(ys); free(sc); for (int i = 0; i < 3; i++) free(l[i]);
    e->ntol = save;
    return nedge;
}

/* ---- lazy SPR (spec mirrored by pepr_amd/csrc/search.cpp) ------------------------------------
 * For every inner node p and neighbour s (ascending): prune the subtree hanging off p through s
 * (p's other neighbours x,y get joined by one branch tx+ty) and try every edge within `radius`
 * edges of the pruning point: insert p in its middle (halves of the branch), keep the pendant
 * length, score WITHOUT re-optimisation.  The best candidate is applied if it beats the current
 * lnL by > 0.01; then the four touched branches are Newton-optimised ((p,s), (p,g), (p,h), (x,y))
 * and the move is kept only if the tree really improved.  Path messages of the pruned tree live
 * in one temporary CLV per depth. */
#define SPR_MIN_GAIN 0.01
#define SPR_MAX_RADIUS 6
typedef struct {
    po_engine *e; po_tree *t; int p, s, x, y, radius;
    side S; double ts;
    double *pm[SPR_MAX_RADIUS + 1]; int *ps[SPR_MAX_RADIUS + 1];   /* path message per depth */
    double *ins; int *insc;
    double best; int bg, bh;
    const unsigned long long *leafs; unsigned long long *LS, *tmp;   /* constraints: leaf sets of the tree, of the pruned subtree */
} spr_ctx;

static double spr_score(spr_ctx *c, side Mgh, int g, int h) {
    po_engine *e = c->e; const po_tree *t = c->t;
    int K = e->K, np = e->npat;
    double tgh = t->len[g][slot_of(t, g, h)];
    side Hg = eng_side(e, t, h, g);
    nv_core(e, Mgh, Hg, 0.5 * tgh, 0.5 * tgh, c->ins, c->insc);
    /* evaluate across the pendant branch: A = S (subtree), B = insertion CLV */
    double (*P)[PO_NS][PO_NS] = (double (*)[PO_NS][PO_NS])malloc(sizeof(double[PO_NS][PO_NS]) * K);
    eng_pmats(e, c->ts, P);
    double total = 0;
    for (int p = 0; p < np; p++) {
        double site = 0;
        for (int k = 0; k < K; k++) {
            const double *xa = c->S.clv ? c->S.clv + ((size_t)p * K + k) * 20 : e->tipvec[c->S.codes[p]];
            const double *xb = c->ins + ((size_t)p * K + k) * 20;
            double cat = 0;
            for (int s2 = 0; s2 < 20; s2++) { double yv = 0; for (int j = 0; j < 20; j++) yv += P[k][s2][j] * xb[j]; cat += e->m->pi[s2] * xa[s2] * yv; }
            site += cat;
        }
        site /= K;
        int sc = (c->S.scl ? c->S.scl[p] : 0) + c->insc[p];
        total += e->a->weight[p] * (log(site) - sc * PO_LOG_2_256);
    }
    free(P); e->n_evaluate++;
    return total;
}
/* candidate edge (g,h) reached with path message Mgh (message from g towards h in the pruned tree) */
static void spr_explore(spr_ctx *c, int g, int h, int depth, side Mgh) {
    const po_tree *t = c->t; po_engine *e = c->e;
    if (e->ncons > 0) {         /* regrafting beyond (g,h) turns its split into leaves(h side) + leaves(S): incompatible -> neither this edge nor anything behind it */
        set_of(t, c->leafs, e->cwords, h, g, c->tmp);
        for (int w = 0; w < e->cwords; w++) c->tmp[w] |= c->LS[w];
        if (!cons_compatible(e, c->tmp)) return;
    }
    double sc = spr_score(c, Mgh, g, h);
    if (sc > c->best) { c->best = sc; c->bg = g; c->bh = h; }
    if (h < e->ntax || depth >= c->radius) return;
    int ch[2]; double lc[2]; others(t, h, g, ch, lc);
    double tgh = t->len[g][slot_of(t, g, h)];
    for (int i = 0; i < 2; i++) {
        /* message from h towards ch[i]: combines Mgh (over the full branch g-h) and the other child */
        side O = eng_side(e, t, ch[1 - i], h);
        nv_core(e, Mgh, O, tgh, lc[1 - i], c->pm[depth], c->ps[depth]);
        side M = {c->pm[depth], c->ps[depth], 0};
        spr_explore(c, h, ch[i], depth + 1, M);
    }
}
static void spr_apply(po_tree *t, int p, int x, int y, int g, int h) {
    int kx = slot_of(t, p, x), ky = slot_of(t, p, y);
    double tx = t->len[p][kx], ty = t->len[p][ky], tgh = t->len[g][slot_of(t, g, h)];
    /* join x-y */
    int sx = slot_of(t, x, p), sy = slot_of(t, y, p);
    t->nbr[x][sx] = y; t->len[x][sx] = tx + ty; t->nbr[y][sy] = x; t->len[y][sy] = tx + ty;
    /* insert p into (g,h) */
    int sg = slot_of(t, g, h), sh = slot_of(t, h, g);
    t->nbr[g][sg] = p; t->len[g][sg] = 0.5 * tgh; t->nbr[h][sh] = p; t->len[h][sh] = 0.5 * tgh;
    t->nbr[p][kx] = g; t->len[p][kx] = 0.5 * tgh; t->nbr[p][ky] = h; t->len[p][ky] = 0.5 * tgh;
}
static void newton_edge(po_engine *e, po_tree *t, int u, int v) {
    eng_sumtable(e, t, u, v, NULL);
    double old = t->len[u][slot_of(t, u, v)], nl = eng_newton_branch(e, old, NULL);
    if (nl != old) { tree_set_len(t, u, v, nl); eng_branch_changed(e, t, u, v); }
}
static int spr_round(po_engine *e, po_tree *t, int radius, double *lnl) {
    int n = e->ntax, np = e->npat, K = e->K, moves = 0;
    if (radius > SPR_MAX_RADIUS) radius = SPR_MAX_RADIUS;
    if (n < 5) return 0;
    spr_ctx c; memset(&c, 0, sizeof c); c.e = e; c.t = t; c.radius = radius;
    for (int d = 0; d <= SPR_MAX_RADIUS; d++) { c.pm[d] = (double *)malloc(sizeof(double) * (size_t)np * K * 20); c.ps[d] = (int *)malloc(sizeof(int) * np); }
    c.ins = (double *)malloc(sizeof(double) * (size_t)np * K * 20); c.insc = (int *)malloc(sizeof(int) * np);
    unsigned long long *leafs_spr = NULL;
    c.LS = (unsigned long long *)calloc((size_t)(n + 63) / 64, sizeof(unsigned long long)); c.tmp = (unsigned long long *)calloc((size_t)(n + 63) / 64, sizeof(unsigned long long));
    for (int p = n; p < t->nnodes; p++) for (int ks = 0; ks < 3; ks++) {
        int s = t->nbr[p][ks], xy[2]; double lxy[2];
        others(t, p, s, xy, lxy);
        int x = xy[0], y = xy[1]; double tx = lxy[0], ty = lxy[1];
        c.p = p; c.s = s; c.x = x; c.y = y; c.ts = t->len[p][ks];
        c.S = eng_side(e, t, s, p);
        c.best = -1e300; c.bg = c.bh = -1;
        if (e->ncons > 0) { free(leafs_spr); leafs_spr = tree_leaf_sets(t); c.leafs = leafs_spr; set_of(t, leafs_spr, e->cwords, s, p, c.LS); }
        /* side of x: pruned-tree message from x to each child = f(message y->p over tx+ty, other child) */
        for (int sidei = 0; sidei < 2; sidei++) {
            int a = sidei == 0 ? x : y, b = sidei == 0 ? y : x;      /* explore into a's side; b is across */
            if (a < n) continue;
            side B = eng_side(e, t, b, p);
            int ch[2]; double lc[2]; others(t, a, p, ch, lc);
            for (int i = 0; i < 2; i++) {
                side O = eng_side(e, t, ch[1 - i], a);
                nv_core(e, B, O, tx + ty, lc[1 - i], c.pm[0], c.ps[0]);
                side M = {c.pm[0], c.ps[0], 0};
                spr_explore(&c, a, ch[i], 1, M);
            }
        }
        if (c.bg < 0 || !(c.best > *lnl + SPR_MIN_GAIN)) continue;
        po_tree *backup = po_tree_copy(t);
        spr_apply(t, p, x, y, c.bg, c.bh);
        eng_invalidate_all(e);
        newton_edge(e, t, p, s); newton_edge(e, t, p, c.bg); newton_edge(e, t, p, c.bh); newton_edge(e, t, x, y);
        double l1 = po_engine_lnl(e, t, NULL);
        if (l1 > *lnl + 1e-6) { *lnl = l1; moves++; }
        else { tree_assign(t, backup); eng_invalidate_all(e); }
        po_tree_free(backup);
    }
    for (int d = 0; d <= SPR_MAX_RADIUS; d++) { free(c.pm[d]); free(c.ps[d]); }
    free(c.ins); free(c.insc); free(c.LS); free(c.tmp); free(leafs_spr);
    return moves;
}

double po_engine_search(po_engine *e, po_tree **t_inout, int spr_radius, double eps) {
    if (!*t_inout) *t_inout = po_nj_tree(e->a);
    if (e->ncons > 0 && !tree_displays(e, *t_inout)) { po_tree_free(*t_inout); *t_inout = po_nj_tree_constrained(e); }   /* a start tree that violates the constraints */
    po_tree *t = *t_inout;
    eng_bind(e, t);
    e->ntol = 1e-6;                      /* candidate ranking and local moves: coarse Newton */
    double lnl = po_engine_optimize(e, t, 1, 0.1);
    for (int outer = 0; outer < 20; outer++) {
        int moves = 0;
        for (int round = 0; round < 100; round++) { int m = nni_round(e, t, &lnl); if (!m) break; moves += m; }
        if (spr_radius > 0) for (int round = 0; round < 10; round++) { int m = spr_round(e, t, spr_radius, &lnl); if (!m) break; moves += m; }
        lnl = po_engine_optimize(e, t, 1, 0.1);
        if (!moves) break;
    }
    e->ntol = 1e-8;
    return po_engine_optimize(e, t, 1, eps);
}

/* ------------------------------------------------------------------------------------------
 * Maximum-parsimony start tree (`raxmlHPC -f d -y`, reference call site RAxMLRunner.java:215-251;
 * the program's source is not in the reference).  Restates the published procedure: Fitch (1971)
 * state sets, randomised stepwise addition (each taxon goes to the edge with the smallest length
 * increase), then subtree-pruning-regrafting hill climbing within `radius` edges.  Integer
 * arithmetic only, so the HIP path must agree bit for bit.  PARITY UNPINNED against RAxML 7.2.5
 * (its random addition order cannot be reproduced without running it); pinned against an
 * independent Fitch length in tests/util.py and brute-force minimum length on tiny cases.
 *
 * Shared spec (engine: pepr_amd/csrc/parsimony.hip):
 *   F(l,r) = (l&r) ? (l&r) : (l|r);   S(tip) = po_code_mask(code);   S(v->u) = F of v's other two inputs
 *   order: identity if seed==0 else Fisher-Yates (i = n-1..1, j = splitmix64() % (i+1))
 *   start: inner node ntax joined to order[0..2]; step t adds tip order[t] with new inner node ntax+t-2
 *   edge enumeration: v ascending over present nodes, slot k ascending, neighbour u > v
 *   cost(e, X) = sum_p w_p [F(S(v->u), S(u->v)) & X == 0]; first minimum wins
 *   SPR round: prunes enumerated (v inner ascending, k ascending: subtree behind nbr[v][k]),
 *   a = nbr[v][(k+1)%3], b = nbr[v][(k+2)%3]; candidates by DFS from a (carrying S(b->v)) then from
 *   b (carrying S(a->v)), children in slot order, depth <= radius; gain = cost(orig) - cost(cand);
 *   the largest gain > 0 (first in enumeration on ties) is applied; repeat until none.
 * ------------------------------------------------------------------------------------------ */
static unsigned long long sm64(unsigned long long *s) {
    unsigned long long z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
typedef struct {
    const po_aln *a; po_tree *t; int np;
    unsigned *tip;          /* [ntax][np] */
    unsigned *msg;          /* [(v-ntax)*3+k][np] = S(v -> nbr[v][k]) */
    unsigned char *ok;
} pars_ctx;
static unsigned fitch(unsigned l, unsigned r) { unsigned x = l & r; return x ? x : (l | r); }
static const unsigned *pars_msg(pars_ctx *c, int v, int to) {
    const po_tree *t = c->t;
    if (v < t->ntax) return c->tip + (size_t)v * c->np;
    int k = slot_of(t, v, to), i = (v - t->ntax) * 3 + k;
    unsigned *out = c->msg + (size_t)i * c->np;
    if (!c->ok[i]) {
        const unsigned *l = pars_msg(c, t->nbr[v][(k + 1) % 3], v), *r = pars_msg(c, t->nbr[v][(k + 2) % 3], v);
        for (int p = 0; p < c->np; p++) out[p] = fitch(l[p], r[p]);
        c->ok[i] = 1;
    }
    return out;
}
static long long pars_cost3(const pars_ctx *c, const unsigned *l, const unsigned *r, const unsigned *x) {
    long long s = 0;
    for (int p = 0; p < c->np; p++) if (!(fitch(l[p], r[p]) & x[p])) s += c->a->weight[p];
    return s;
}
static void pars_reset(pars_ctx *c) { memset(c->ok, 0, (size_t)(c->t->nnodes - c->t->ntax) * 3); }
static long long pars_len_rec(pars_ctx *c, int v, int from) {      /* changes below v seen from `from` */
    const po_tree *t = c->t;
    if (v < t->ntax) return 0;
    int k = slot_of(t, v, from), x = t->nbr[v][(k + 1) % 3], y = t->nbr[v][(k + 2) % 3];
    long long s = pars_len_rec(c, x, v) + pars_len_rec(c, y, v);
    const unsigned *l = pars_msg(c, x, v), *r = pars_msg(c, y, v);
    for (int p = 0; p < c->np; p++) if (!(l[p] & r[p])) s += c->a->weight[p];
    return s;
}
static long long pars_length(pars_ctx *c, int root_tip) {
    int r = c->t->nbr[root_tip][0];
    long long s = pars_len_rec(c, r, root_tip);
    const unsigned *l = pars_msg(c, r, root_tip), *x = c->tip + (size_t)root_tip * c->np;
    for (int p = 0; p < c->np; p++) if (!(l[p] & x[p])) s += c->a->weight[p];
    return s;
}
long long po_parsimony_length(const po_aln *a, const po_tree *t) {
    pars_ctx c; c.a = a; c.t = (po_tree *)t; c.np = a->npat;
    c.tip = (unsigned *)malloc(sizeof(unsigned) * (size_t)a->ntax * a->npat);
    c.msg = (unsigned *)malloc(sizeof(unsigned) * (size_t)(t->nnodes - t->ntax) * 3 * a->npat);
    c.ok = (unsigned char *)calloc((size_t)(t->nnodes - t->ntax) * 3, 1);
    for (int i = 0; i < a->ntax; i++) for (int p = 0; p < a->npat; p++) c.tip[(size_t)i * a->npat + p] = po_code_mask(a->codes[(size_t)i * a->npat + p]);
    long long L = pars_length(&c, 0);
    free(c.tip); free(c.msg); free(c.ok);
    return L;
}
typedef struct { pars_ctx *c; const unsigned *P; int radius; long long base, best_gain; int bv, bk, bg, bh; int v, k; unsigned *path; } pspr;
static void pspr_explore(pspr *s, const unsigned *M0, int g, int from, int depth) {
    /* M0 = message arriving at g from the pruned side (tree without the subtree) */
    pars_ctx *c = s->c; const po_tree *t = c->t;
    if (g < t->ntax) return;
    int kf = slot_of(t, g, from);
    for (int j = 1; j <= 2; j++) {
        int h = t->nbr[g][(kf + j) % 3], o = t->nbr[g][(kf + 3 - j) % 3];
        unsigned *M1 = s->path + (size_t)depth * c->np;
        const unsigned *so = pars_msg(c, o, g);
        for (int p = 0; p < c->np; p++) M1[p] = fitch(M0[p], so[p]);
        long long cost = pars_cost3(c, M1, pars_msg(c, h, g), s->P);
        long long gain = s->base - cost;
        if (gain > s->best_gain) { s->best_gain = gain; s->bv = s->v; s->bk = s->k; s->bg = g; s->bh = h; }
        if (depth < s->radius) pspr_explore(s, M1, h, g, depth + 1);
    }
}
static void pars_spr_apply(po_tree *t, int v, int k, int g, int h) {
    int a = t->nbr[v][(k + 1) % 3], b = t->nbr[v][(k + 2) % 3];
    t->nbr[a][slot_of(t, a, v)] = b; t->nbr[b][slot_of(t, b, v)] = a;
    t->nbr[g][slot_of(t, g, h)] = v; t->nbr[h][slot_of(t, h, g)] = v;
    t->nbr[v][(k + 1) % 3] = g; t->nbr[v][(k + 2) % 3] = h;
}
po_tree *po_parsimony_tree(const po_aln *a, unsigned seed, int radius, long long *length_out, int *moves_out) {
    int n = a->ntax, np = a->npat;
    po_tree *t = t_alloc(n);
    for (int i = 0; i < t->nnodes; i++) for (int k = 0; k < 3; k++) t->len[i][k] = 0.1;
    pars_ctx c; c.a = a; c.t = t; c.np = np;
    c.tip = (unsigned *)malloc(sizeof(unsigned) * (size_t)n * np);
    c.msg = (unsigned *)malloc(sizeof(unsigned) * (size_t)(n - 2) * 3 * np);
    c.ok = (unsigned char *)calloc((size_t)(n - 2) * 3, 1);
    for (int i = 0; i < n; i++) for (int p = 0; p < np; p++) c.tip[(size_t)i * np + p] = po_code_mask(a->codes[(size_t)i * np + p]);
    int *order = (int *)malloc(sizeof(int) * n);
    for (int i = 0; i < n; i++) order[i] = i;
    if (seed) { unsigned long long st = seed; for (int i = n - 1; i >= 1; i--) { int j = (int)(sm64(&st) % (unsigned long long)(i + 1)); int x = order[i]; order[i] = order[j]; order[j] = x; } }
    for (int k = 0; k < 3; k++) { t->nbr[n][k] = order[k]; t->nbr[order[k]][0] = n; }
    for (int s = 3; s < n; s++) {
        int x = order[s], w = n + s - 2;
        const unsigned *X = c.tip + (size_t)x * np;
        pars_reset(&c);
        long long best = -1; int bv = -1, bu = -1;
        for (int v = 0; v < t->nnodes; v++) {
            if (t->nbr[v][0] < 0) continue;
            for (int k = 0; k < 3; k++) {
                int u = t->nbr[v][k];
                if (u < 0 || u < v) continue;
                long long cost = pars_cost3(&c, pars_msg(&c, v, u), pars_msg(&c, u, v), X);
                if (best < 0 || cost < best) { best = cost; bv = v; bu = u; }
            }
        }
        t->nbr[bv][slot_of(t, bv, bu)] = w; t->nbr[bu][slot_of(t, bu, bv)] = w;
        t->nbr[w][0] = bv; t->nbr[w][1] = bu; t->nbr[w][2] = x; t->nbr[x][0] = w;
    }
    int moves = 0;
    if (radius > 0 && n > 4) {
        pspr s; s.c = &c; s.radius = radius; s.path = (unsigned *)malloc(sizeof(unsigned) * (size_t)(radius + 2) * np);
        for (int round = 0; round < 20 * n; round++) {
            pars_reset(&c);
            s.best_gain = 0;
            for (int v = n; v < t->nnodes; v++) for (int k = 0; k < 3; k++) {
                int sub = t->nbr[v][k], x = t->nbr[v][(k + 1) % 3], y = t->nbr[v][(k + 2) % 3];
                if (x < n && y < n) continue;
                s.P = pars_msg(&c, sub, v); s.v = v; s.k = k;
                s.base = pars_cost3(&c, pars_msg(&c, x, v), pars_msg(&c, y, v), s.P);
                pspr_explore(&s, pars_msg(&c, y, v), x, v, 1);
                pspr_explore(&s, pars_msg(&c, x, v), y, v, 1);
            }
            if (s.best_gain <= 0) break;
            pars_spr_apply(t, s.bv, s.bk, s.bg, s.bh); moves++;
        }
        free(s.path);
    }
    pars_reset(&c);
    if (length_out) *length_out = pars_length(&c, 0);
    if (moves_out) *moves_out = moves;
    free(order); free(c.tip); free(c.msg); free(c.ok);
    return t;
}
