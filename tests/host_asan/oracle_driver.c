/* The CPU oracle under AddressSanitizer + UBSan: every public entry point once on small random data. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../oracle/pml_oracle.h"

int main(void) {
    const char *names[] = {"t0", "t1", "t2", "t3", "t4", "t5", "t6"};
    const char aa[] = "ARNDCQEGHILKMFPSTWYV-?XBZ";
    char rows_buf[7][121]; const char *rows[7];
    srand(5);
    for (int i = 0; i < 7; i++) { for (int s = 0; s < 120; s++) rows_buf[i][s] = (s < 90 && i) ? rows_buf[0][s] : aa[rand() % 25]; for (int s = 0; s < 120; s++) if (rand() % 5 == 0) rows_buf[i][s] = aa[rand() % 25]; rows_buf[i][120] = 0; rows[i] = rows_buf[i]; }
    po_model m; po_model_init(&m, PO_PI_RAXML3DP);
    po_aln *a = po_aln_create(7, 120, names, rows, 1);
    char err[256];
    po_tree *t = po_tree_parse("((t0:0.1,t1:0.2):0.05,(t2:0.3,t3:0.1):0.1,(t4:0.2,(t5:0.1,t6:0.1):0.1):0.1);", a, err, sizeof err);
    if (!t) { fprintf(stderr, "parse: %s\n", err); return 1; }
    po_engine *e = po_engine_create(a, &m, 4, 0.8);
    double *site = (double *)malloc(sizeof(double) * 120);
    double l0 = po_engine_site_lnl(e, t, site), l1 = po_engine_lnl(e, t, NULL);
    if (!(l0 < 0) || (l0 - l1) * (l0 - l1) > 1e-12) return 2;
    double d1, d2, lb; po_engine_branch_derivs(e, t, 0, t->nbr[0][0], &lb, &d1, &d2);
    double lo = po_engine_optimize(e, t, 1, 1e-3);
    if (lo < l0 - 1e-9) return 3;
    po_tree *s = NULL; double ls = po_engine_search(e, &s, 5, 1e-3);
    if (!(ls >= lo - 1.0)) return 4;
    double sup[8]; int ne = po_engine_sh_support(e, s, 200, 7ull, sup);
    if (ne != 4) return 5;
    long long plen; int moves; po_tree *p = po_parsimony_tree(a, 3, 20, &plen, &moves);
    if (plen != po_parsimony_length(a, p)) return 6;
    po_tree *nj = po_nj_tree(a);
    char *nw = po_tree_newick(s, a, 8); int rf = po_tree_rf(s, nj); (void)rf;
    po_tree *c = po_tree_copy(s);
    const char *n4[] = {"a", "b", "c", "d"}; const char *r4[] = {"ARND", "ARNE", "AQND", "GRND"};
    po_aln *a4 = po_aln_create(4, 4, n4, r4, 1);
    po_tree *t4 = po_tree_parse("((a:0.1,b:0.1):0.1,c:0.1,d:0.1);", a4, err, sizeof err);
    double bf = po_bruteforce_lnl(a4, &m, 4, 1.0, t4);
    po_engine *e4 = po_engine_create(a4, &m, 4, 1.0);
    double pr = po_engine_lnl(e4, t4, NULL);
    if ((bf - pr) * (bf - pr) > 1e-16 * bf * bf) return 7;
    double rates[4]; po_gamma_rates(0.5, 4, 0, rates);
    printf("oracle asan driver ok: lnl %.4f opt %.4f search %.4f parsimony %lld bf %.6f\n", l0, lo, ls, plen, bf);
    free(nw); free(site);
    po_tree_free(c); po_tree_free(nj); po_tree_free(p); po_tree_free(s); po_tree_free(t); po_tree_free(t4);
    po_engine_free(e); po_engine_free(e4); po_aln_free(a); po_aln_free(a4);
    return 0;
}
