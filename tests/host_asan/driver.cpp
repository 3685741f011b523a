// Host-side code of libpeprml (no device code) under AddressSanitizer + UBSan: Newick dialect, encoder, NJ,
// RF / support counting, refinement queries, constraints -- fed with valid, odd and malformed inputs.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../../pepr_amd/csrc/host.hpp"

using namespace pml;

static int checks = 0;
#define CHECK(c) do { ++checks; if (!(c)) { std::fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main() {
    std::string err;
    // ---- Newick dialect ----
    std::vector<std::string> names = {"a", "b", "c", "d", "e"};
    const char *good[] = {"((a:0.1,b:0.2):0.05,c:0.3,(d:0.1,e:0.1):0.2);", "((a,b),c,(d,e))", "(((a:1,b:1)95:1,c:1)80:1,(d:1,e:1):1);",
                          "((a:1,b:1):1[95],c:1,(d:1,e:1):1[80]);", " ( ( 'a' : 1 , b:1e-3 ) , c , ( d , e ) ) ; ", "(a,b,c,d,e);", "((((a,b),c),d),e);"};
    for (const char *nw : good) {
        Tree t; CHECK(Tree::parse(nw, names, t, err));
        CHECK(t.ntax == 5 && t.nnodes() == 8);
        const std::string out = t.newick(names, 6);
        Tree t2; CHECK(Tree::parse(out.c_str(), names, t2, err));
        CHECK(rf_distance(t, t2) == 0);
        CHECK(!t.newick(names, -1).empty());
    }
    const char *bad[] = {"", "(", "((a,b),c", "(a,b,c,d,x);", "(a,b);", "((a,b),c,(d,e)));", "(a:xyz,b,c,d,e);", "(a,a,b,c,d,e);", "((a,b),c,(d,));", "(,,,,);",
                         "((((((((((((((((((((a", "(a:1,b:1,c:1,d:1,e:1", "(a,b,c,d,e)[", "'a", "(a,b,c,d,e);;;;((("};
    for (const char *nw : bad) { Tree t; (void)Tree::parse(nw, names, t, err); ++checks; }
    {   // random garbage and mutated valid strings must never crash
        std::mt19937 rng(7); const std::string base = good[2]; const char alphabet[] = "(),:;[]'ab cde0123456789.e-";
        for (int it = 0; it < 20000; ++it) {
            std::string s = base;
            const int nmut = 1 + (int)(rng() % 6);
            for (int m = 0; m < nmut; ++m) {
                const size_t pos = rng() % (s.size() + 1); const int op = (int)(rng() % 3);
                const char ch = alphabet[rng() % (sizeof alphabet - 1)];
                if (op == 0 && pos < s.size()) s[pos] = ch; else if (op == 1) s.insert(s.begin() + (long)pos, ch); else if (pos < s.size()) s.erase(s.begin() + (long)pos);
            }
            Tree t; std::vector<std::string> nm;
            if (Tree::parse(s.c_str(), names, t, err)) { (void)t.newick(names, 3); (void)t.length(); }
            Tree f; if (Tree::parse_free(s.c_str(), nm, f, err)) (void)f.newick(nm, 2);
            std::string in; std::vector<int> means; (void)refine_query(s.c_str(), 100, {}, in, means, err);
            ++checks;
        }
    }
    // ---- encoder + NJ ----
    {
        const char *nm[] = {"t0", "t1", "t2", "t3", "t4", "t5"};
        std::mt19937 rng(3); const char aa[] = "ARNDCQEGHILKMFPSTWYV-?XBZJUO*.arndc";
        for (int it = 0; it < 300; ++it) {
            const int n = 3 + (int)(rng() % 4), L = 1 + (int)(rng() % 200);
            std::vector<std::string> rows(n, std::string((size_t)L, 'A'));
            for (auto &r : rows) for (auto &c : r) c = aa[rng() % (sizeof aa - 1)];
            std::vector<const char *> rp; for (auto &r : rows) rp.push_back(r.c_str());
            EncodedAlignment e; CHECK(e.encode(n, L, nm, rp.data(), err));
            CHECK(e.npat >= 1 && e.npat <= L && e.mpad % 32 == 0 && (int)e.site2pat.size() == L);
            double w = 0; for (double x : e.weight) w += x; CHECK((int)w == L);
            Tree t = nj_tree(e); CHECK(t.ntax == n && t.nnodes() == 2 * n - 2);
            std::vector<int64_t> cmp, diff; pair_counts(e, cmp, diff);
            Tree t2 = nj_from_counts(n, cmp, diff); CHECK(rf_distance(t, t2) == 0);
            std::vector<std::string> names6(nm, nm + n);
            Tree back; CHECK(Tree::parse(t.newick(names6, 10).c_str(), names6, back, err) && rf_distance(t, back) == 0);
        }
        EncodedAlignment e; const char *r2[] = {"AR", "A"};     // ragged rows: error, not a crash (rows are NUL-terminated)
        (void)e.encode(2, 2, nm, r2, err); ++checks;
    }
    // ---- supports, constraints, refinement ----
    {
        Tree m; CHECK(Tree::parse("((a,b),c,(d,e));", names, m, err));
        std::vector<Tree> others(3);
        CHECK(Tree::parse("((a,b),c,(d,e));", names, others[0], err) && Tree::parse("((a,c),b,(d,e));", names, others[1], err) && Tree::parse("((a,b),(c,d),e);", names, others[2], err));
        auto cnt = support_counts(m, others);
        CHECK(!m.newick_labeled(names, 3, cnt).empty());
        std::vector<std::vector<double>> lab((size_t)m.nnodes(), std::vector<double>(3, 0.5));
        CHECK(!m.newick_labeled(names, 3, lab, 3).empty());
        Constraint c; c.one = {0b00011}; c.zero = {0b11100};
        CHECK(tree_displays(m, {c}));
        Constraint c2; c2.one = {0b00101}; c2.zero = {0b11010};
        CHECK(!tree_displays(m, {c2}));
        std::string in; std::vector<int> means;
        CHECK(refine_query("((a:1,b:1)100:1,((c:1,d:1)60:1,(e:1,(f:1,g:1)100:1)100:1)100:1,h:1);", 100, {}, in, means, err));
        CHECK(in == "c,d,e,f,g" && means.size() == 14);
    }
    double r[4]; gamma_rates(0.5, 4, r); CHECK(r[0] > 0.03 && r[0] < 0.04 && r[3] > 2.8 && r[3] < 3.0);
    Model mdl; mdl.init(0); double s = 0; for (double p : mdl.pi) s += p; CHECK(s > 0.999999 && s < 1.000001);
    std::printf("host asan driver: %d checks ok\n", checks);
    return 0;
}
