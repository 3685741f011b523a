// host.hpp -- host-side model / alignment / tree types of libpeprml (no device code here).
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>

namespace pml {

constexpr double TMIN = 1.0e-6;     // RAxML 7.2.5 zmax = 1 - 1e-6
constexpr double TMAX = 34.5;       // RAxML 7.2.5 zmin = 1e-15
constexpr double ALPHA_MIN = 0.02;
constexpr double ALPHA_MAX = 1000.0;

struct Model {
    double pi[20];
    double Q[400];
    double eval[20];
    double U[400];      // P(t) = U diag(exp(eval t)) Uinv
    double Uinv[400];
    void init(int pi_mode);
    void init_pi(const double *pi20);    // WAG exchangeabilities with the given frequencies (PROTGAMMAWAGF)
};
struct EncodedAlignment;
// empirical amino-acid frequencies, RAxML's "F" models (spec: oracle/pml_oracle.c po_empirical_freqs)
void empirical_freqs(const EncodedAlignment &a, double *pi20);

// mean rates of K equal-probability Gamma(alpha, mean 1) bins (Yang 1994); K==1 -> {1}
void gamma_rates(double alpha, int K, double *rates);

int aa_code(int ch);                 // 0..19, 20 = B, 21 = Z, 22 = gap/unknown

struct Tree {
    int ntax = 0;
    std::vector<std::array<int, 3>> nbr;       // -1 = unused (tips use slot 0 only)
    std::vector<std::array<double, 3>> len;
    int nnodes() const { return (int)nbr.size(); }
    int slot(int v, int w) const { for (int k = 0; k < 3; ++k) if (nbr[v][k] == w) return k; return -1; }
    void set_len(int u, int v, double l) { len[u][slot(u, v)] = l; len[v][slot(v, u)] = l; }
    double length() const;
    // names[i] is the label of tip i; returns false and fills err on failure
    static bool parse(const char *newick, const std::vector<std::string> &names, Tree &out, std::string &err);
    // parse without an alignment: tips numbered in order of appearance, names returned
    static bool parse_free(const char *newick, std::vector<std::string> &names, Tree &out, std::string &err);
    std::string newick(const std::vector<std::string> &names, int digits) const;
    // labels[v] (v >= ntax) is printed after the ')' of inner node v when >= 0 (support values);
    // the label belongs to the branch between v and its parent in the printed orientation
    std::string newick_labeled(const std::vector<std::string> &names, int digits, const std::vector<std::vector<int>> &edge_label) const;
    // same with real-valued labels printed with label_digits decimals (FastTree's 0-1 supports); < 0 = no label
    std::string newick_labeled(const std::vector<std::string> &names, int digits, const std::vector<std::vector<double>> &edge_label, int label_digits) const;
};

int rf_distance(const Tree &a, const Tree &b);    // (|A|+|B|-2|A&B|)/2 over non-trivial splits
// counts[u][k] = number of `others` containing the bipartition of main's internal edge (u, nbr[u][k]); -1 elsewhere
std::vector<std::vector<int>> support_counts(const Tree &main, const std::vector<Tree> &others);

// PhylogeneticTreeRefiner.getNextIndexToRefine / AdvancedTree.getMeanDescendantSupportValues on a rooted
// support-labelled Newick: ingroup = comma-joined sorted leaves of the next clade to refine ("" = none);
// done = clades already refined (same format); mean_support per node in order of appearance
bool refine_query(const char *newick, int cutoff, const std::vector<std::string> &done, std::string &ingroup,
                  std::vector<int> &mean_support, std::string &err);

struct EncodedAlignment {
    int ntax = 0, nsites = 0, npat = 0, mpad = 0;
    std::vector<std::string> names;
    std::vector<uint8_t> codes;     // [ntax][mpad], padding = gap code
    std::vector<double> weight;     // [mpad], padding = 0
    std::vector<int> site2pat;      // [nsites]
    // rows: ntax pointers to nsites chars; identical columns are merged (first-occurrence order)
    bool encode(int ntax, int nsites, const char *const *names, const char *const *rows, std::string &err);
};

// a constrained split over the gene's taxa: side '1' and side '0' as bitsets ('-' taxa in neither)
struct Constraint { std::vector<uint64_t> one, zero; };
// X (bitset of a cluster / clade) is compatible with the split: misses one side or contains one side
bool split_compatible(const Constraint &c, const std::vector<uint64_t> &X);
bool compatible_with_all(const std::vector<Constraint> &cs, const std::vector<uint64_t> &X);
// leaf sets of all directed messages: L[(v-ntax)*3+k] = taxa on v's side of edge (v, nbr[v][k])
std::vector<std::vector<uint64_t>> leaf_sets(const Tree &t);
bool tree_displays(const Tree &t, const std::vector<Constraint> &cs);

// NJ start tree; with constraints only joins whose cluster is compatible with every split are made
Tree nj_tree(const EncodedAlignment &a, const std::vector<Constraint> *cons = nullptr);
// the two halves of nj_tree: exact integer pair counts ([n*n]: comparable columns, differing columns), and
// Kimura distances + neighbour joining from such counts
void pair_counts(const EncodedAlignment &a, std::vector<int64_t> &cmp, std::vector<int64_t> &diff);
Tree nj_from_counts(int n, const std::vector<int64_t> &cmp, const std::vector<int64_t> &diff, const std::vector<Constraint> *cons = nullptr);

// resumable Brent minimiser on a fixed interval (same control flow as the oracle's eng_opt_alpha)
// FastTree's Gamma20 discretisation (FastTree 2.1 GammaLogLk): 20 fixed rates 0.05 * 400^(k/19);
// weight of rate k = P(mult * hi_k; alpha) - P(mult * lo_k; alpha) for a Gamma(shape alpha, mean 1) rate distribution,
// bin edges at the arithmetic midpoints of adjacent rates (first bin from 0, last to infinity)
void g20_rates(double *rates20);
void g20_weights(double alpha, double mult, double *w20);

struct Brent {
    double a, b, x, w, v, fx, fw, fv, d, e, u, tol = 1e-4;
    int iter; bool done;
    void start(double lo, double hi, double x0, double fx0, double tol_ = 1e-4);
    bool propose();            // sets u; returns false when converged
    void update(double fu);
};

}  // namespace pml
