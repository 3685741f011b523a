// host.cpp -- model constants, alignment encoding, tree I/O (host side of libpeprml).
#include "host.hpp"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <unordered_map>

namespace pml {

// ------------------------------------------------------------------------------------------
// WAG (Whelan & Goldman 2001), amino-acid order ARNDCQEGHILKMFPSTWYV; lower triangle by rows.
// RAxML 7.2.5 PROTGAMMAWAG uses the 3-decimal frequencies, FastTree the full ones.
// ------------------------------------------------------------------------------------------
static const double kWagLower[190] = {
    0.551571, 0.509848, 0.635346, 0.738998, 0.147304, 5.429420, 1.027040, 0.528191, 0.265256, 0.0302949,
    0.908598, 3.035500, 1.543640, 0.616783, 0.0988179, 1.582850, 0.439157, 0.947198, 6.174160, 0.021352,
    5.469470, 1.416720, 0.584665, 1.125560, 0.865584, 0.306674, 0.330052, 0.567717, 0.316954, 2.137150,
    3.956290, 0.930676, 0.248972, 4.294110, 0.570025, 0.249410, 0.193335, 0.186979, 0.554236, 0.039437,
    0.170135, 0.113917, 0.127395, 0.0304501, 0.138190, 0.397915, 0.497671, 0.131528, 0.0848047, 0.384287,
    0.869489, 0.154263, 0.0613037, 0.499462, 3.170970, 0.906265, 5.351420, 3.012010, 0.479855, 0.0740339,
    3.894900, 2.584430, 0.373558, 0.890432, 0.323832, 0.257555, 0.893496, 0.683162, 0.198221, 0.103754,
    0.390482, 1.545260, 0.315124, 0.174100, 0.404141, 4.257460, 4.854020, 0.934276, 0.210494, 0.102711,
    0.0961621, 0.0467304, 0.398020, 0.0999208, 0.0811339, 0.049931, 0.679371, 1.059470, 2.115170, 0.088836,
    1.190630, 1.438550, 0.679489, 0.195081, 0.423984, 0.109404, 0.933372, 0.682355, 0.243570, 0.696198,
    0.0999288, 0.415844, 0.556896, 0.171329, 0.161444, 3.370790, 1.224190, 3.974230, 1.071760, 1.407660,
    1.028870, 0.704939, 1.341820, 0.740169, 0.319440, 0.344739, 0.967130, 0.493905, 0.545931, 1.613280,
    2.121110, 0.554413, 2.030060, 0.374866, 0.512984, 0.857928, 0.822765, 0.225833, 0.473307, 1.458160,
    0.326622, 1.386980, 1.516120, 0.171903, 0.795384, 4.378020, 0.113133, 1.163920, 0.0719167, 0.129767,
    0.717070, 0.215737, 0.156557, 0.336983, 0.262569, 0.212483, 0.665309, 0.137505, 0.515706, 1.529640,
    0.139405, 0.523742, 0.110864, 0.240735, 0.381533, 1.086000, 0.325711, 0.543833, 0.227710, 0.196303,
    0.103604, 3.873440, 0.420170, 0.398618, 0.133264, 0.428437, 6.454280, 0.216046, 0.786993, 0.291148,
    2.485390, 2.006010, 0.251849, 0.196246, 0.152335, 1.002140, 0.301281, 0.588731, 0.187247, 0.118358,
    7.821300, 1.800340, 0.305434, 2.058450, 0.649892, 0.314887, 0.232739, 1.388230, 0.365369, 0.314730};
static const double kWagPiFull[20] = {0.0866279, 0.043972, 0.0390894, 0.0570451, 0.0193078, 0.0367281, 0.0580589,
                                      0.0832518, 0.0244313, 0.048466, 0.086209, 0.0620286, 0.0195027, 0.0384319,
                                      0.0457631, 0.0695179, 0.0610127, 0.0143859, 0.0352742, 0.0708956};
static const double kWagPi3dp[20] = {0.087, 0.044, 0.039, 0.057, 0.019, 0.037, 0.058, 0.083, 0.024, 0.049,
                                     0.086, 0.062, 0.020, 0.038, 0.046, 0.070, 0.061, 0.014, 0.035, 0.071};

// symmetric eigen-solver (cyclic Jacobi rotations), eigenvalues sorted descending
static void sym_eig20(std::vector<double> &A, double *eval, std::vector<double> &V) {
    const int n = 20;
    V.assign(n * n, 0.0);
    for (int i = 0; i < n; ++i) V[i * n + i] = 1.0;
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0;
        for (int p = 0; p < n; ++p) for (int q = p + 1; q < n; ++q) off += A[p * n + q] * A[p * n + q];
        if (off < 1e-40) break;
        for (int p = 0; p < n; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[p * n + q];
                if (std::fabs(apq) < 1e-300) continue;
                const double theta = (A[q * n + q] - A[p * n + p]) / (2 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
                const double c = 1 / std::sqrt(t * t + 1), s = t * c;
                for (int i = 0; i < n; ++i) { double x = A[i * n + p], y = A[i * n + q]; A[i * n + p] = c * x - s * y; A[i * n + q] = s * x + c * y; }
                for (int i = 0; i < n; ++i) { double x = A[p * n + i], y = A[q * n + i]; A[p * n + i] = c * x - s * y; A[q * n + i] = s * x + c * y; }
                for (int i = 0; i < n; ++i) { double x = V[i * n + p], y = V[i * n + q]; V[i * n + p] = c * x - s * y; V[i * n + q] = s * x + c * y; }
            }
    }
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return A[a * n + a] > A[b * n + b]; });
    std::vector<double> V2(n * n);
    for (int k = 0; k < n; ++k) { eval[k] = A[order[k] * n + order[k]]; for (int i = 0; i < n; ++i) V2[i * n + k] = V[i * n + order[k]]; }
    V.swap(V2);
}

void Model::init(int pi_mode) { init_pi(pi_mode == 1 ? kWagPiFull : kWagPi3dp); }
void Model::init_pi(const double *pi20) {
    double S[400] = {0};
    int k = 0;
    for (int i = 1; i < 20; ++i) for (int j = 0; j < i; ++j) { S[i * 20 + j] = S[j * 20 + i] = kWagLower[k++]; }
    double sum = 0;
    for (int i = 0; i < 20; ++i) { pi[i] = pi20[i]; sum += pi[i]; }
    for (int i = 0; i < 20; ++i) pi[i] /= sum;
    double mu = 0;
    for (int i = 0; i < 20; ++i) {
        double row = 0;
        for (int j = 0; j < 20; ++j) if (j != i) { Q[i * 20 + j] = S[i * 20 + j] * pi[j]; row += Q[i * 20 + j]; }
        Q[i * 20 + i] = -row; mu += pi[i] * row;
    }
    for (int i = 0; i < 400; ++i) Q[i] /= mu;
    std::vector<double> B(400), V;
    for (int i = 0; i < 20; ++i) for (int j = 0; j < 20; ++j) B[i * 20 + j] = std::sqrt(pi[i]) * Q[i * 20 + j] / std::sqrt(pi[j]);
    for (int i = 0; i < 20; ++i) for (int j = i + 1; j < 20; ++j) B[i * 20 + j] = B[j * 20 + i] = 0.5 * (B[i * 20 + j] + B[j * 20 + i]);
    sym_eig20(B, eval, V);
    for (int i = 0; i < 20; ++i) for (int j = 0; j < 20; ++j) {
        U[i * 20 + j] = V[i * 20 + j] / std::sqrt(pi[i]);
        Uinv[j * 20 + i] = V[i * 20 + j] * std::sqrt(pi[i]);
    }
}

// regularised lower incomplete gamma P(a,x)
static double inc_gamma(double a, double x) {
    if (x <= 0) return 0;
    const double gln = std::lgamma(a);
    if (x < a + 1) {
        double ap = a, sum = 1 / a, del = sum;
        for (int n = 0; n < 100000; ++n) { ap += 1; del *= x / ap; sum += del; if (std::fabs(del) < std::fabs(sum) * 1e-17) break; }
        return sum * std::exp(-x + a * std::log(x) - gln);
    }
    const double tiny = 1e-300;
    double b = x + 1 - a, c = 1 / tiny, d = 1 / b, h = d;
    for (int i = 1; i < 100000; ++i) {
        const double an = -i * (i - a); b += 2;
        d = an * d + b; if (std::fabs(d) < tiny) d = tiny;
        c = b + an / c; if (std::fabs(c) < tiny) c = tiny;
        d = 1 / d; const double del = d * c; h *= del;
        if (std::fabs(del - 1) < 1e-16) break;
    }
    return 1 - std::exp(-x + a * std::log(x) - gln) * h;
}
// quantile of Gamma(shape a, scale 1): Newton on y = log x inside the bisection bracket (a step that leaves the bracket
// falls back to its midpoint).  Same root as plain bisection to ~1e-15 relative, in ~6 instead of ~60 evaluations of
// P(a, x) -- the alpha optimisation calls this three times per gene and Brent step while the device waits.
static double gamma_quantile(double p, double a) {
    double lo = -1600, hi = std::log(a + 40 * std::sqrt(a) + 400);
    const double gln = std::lgamma(a);
    // start: small shapes x ~ (p Gamma(a+1))^(1/a); otherwise Wilson-Hilferty with a crude normal quantile
    double y;
    if (a < 1.0) y = (std::log(p) + std::lgamma(a + 1)) / a;
    else {
        const double z = (p < 0.5 ? -1.0 : 1.0) * std::sqrt(-2.0 * std::log(p < 0.5 ? p : 1 - p)) * 0.6, c = 1.0 / (9.0 * a);
        const double w = 1.0 - c + z * std::sqrt(c);
        y = std::log(a) + 3.0 * std::log(w > 0.05 ? w : 0.05);
    }
    if (!(y > lo && y < hi)) y = 0.5 * (lo + hi);
    for (int i = 0; i < 200; ++i) {
        const double x = std::exp(y), F = inc_gamma(a, x) - p;
        if (F < 0) lo = y; else hi = y;
        const double dFdy = std::exp(a * y - x - gln);            // density * x
        double yn = (dFdy > 0 && std::isfinite(dFdy)) ? y - F / dFdy : 0.5 * (lo + hi);
        if (!(yn > lo && yn < hi)) yn = 0.5 * (lo + hi);
        const bool done = std::fabs(yn - y) <= 4e-16 * std::max(1.0, std::fabs(y)) || hi - lo < 1e-15 * std::max(1.0, std::fabs(y));
        y = yn;
        if (done) break;
    }
    return std::exp(y);
}
void g20_rates(double *r) { for (int k = 0; k < 20; ++k) r[k] = 0.05 * std::pow(400.0, k / 19.0); }
void g20_weights(double alpha, double mult, double *w) {
    double r[20]; g20_rates(r);
    double prev = 0.0;
    for (int k = 0; k < 20; ++k) {
        const double cur = (k == 19) ? 1.0 : inc_gamma(alpha, mult * 0.5 * (r[k] + r[k + 1]) * alpha);      // shape alpha, rate alpha: mean 1
        w[k] = cur - prev; prev = cur;
    }
}
void gamma_rates(double alpha, int K, double *rates) {
    if (K <= 1) { rates[0] = 1.0; return; }
    double prev = 0;
    for (int i = 0; i < K; ++i) {
        const double cur = (i == K - 1) ? 1.0 : inc_gamma(alpha + 1, gamma_quantile((i + 1.0) / K, alpha));
        rates[i] = (cur - prev) * K; prev = cur;
    }
}

int aa_code(int ch) {
    struct Table {
        int8_t t[256];
        Table() {
            std::memset(t, 22, sizeof t);
            const char *aa = "ARNDCQEGHILKMFPSTWYV";
            for (int i = 0; i < 20; ++i) { t[(unsigned char)aa[i]] = (int8_t)i; t[(unsigned char)std::tolower(aa[i])] = (int8_t)i; }
            t['B'] = t['b'] = 20; t['Z'] = t['z'] = 21;
        }
    };
    static const Table table;          // thread-safe initialisation (genes are encoded on several threads)
    return table.t[(unsigned char)ch];
}

static unsigned host_code_mask(int code) { return code < 20 ? (1u << code) : (code == 20 ? 0xCu : (code == 21 ? 0x60u : 0xFFFFFu)); }
// PROTGAMMAWAGF: eight sweeps of proportional counting from 1/20, then a floor of 0.001 (oracle: po_empirical_freqs)
void empirical_freqs(const EncodedAlignment &a, double *pi) {
    double f[20], acc[20];
    for (int l = 0; l < 20; ++l) f[l] = 0.05;
    for (int sweep = 0; sweep < 8; ++sweep) {
        for (int l = 0; l < 20; ++l) acc[l] = 0.0;
        for (int i = 0; i < a.ntax; ++i) for (int p = 0; p < a.npat; ++p) {
            const unsigned mk = host_code_mask(a.codes[(size_t)i * a.mpad + p]);
            double sum = 0.0;
            for (int l = 0; l < 20; ++l) if ((mk >> l) & 1) sum += f[l];
            const double wj = a.weight[p] / sum;
            for (int l = 0; l < 20; ++l) if ((mk >> l) & 1) acc[l] += wj * f[l];
        }
        double tot = 0.0;
        for (int l = 0; l < 20; ++l) tot += acc[l];
        for (int l = 0; l < 20; ++l) f[l] = acc[l] / tot;
    }
    for (int round = 0; round < 100; ++round) {
        double lift = 0.0, big = 0.0; int low = 0;
        for (int l = 0; l < 20; ++l) { if (f[l] < 0.001) { lift += 0.001 - f[l]; ++low; } else big += f[l]; }
        if (!low) break;
        for (int l = 0; l < 20; ++l) f[l] = f[l] < 0.001 ? 0.001 : f[l] * (1.0 - lift / big);
    }
    for (int l = 0; l < 20; ++l) pi[l] = f[l];
}

bool EncodedAlignment::encode(int nt, int ns, const char *const *nm, const char *const *rows, std::string &err) {
    if (nt < 3) { err = "alignment needs at least 3 taxa"; return false; }
    if (ns < 0 || !nm || !rows) { err = "bad alignment"; return false; }
    ntax = nt; nsites = ns;
    names.clear();
    for (int i = 0; i < nt; ++i) {
        if (!nm[i] || !rows[i]) { err = "null name or row"; return false; }
        names.emplace_back(nm[i]);
        if ((int)strnlen(rows[i], (size_t)ns) < ns) { err = "row " + std::to_string(i) + " shorter than nsites"; return false; }
    }
    std::vector<uint8_t> col((size_t)ns * nt);
    for (int i = 0; i < nt; ++i) { const char *r = rows[i]; for (int s = 0; s < ns; ++s) col[(size_t)s * nt + i] = (uint8_t)aa_code(r[s]); }
    site2pat.assign(ns, 0);
    std::vector<int> first; std::vector<int> wt;
    std::unordered_multimap<uint64_t, int> seen; seen.reserve((size_t)ns * 2);
    for (int s = 0; s < ns; ++s) {
        const uint8_t *c = &col[(size_t)s * nt];
        uint64_t h = 1469598103934665603ULL;
        for (int i = 0; i < nt; ++i) { h ^= c[i]; h *= 1099511628211ULL; }
        int pat = -1;
        auto range = seen.equal_range(h);
        for (auto it = range.first; it != range.second; ++it)
            if (std::memcmp(&col[(size_t)first[it->second] * nt], c, nt) == 0) { pat = it->second; break; }
        if (pat < 0) { pat = (int)first.size(); first.push_back(s); wt.push_back(0); seen.emplace(h, pat); }
        site2pat[s] = pat; wt[pat]++;
    }
    npat = (int)first.size();
    mpad = std::max(32, (npat + 31) / 32 * 32);
    codes.assign((size_t)nt * mpad, 22);
    weight.assign(mpad, 0.0);
    for (int p = 0; p < npat; ++p) { weight[p] = wt[p]; for (int i = 0; i < nt; ++i) codes[(size_t)i * mpad + p] = col[(size_t)first[p] * nt + i]; }
    return true;
}

// ------------------------------------------------------------------------------------------
// Newick (dialect of BasicTree.parseNewickTreeString, reference BasicTree.java:131-409: inner
// labels or [..] comments as supports, optional ';', rooted input is unrooted on read)
// ------------------------------------------------------------------------------------------
namespace {
struct RNode { std::vector<int> kids; double len = 0.1; bool haslen = false; std::string label; };
struct Parser {
    const char *s; size_t pos = 0; std::vector<RNode> nodes; std::string err, comment;
    void ws() {
        for (;;) {
            while (s[pos] && std::isspace((unsigned char)s[pos])) ++pos;
            if (s[pos] == '[') { const size_t b = pos + 1; while (s[pos] && s[pos] != ']') ++pos; comment.assign(s + b, pos - b); if (s[pos]) ++pos; } else break;
        }
    }
    bool fail(const std::string &m) { if (err.empty()) err = m + " (at char " + std::to_string(pos) + ")"; return false; }
    bool label_len(int id) {
        ws();
        size_t b = pos, e;
        if (s[pos] == '\'') { b = ++pos; while (s[pos] && s[pos] != '\'') ++pos; e = pos; if (s[pos]) ++pos; }
        else { while (s[pos] && !std::strchr(",():;[", s[pos]) && !std::isspace((unsigned char)s[pos])) ++pos; e = pos; }
        nodes[id].label.assign(s + b, e - b);
        ws();
        if (s[pos] == ':') {
            ++pos; ws();
            char *end; const double v = std::strtod(s + pos, &end);
            if (end == s + pos) return fail("bad branch length");
            pos = (size_t)(end - s); nodes[id].len = v; nodes[id].haslen = true;
        }
        comment.clear(); ws();
        if (nodes[id].label.empty() && !comment.empty()) nodes[id].label = comment;      // "...:0.4[95]" support form
        return true;
    }
    int subtree(int depth) {
        if (depth > 100000) { fail("tree too deep"); return -1; }
        ws();
        const int id = (int)nodes.size(); nodes.emplace_back();
        if (s[pos] == '(') {
            ++pos;
            for (;;) {
                const int c = subtree(depth + 1); if (c < 0) return -1;
                nodes[id].kids.push_back(c); ws();
                if (s[pos] == ',') { ++pos; continue; }
                if (s[pos] == ')') { ++pos; break; }
                fail("expected ',' or ')'"); return -1;
            }
        }
        if (!label_len(id)) return -1;
        if (nodes[id].kids.empty() && nodes[id].label.empty()) { fail("empty leaf name"); return -1; }
        return id;
    }
};

struct Builder {
    Parser &P; Tree &T; const std::vector<int> &tipid; int next_inner;
    void connect(int a, int b, double l) {
        int k = 0; while (k < 3 && T.nbr[a][k] >= 0) ++k;
        int m = 0; while (m < 3 && T.nbr[b][m] >= 0) ++m;
        T.nbr[a][k] = b; T.len[a][k] = l; T.nbr[b][m] = a; T.len[b][m] = l;
    }
    // returns node id of the (binary-resolved) subtree rooted at parsed node r; extra receives the
    // length accumulated through unary nodes
    int build(int r, double &uplen) {
        uplen = P.nodes[r].len;
        while (P.nodes[r].kids.size() == 1) { r = P.nodes[r].kids[0]; uplen += P.nodes[r].len; }
        if (P.nodes[r].kids.empty()) return tipid[r];
        std::vector<int> ids; std::vector<double> ls;
        for (int c : P.nodes[r].kids) { double l; ids.push_back(build(c, l)); ls.push_back(l); }
        int cur = ids[0]; double curl = ls[0];
        for (size_t i = 1; i + 1 < ids.size(); ++i) {       // resolve polytomies with TMIN branches
            const int nid = next_inner++;
            connect(nid, cur, curl); connect(nid, ids[i], ls[i]);
            cur = nid; curl = TMIN;
        }
        const int id = next_inner++;
        connect(id, cur, curl); connect(id, ids.back(), ls.back());
        return id;
    }
};
}  // namespace

static bool build_tree(Parser &P, int root, const std::vector<int> &tipid, int ntax, Tree &out, std::string &err) {
    if (ntax < 3) { err = "need at least 3 taxa"; return false; }
    out.ntax = ntax;
    out.nbr.assign(2 * ntax - 2, {-1, -1, -1});
    out.len.assign(2 * ntax - 2, {0.0, 0.0, 0.0});
    while (P.nodes[root].kids.size() == 1) root = P.nodes[root].kids[0];
    Builder B{P, out, tipid, ntax};
    auto &kids = P.nodes[root].kids;
    if (kids.size() < 2) { err = "tree has a single leaf"; return false; }
    std::vector<int> ids; std::vector<double> ls;
    for (int c : kids) { double l; ids.push_back(B.build(c, l)); ls.push_back(l); }
    if (ids.size() == 2) B.connect(ids[0], ids[1], ls[0] + ls[1]);
    else {
        int cur = ids[0]; double curl = ls[0];
        for (size_t i = 1; i + 2 < ids.size(); ++i) {
            const int nid = B.next_inner++;
            B.connect(nid, cur, curl); B.connect(nid, ids[i], ls[i]);
            cur = nid; curl = TMIN;
        }
        const int id = B.next_inner++;
        B.connect(id, cur, curl); B.connect(id, ids[ids.size() - 2], ls[ids.size() - 2]); B.connect(id, ids.back(), ls.back());
    }
    if (B.next_inner != 2 * ntax - 2) { err = "internal: node count mismatch"; return false; }
    for (auto &l : out.len) for (double &x : l) { if (!(x >= 0)) x = 0; if (x > TMAX) x = TMAX; }
    return true;
}

bool Tree::parse(const char *newick, const std::vector<std::string> &names, Tree &out, std::string &err) {
    if (!newick) { err = "null newick"; return false; }
    Parser P{newick};
    const int root = P.subtree(0);
    if (root < 0) { err = P.err; return false; }
    std::unordered_map<std::string, int> idx;
    for (size_t i = 0; i < names.size(); ++i) idx[names[i]] = (int)i;
    std::vector<int> tipid(P.nodes.size(), -1); std::vector<char> seen(names.size(), 0);
    for (size_t i = 0; i < P.nodes.size(); ++i) if (P.nodes[i].kids.empty()) {
        auto it = idx.find(P.nodes[i].label);
        if (it == idx.end()) { err = "leaf '" + P.nodes[i].label + "' not in alignment"; return false; }
        if (seen[it->second]) { err = "duplicate leaf '" + P.nodes[i].label + "'"; return false; }
        seen[it->second] = 1; tipid[i] = it->second;
    }
    for (size_t i = 0; i < names.size(); ++i) if (!seen[i]) { err = "taxon '" + names[i] + "' missing from tree"; return false; }
    return build_tree(P, root, tipid, (int)names.size(), out, err);
}

bool Tree::parse_free(const char *newick, std::vector<std::string> &names, Tree &out, std::string &err) {
    if (!newick) { err = "null newick"; return false; }
    Parser P{newick};
    const int root = P.subtree(0);
    if (root < 0) { err = P.err; return false; }
    names.clear();
    std::vector<int> tipid(P.nodes.size(), -1);
    for (size_t i = 0; i < P.nodes.size(); ++i) if (P.nodes[i].kids.empty()) { tipid[i] = (int)names.size(); names.push_back(P.nodes[i].label); }
    return build_tree(P, root, tipid, (int)names.size(), out, err);
}

double Tree::length() const {
    double s = 0;
    for (int i = 0; i < nnodes(); ++i) for (int k = 0; k < 3; ++k) if (nbr[i][k] > i) s += len[i][k];
    return s;
}

std::string Tree::newick(const std::vector<std::string> &names, int digits) const {
    std::string out; char buf[64];
    std::function<void(int, int, double)> rec = [&](int v, int from, double l) {
        if (v < ntax) out += names[v];
        else {
            out += '('; bool first = true;
            for (int k = 0; k < 3; ++k) { const int w = nbr[v][k]; if (w < 0 || w == from) continue; if (!first) out += ','; first = false; rec(w, v, len[v][k]); }
            out += ')';
        }
        if (digits >= 0) { std::snprintf(buf, sizeof buf, ":%.*f", digits, l); out += buf; }     // digits < 0: topology only
    };
    const int r = nbr[0][0];
    out += '('; out += names[0]; if (digits >= 0) { std::snprintf(buf, sizeof buf, ":%.*f", digits, len[0][0]); out += buf; }
    for (int k = 0; k < 3; ++k) { const int w = nbr[r][k]; if (w < 0 || w == 0) continue; out += ','; rec(w, r, len[r][k]); }
    out += ");";
    return out;
}

// bipartition of the edge (from -> v): bitset of the taxa below v, canonicalised to the side
// that does not contain taxon 0
static void collect_splits(const Tree &t, std::vector<std::vector<uint64_t>> &out, std::vector<std::pair<int, int>> *edges) {
    const int n = t.ntax, words = (n + 63) / 64;
    std::function<std::vector<uint64_t>(int, int)> rec = [&](int v, int from) {
        std::vector<uint64_t> s(words, 0);
        if (v < n) { s[v >> 6] |= 1ULL << (v & 63); return s; }
        for (int k = 0; k < 3; ++k) { const int w = t.nbr[v][k]; if (w < 0 || w == from) continue; auto c = rec(w, v); for (int i = 0; i < words; ++i) s[i] |= c[i]; }
        if (from >= n) { out.push_back(s); if (edges) edges->push_back({from, v}); }
        return s;
    };
    const int r = t.nbr[0][0];
    for (int k = 0; k < 3; ++k) { const int w = t.nbr[r][k]; if (w < 0 || w == 0) continue; rec(w, r); }
}

std::vector<std::vector<int>> support_counts(const Tree &main, const std::vector<Tree> &others) {
    std::vector<std::vector<uint64_t>> ms; std::vector<std::pair<int, int>> edges;
    collect_splits(main, ms, &edges);
    std::vector<std::vector<int>> counts(main.nnodes(), std::vector<int>(3, -1));
    std::vector<int> c(ms.size(), 0);
    for (const Tree &o : others) {
        std::vector<std::vector<uint64_t>> os; collect_splits(o, os, nullptr);
        std::sort(os.begin(), os.end());
        for (size_t i = 0; i < ms.size(); ++i) if (std::binary_search(os.begin(), os.end(), ms[i])) c[i]++;
    }
    for (size_t i = 0; i < ms.size(); ++i) {
        const int u = edges[i].first, v = edges[i].second;
        counts[u][main.slot(u, v)] = c[i]; counts[v][main.slot(v, u)] = c[i];
    }
    return counts;
}

std::string Tree::newick_labeled(const std::vector<std::string> &names, int digits, const std::vector<std::vector<int>> &lab) const {
    std::string out; char buf[64];
    std::function<void(int, int, double)> rec = [&](int v, int from, double l) {
        if (v < ntax) out += names[v];
        else {
            out += '('; bool first = true;
            for (int k = 0; k < 3; ++k) { const int w = nbr[v][k]; if (w < 0 || w == from) continue; if (!first) out += ','; first = false; rec(w, v, len[v][k]); }
            out += ')';
            const int c = lab[v][slot(v, from)];
            if (c >= 0) out += std::to_string(c);
        }
        std::snprintf(buf, sizeof buf, ":%.*f", digits, l); out += buf;
    };
    const int r = nbr[0][0];
    out += '('; out += names[0]; std::snprintf(buf, sizeof buf, ":%.*f", digits, len[0][0]); out += buf;
    for (int k = 0; k < 3; ++k) { const int w = nbr[r][k]; if (w < 0 || w == 0) continue; out += ','; rec(w, r, len[r][k]); }
    out += ");";
    return out;
}

// Progressive-refinement queries on a rooted, support-labelled Newick exactly as the Java side holds it
// (node order = order of appearance = AdvancedTree's preorder sequence, AdvancedTree.java:184-203).
bool refine_query(const char *newick, int cutoff, const std::vector<std::string> &done, std::string &ingroup,
                  std::vector<int> &mean_support, std::string &err) {
    Parser P{newick};
    const int root = P.subtree(0);
    if (root < 0) { err = P.err; return false; }
    const int n = (int)P.nodes.size();
    std::vector<int> bs(n, 100);                                  // AdvancedTree.getBranchSupports :484-506
    for (int v = 0; v < n; ++v) {
        const RNode &nd = P.nodes[v];
        if (nd.kids.empty() || nd.label.empty()) continue;
        char *end; const long iv = std::strtol(nd.label.c_str(), &end, 10);
        if (*end == 0) bs[v] = (int)iv;
        else { const double d = std::strtod(nd.label.c_str(), &end); if (*end != 0) { err = "support label '" + nd.label + "' is not a number"; return false; } bs[v] = (int)(d * 100); }
    }
    std::vector<long long> sum(n, 0), cnt(n, 0);
    mean_support.assign(n, 0);
    std::vector<std::vector<std::string>> leaves(n);
    for (int v = n - 1; v >= 0; --v) {                            // getMeanDescendantSupportValues :1061-1098
        const RNode &nd = P.nodes[v];
        if (nd.kids.empty()) { leaves[v].push_back(nd.label); continue; }
        for (int c : nd.kids) { sum[v] += sum[c] + bs[c]; cnt[v] += cnt[c] + 1; leaves[v].insert(leaves[v].end(), leaves[c].begin(), leaves[c].end()); }
        mean_support[v] = cnt[v] ? (int)std::floor((double)sum[v] / (double)cnt[v]) : 0;
    }
    ingroup.clear();
    for (int v = 2; v < n; ++v) {                                 // getNextIndexToRefine :298-359 (0 = root, 1 = its first child)
        const RNode &nd = P.nodes[v];
        if (!(mean_support[v] < cutoff && bs[v] >= cutoff && (int)leaves[v].size() >= 3)) continue;
        bool all_full = true;
        for (int c : nd.kids) if (bs[c] < cutoff) all_full = false;
        if (all_full) continue;
        std::vector<std::string> l = leaves[v]; std::sort(l.begin(), l.end());
        std::string key; for (size_t i = 0; i < l.size(); ++i) { if (i) key += ','; key += l[i]; }
        if (std::find(done.begin(), done.end(), key) != done.end()) continue;
        ingroup = key;
        break;
    }
    return true;
}

std::string Tree::newick_labeled(const std::vector<std::string> &names, int digits, const std::vector<std::vector<double>> &lab, int label_digits) const {
    std::string out; char buf[64];
    std::function<void(int, int, double)> rec = [&](int v, int from, double l) {
        if (v < ntax) out += names[v];
        else {
            out += '('; bool first = true;
            for (int k = 0; k < 3; ++k) { const int w = nbr[v][k]; if (w < 0 || w == from) continue; if (!first) out += ','; first = false; rec(w, v, len[v][k]); }
            out += ')';
            const double c = lab[v][slot(v, from)];
            if (c >= 0) { std::snprintf(buf, sizeof buf, "%.*f", label_digits, c); out += buf; }
        }
        std::snprintf(buf, sizeof buf, ":%.*f", digits, l); out += buf;
    };
    const int r = nbr[0][0];
    out += '('; out += names[0]; std::snprintf(buf, sizeof buf, ":%.*f", digits, len[0][0]); out += buf;
    for (int k = 0; k < 3; ++k) { const int w = nbr[r][k]; if (w < 0 || w == 0) continue; out += ','; rec(w, r, len[r][k]); }
    out += ");";
    return out;
}

int rf_distance(const Tree &a, const Tree &b) {
    const int n = a.ntax, words = (n + 63) / 64;
    auto splits = [&](const Tree &t) {
        std::vector<std::vector<uint64_t>> out;
        std::function<std::vector<uint64_t>(int, int)> rec = [&](int v, int from) {
            std::vector<uint64_t> s(words, 0);
            if (v < n) { s[v >> 6] |= 1ULL << (v & 63); return s; }
            for (int k = 0; k < 3; ++k) { const int w = t.nbr[v][k]; if (w < 0 || w == from) continue; auto c = rec(w, v); for (int i = 0; i < words; ++i) s[i] |= c[i]; }
            if (from >= n) out.push_back(s);
            return s;
        };
        const int r = t.nbr[0][0];
        for (int k = 0; k < 3; ++k) { const int w = t.nbr[r][k]; if (w < 0 || w == 0) continue; rec(w, r); }
        std::sort(out.begin(), out.end());
        return out;
    };
    auto sa = splits(a), sb = splits(b);
    size_t i = 0, j = 0; int common = 0;
    while (i < sa.size() && j < sb.size()) { if (sa[i] == sb[j]) { ++common; ++i; ++j; } else if (sa[i] < sb[j]) ++i; else ++j; }
    return ((int)sa.size() + (int)sb.size() - 2 * common) / 2;
}

// ------------------------------------------------------------------------------------------
// start tree: neighbour joining on Kimura-corrected protein distances (spec in DESIGN.md
// "Start tree"; FastTree also starts from NJ -- FastTreeRunner.java:67-94 / SURVEY 3.3)
// ------------------------------------------------------------------------------------------
bool split_compatible(const Constraint &c, const std::vector<uint64_t> &X) {
    bool hit1 = false, hit0 = false, all1 = true, all0 = true;
    for (size_t i = 0; i < X.size(); ++i) {
        if (X[i] & c.one[i]) hit1 = true;
        if (X[i] & c.zero[i]) hit0 = true;
        if ((X[i] & c.one[i]) != c.one[i]) all1 = false;
        if ((X[i] & c.zero[i]) != c.zero[i]) all0 = false;
    }
    return !hit1 || !hit0 || all1 || all0;
}
bool compatible_with_all(const std::vector<Constraint> &cs, const std::vector<uint64_t> &X) {
    for (const Constraint &c : cs) if (!split_compatible(c, X)) return false;
    return true;
}
std::vector<std::vector<uint64_t>> leaf_sets(const Tree &t) {
    const int n = t.ntax, words = (n + 63) / 64;
    std::vector<std::vector<uint64_t>> L((size_t)3 * (n - 2), std::vector<uint64_t>(words, 0));
    std::vector<uint64_t> all(words, 0);
    for (int i = 0; i < n; ++i) all[i >> 6] |= 1ULL << (i & 63);
    // post-order from the inner neighbour of taxon 0: sets of messages pointing towards it, then
    // every opposite direction is the complement
    struct F { int v, from, k; };
    const int r = t.nbr[0][0];
    std::vector<F> st{{r, -1, 0}};
    std::vector<int> order;       // (v, from) pairs in pre-order, processed in reverse
    std::vector<std::pair<int, int>> pre;
    while (!st.empty()) {
        F f = st.back(); st.pop_back();
        pre.push_back({f.v, f.from});
        if (f.v < n) continue;
        for (int k = 0; k < 3; ++k) { const int w = t.nbr[f.v][k]; if (w >= 0 && w != f.from) st.push_back({w, f.v, 0}); }
    }
    auto down = [&](int v, int from) -> std::vector<uint64_t> & { return L[(v - n) * 3 + t.slot(v, from)]; };
    for (size_t i = pre.size(); i-- > 0;) {
        const int v = pre[i].first, from = pre[i].second;
        if (v < n || from < 0) continue;
        std::vector<uint64_t> &S = down(v, from);
        for (int k = 0; k < 3; ++k) {
            const int w = t.nbr[v][k];
            if (w == from) continue;
            if (w < n) S[w >> 6] |= 1ULL << (w & 63);
            else { const std::vector<uint64_t> &c = down(w, v); for (int q = 0; q < words; ++q) S[q] |= c[q]; }
        }
    }
    // the root node's three messages and all upward messages: complement of the opposite message
    for (size_t i = 0; i < pre.size(); ++i) {
        const int v = pre[i].first, from = pre[i].second;
        if (v < n) { if (from >= n) { std::vector<uint64_t> &U = L[(from - n) * 3 + t.slot(from, v)]; for (int q = 0; q < words; ++q) U[q] = all[q]; U[v >> 6] &= ~(1ULL << (v & 63)); } continue; }
        if (from < 0) continue;
        if (from >= n) { std::vector<uint64_t> &U = L[(from - n) * 3 + t.slot(from, v)]; const std::vector<uint64_t> &D = down(v, from); for (int q = 0; q < words; ++q) U[q] = all[q] & ~D[q]; }
    }
    return L;
}
bool tree_displays(const Tree &t, const std::vector<Constraint> &cs) {
    if (cs.empty()) return true;
    const auto L = leaf_sets(t);
    for (const auto &X : L) if (!compatible_with_all(cs, X)) return false;
    return true;
}

// comparable / differing column counts of every taxon pair (integer weights: exact, so the counts of a
// concatenation are the sums of its genes' counts -- the replicate path never touches the columns again)
void pair_counts(const EncodedAlignment &a, std::vector<int64_t> &cmp, std::vector<int64_t> &diff) {
    const int n = a.ntax, mp = a.mpad;
    cmp.assign((size_t)n * n, 0); diff.assign((size_t)n * n, 0);
    std::vector<int32_t> w(a.npat);
    for (int p = 0; p < a.npat; ++p) w[p] = (int32_t)a.weight[p];
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            // branch-free integer counts (auto-vectorised)
            int64_t icmp = 0, idiff = 0;
            const uint8_t *ci = &a.codes[(size_t)i * mp], *cj = &a.codes[(size_t)j * mp];
            for (int p = 0; p < a.npat; ++p) {
                const int32_t ok = (ci[p] < 20) & (cj[p] < 20);
                icmp += ok * w[p]; idiff += (ok & (ci[p] != cj[p])) * w[p];
            }
            cmp[(size_t)i * n + j] = cmp[(size_t)j * n + i] = icmp; diff[(size_t)i * n + j] = diff[(size_t)j * n + i] = idiff;
        }
}
Tree nj_tree(const EncodedAlignment &a, const std::vector<Constraint> *cons) {
    std::vector<int64_t> cmp, diff;
    pair_counts(a, cmp, diff);
    return nj_from_counts(a.ntax, cmp, diff, cons);
}
Tree nj_from_counts(int n, const std::vector<int64_t> &cmpc, const std::vector<int64_t> &diffc, const std::vector<Constraint> *cons) {
    const int N = 2 * n - 2;
    std::vector<double> D((size_t)N * N, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            const double cmp = (double)cmpc[(size_t)i * n + j], diff = (double)diffc[(size_t)i * n + j];
            double d = 3.0;
            if (cmp > 0) { const double pd = diff / cmp; d = -std::log(std::max(1.0 - pd - 0.2 * pd * pd, 0.05)); }
            D[(size_t)i * N + j] = D[(size_t)j * N + i] = d;
        }
    Tree T; T.ntax = n; T.nbr.assign(N, {-1, -1, -1}); T.len.assign(N, {0.0, 0.0, 0.0});
    auto connect = [&](int x, int y, double l) {
        l = std::min(std::max(l, TMIN), TMAX);
        int k = 0; while (T.nbr[x][k] >= 0) ++k; int m = 0; while (T.nbr[y][m] >= 0) ++m;
        T.nbr[x][k] = y; T.len[x][k] = l; T.nbr[y][m] = x; T.len[y][m] = l;
    };
    std::vector<int> act(n); for (int i = 0; i < n; ++i) act[i] = i;
    int next = n;
    std::vector<double> r(N, 0.0);
    const bool constrained = cons && !cons->empty();
    const int words = (n + 63) / 64;
    std::vector<std::vector<uint64_t>> cl;            // leaf set per node (constrained mode)
    if (constrained) { cl.assign(N, std::vector<uint64_t>(words, 0)); for (int i = 0; i < n; ++i) cl[i][i >> 6] |= 1ULL << (i & 63); }
    while (act.size() > 3) {
        const int m = (int)act.size();
        for (int x : act) { double s = 0; for (int y : act) s += D[(size_t)x * N + y]; r[x] = s; }
        double best = 1e300; int bi = -1, bj = -1;
        std::vector<uint64_t> X(words);
        for (int pass = 0; pass < 2 && bi < 0; ++pass)   // pass 1 (only if no compatible pair exists): unconstrained
        for (int ai = 0; ai < m; ++ai) for (int bjx = ai + 1; bjx < m; ++bjx) {
            const int x = act[ai], y = act[bjx];
            const double q = (m - 2) * D[(size_t)x * N + y] - r[x] - r[y];
            if (q < best) {
                if (constrained && pass == 0) {
                    for (int w = 0; w < words; ++w) X[w] = cl[x][w] | cl[y][w];
                    if (!compatible_with_all(*cons, X)) continue;
                }
                best = q; bi = ai; bj = bjx;
            }
        }
        const int x = act[bi], y = act[bj], u = next++;
        if (constrained) for (int w = 0; w < words; ++w) cl[u][w] = cl[x][w] | cl[y][w];
        const double dxy = D[(size_t)x * N + y];
        const double lx = 0.5 * dxy + (r[x] - r[y]) / (2.0 * (m - 2));
        connect(u, x, lx); connect(u, y, dxy - lx);
        for (int z : act) if (z != x && z != y) { const double d = 0.5 * (D[(size_t)x * N + z] + D[(size_t)y * N + z] - dxy); D[(size_t)u * N + z] = D[(size_t)z * N + u] = d; }
        act[bi] = u; act.erase(act.begin() + bj);
    }
    const int x = act[0], y = act[1], z = act[2], u = next++;
    const double dxy = D[(size_t)x * N + y], dxz = D[(size_t)x * N + z], dyz = D[(size_t)y * N + z];
    connect(u, x, 0.5 * (dxy + dxz - dyz)); connect(u, y, 0.5 * (dxy + dyz - dxz)); connect(u, z, 0.5 * (dxz + dyz - dxy));
    return T;
}

// ------------------------------------------------------------------------------------------
// Brent (state machine form)
// ------------------------------------------------------------------------------------------
void Brent::start(double lo, double hi, double x0, double fx0, double tol_) {
    a = lo; b = hi; x = w = v = x0; fx = fw = fv = fx0; d = e = 0; iter = 0; done = false; u = x0; tol = tol_;
}
bool Brent::propose() {
    const double gold = 0.3819660112501051;
    if (done || iter >= 60) { done = true; return false; }
    const double xm = 0.5 * (a + b), tol1 = tol, tol2 = 2 * tol1;      // absolute in log(alpha) = relative in alpha
    if (std::fabs(x - xm) <= tol2 - 0.5 * (b - a)) { done = true; return false; }
    bool golden = true;
    if (std::fabs(e) > tol1) {
        double r = (x - w) * (fx - fv), q = (x - v) * (fx - fw), p = (x - v) * q - (x - w) * r;
        q = 2 * (q - r); if (q > 0) p = -p; q = std::fabs(q);
        const double etemp = e; e = d;
        if (!(std::fabs(p) >= std::fabs(0.5 * q * etemp) || p <= q * (a - x) || p >= q * (b - x))) {
            d = p / q; u = x + d;
            if (u - a < tol2 || b - u < tol2) d = (xm - x >= 0) ? tol1 : -tol1;
            golden = false;
        }
    }
    if (golden) { e = (x >= xm) ? a - x : b - x; d = gold * e; }
    u = (std::fabs(d) >= tol1) ? x + d : x + (d >= 0 ? tol1 : -tol1);
    ++iter;
    return true;
}
void Brent::update(double fu) {
    if (fu <= fx) { if (u >= x) a = x; else b = x; v = w; fv = fw; w = x; fw = fx; x = u; fx = fu; }
    else {
        if (u < x) a = u; else b = u;
        if (fu <= fw || w == x) { v = w; fv = fw; w = u; fw = fu; }
        else if (fu <= fv || v == x || v == w) { v = u; fv = fu; }
    }
}

}  // namespace pml
