// shim_main.cpp -- CLI-compatible stand-ins for the executables PEPR spawns, so that a stock
// pepr.jar runs unmodified when the tool-path system properties point here
// (.../util/ExecUtilities.java:168-190; .../pipeline/PhyloPipeline.java:846-870).
// Built twice: -DSHIM_FASTTREE -> bin/FastTree_WAG, -DSHIM_RAXML -> bin/raxmlHPC (+ -PTHREADS).
// Accepts exactly the argv subsets PEPR emits:
//   FastTree_WAG -gamma [-nosupport] [-constraints <c.faa>] <aln.faa>   FastTreeRunner.java:67-86
//     (without -nosupport the tree carries SH-like local supports, 0-1, 3 decimals; -seed / -boot as FastTree)
//   raxmlHPC -f d|e|g -m PROTGAMMAWAG -s <aln.phy> -n <run> [-t tree] [-z trees] [-T n] [-p seed]
//                                                                   RAxMLRunner.java:115-132,196-208,253-272
// Everything numeric happens in libpeprml.so (HIP); these files only parse and print.
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/peprml.h"

struct Aln { std::vector<std::string> names, rows; };

static bool read_fasta(const char *path, Aln &a, std::string &err) {
    std::ifstream f(path);
    if (!f) { err = std::string("cannot open ") + path; return false; }
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '>') {
            size_t e = 1; while (e < line.size() && !std::isspace((unsigned char)line[e])) ++e;
            a.names.push_back(line.substr(1, e - 1)); a.rows.emplace_back();
        } else if (!a.rows.empty()) {
            for (char c : line) if (!std::isspace((unsigned char)c)) a.rows.back().push_back(c);
        }
    }
    if (a.names.size() < 3) { err = "need at least 3 sequences"; return false; }
    for (auto &r : a.rows) if (r.size() != a.rows[0].size()) { err = "sequences have different lengths"; return false; }
    return true;
}
// relaxed PHYLIP as SequenceAlignment.getAlignmentAsExtendedPhylipUsingTaxonNames writes it
// (.../alignment/SequenceAlignment.java:489-522): "n L" then one "name<spaces>sequence" line per taxon
static bool read_phylip(const char *path, Aln &a, std::string &err) {
    std::ifstream f(path);
    if (!f) { err = std::string("cannot open ") + path; return false; }
    size_t n = 0, L = 0;
    if (!(f >> n >> L) || n < 3) { err = "bad PHYLIP header"; return false; }
    for (size_t i = 0; i < n; ++i) {
        std::string name, seq, tok;
        if (!(f >> name)) { err = "truncated PHYLIP file"; return false; }
        while (seq.size() < L && (f >> tok)) seq += tok;
        if (seq.size() != L) { err = "sequence length mismatch for " + name; return false; }
        a.names.push_back(name); a.rows.push_back(seq);
    }
    return true;
}
static std::string read_file(const char *path) { std::ifstream f(path); std::stringstream ss; ss << f.rdbuf(); return ss.str(); }
static pml_alignment view(const Aln &a, std::vector<const char *> &np, std::vector<const char *> &rp) {
    for (auto &s : a.names) np.push_back(s.c_str());
    for (auto &s : a.rows) rp.push_back(s.c_str());
    pml_alignment v; v.ntax = (int)a.names.size(); v.nsites = (int)a.rows[0].size(); v.names = np.data(); v.rows = rp.data();
    return v;
}
// re-print branch lengths with `digits` decimals (library returns 20)
static std::string reformat(const char *nw, int digits, bool raxml_tail, double scale = 1.0) {
    std::string out; const char *p = nw; char buf[64];
    while (*p) {
        if (*p == ':') { char *e; double v = std::strtod(p + 1, &e) * scale; std::snprintf(buf, sizeof buf, ":%.*f", digits, v); out += buf; p = e; }
        else out += *p++;
    }
    if (raxml_tail && out.size() > 1 && out.back() == ';') { out.pop_back(); out += ":0.0;"; }
    return out;
}
static int fail(const char *tool, const std::string &msg) { std::fprintf(stderr, "%s: %s\n", tool, msg.c_str()); return 1; }

#ifdef SHIM_FASTTREE
int main(int argc, char **argv) {
    const char *tool = "FastTree_WAG";
    const char *file = nullptr, *cons_file = nullptr; bool gamma = false, nosupport = false; unsigned long long sh_seed = 314159; int nboot = 1000;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-gamma") gamma = true;
        else if (a == "-nosupport") nosupport = true;
        else if (a == "-quiet" || a == "-nopr") {}
        else if (a == "-seed" && i + 1 < argc) sh_seed = std::strtoull(argv[++i], nullptr, 10);
        else if (a == "-boot" && i + 1 < argc) nboot = std::atoi(argv[++i]);
        else if (a == "-gtr" || a == "-nt") return fail(tool, "nucleotide models are not built (PEPR never requests them)");
        else if (a == "-constraints" && i + 1 < argc) cons_file = argv[++i];
        else if (a == "-log" && i + 1 < argc) ++i;
        else if (a[0] == '-') return fail(tool, "unknown option " + a);
        else file = argv[i];
    }
    if (!file) return fail(tool, "usage: FastTree_WAG -gamma -nosupport alignment.faa > tree");
    Aln a; std::string err;
    if (!read_fasta(file, a, err)) return fail(tool, err);
    pml_ctx *ctx = nullptr; pml_config cfg = {0, 0, 0};
    if (int rc = pml_create(&cfg, &ctx)) return fail(tool, std::string("engine: ") + pml_strerror(rc) + " " + pml_last_error(nullptr));
    std::vector<const char *> np, rp; pml_alignment v = view(a, np, rp);
    pml_model model = {4, 1.0, PML_PI_WAG_FULL};           // FastTree_WAG holds the 7-digit WAG frequencies
    pml_search_opts opts = {1, 1, 0, 1e-3, 0};         // NJ start + NNI rounds, as FastTree's ML stage
    Aln cons; std::vector<const char *> cnp, crp;
    if (cons_file) {                                   // FastTreeRunner.java:54-64: FASTA of 0/1/- rows
        if (!read_fasta(cons_file, cons, err)) return fail(tool, "constraints: " + err);
        for (auto &x : cons.names) cnp.push_back(x.c_str());
        for (auto &x : cons.rows) crp.push_back(x.c_str());
        opts.nconstraints = (int)cons.rows[0].size(); opts.constraint_ntax = (int)cons.names.size();
        opts.constraint_names = cnp.data(); opts.constraint_rows = crp.data();
    }
    pml_result res;
    const int rc = pml_search(ctx, &v, nullptr, &model, &opts, &res);
    if (rc) { std::string m = pml_last_error(ctx); pml_destroy(ctx); return fail(tool, m); }
    std::fprintf(stderr, "FastTree_WAG (peprml, MI355X): %d seqs, %d positions, %d patterns\nGamma(4) LogLk = %.3f alpha = %.3f\n",
                 v.ntax, v.nsites, res.npatterns, res.lnl, res.alpha);
    // -gamma (PEPR always passes it, FastTreeRunner.java:67-70): the likelihood under 20 fixed rates with alpha and a length
    // rescale fitted on the per-site x rate table; the printed tree carries the rescaled lengths
    double rescale = 1.0;
    if (gamma) {
        pml_result g20;
        const int rc3 = pml_gamma20(ctx, &v, res.newick, &model, &g20, &rescale);
        if (rc3) { std::string m = pml_last_error(ctx); pml_result_free(&res); pml_destroy(ctx); return fail(tool, m); }
        std::fprintf(stderr, "Gamma(20) LogLk = %.3f alpha = %.3f rescaling lengths by %.3f\n", g20.lnl, g20.alpha, rescale);
        pml_result_free(&g20);
    }
    if (!nosupport && nboot > 0 && v.ntax > 3) {         // FastTree's default: SH-like local supports (0-1) as inner labels
        pml_model m2 = {4, res.alpha, PML_PI_WAG_FULL};
        pml_result sup;
        const int rc2 = pml_sh_support(ctx, &v, res.newick, &m2, nboot, sh_seed, &sup);
        if (rc2) { std::string m = pml_last_error(ctx); pml_result_free(&res); pml_destroy(ctx); return fail(tool, m); }
        std::printf("%s\n", reformat(sup.newick, 5, false, rescale).c_str());
        pml_result_free(&sup);
    } else std::printf("%s\n", reformat(res.newick, 5, false, rescale).c_str());
    pml_result_free(&res); pml_destroy(ctx);
    return 0;
}
#endif

#ifdef SHIM_RAXML
int main(int argc, char **argv) {
    const char *tool = "raxmlHPC";
    std::string f = "d", model_s = "PROTGAMMAWAG", aln_f, run, tree_f, trees_f;
    bool pars_only = false; unsigned seed = 12345; unsigned long long bs_seed = 12345; int bs_reps = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto val = [&](std::string &dst) { if (i + 1 >= argc) return false; dst = argv[++i]; return true; };
        std::string dummy;
        if (a == "-f") { if (!val(f)) return fail(tool, "-f needs a value"); }
        else if (a == "-m") { if (!val(model_s)) return fail(tool, "-m needs a value"); }
        else if (a == "-s") { if (!val(aln_f)) return fail(tool, "-s needs a value"); }
        else if (a == "-n") { if (!val(run)) return fail(tool, "-n needs a value"); }
        else if (a == "-t") { if (!val(tree_f)) return fail(tool, "-t needs a value"); }
        else if (a == "-z") { if (!val(trees_f)) return fail(tool, "-z needs a value"); }
        else if (a == "-T") { if (!val(dummy)) return fail(tool, a + " needs a value"); }
        else if (a == "-p") { if (!val(dummy)) return fail(tool, a + " needs a value"); seed = (unsigned)std::strtoul(dummy.c_str(), nullptr, 10); }
        else if (a == "-y") pars_only = true;              // RAxMLRunner.java:134-140,241-251: parsimony start tree only
        else if (a == "-Y") return fail(tool, "parsimony bootstrap (-Y -N) is not built");
        else if (a == "-x") { if (!val(dummy)) return fail(tool, "-x needs a value"); bs_seed = std::strtoull(dummy.c_str(), nullptr, 10); }
        else if (a == "-N" || a == "-#") { if (!val(dummy)) return fail(tool, a + " needs a value"); bs_reps = std::atoi(dummy.c_str()); }
        else return fail(tool, "unknown option " + a);
    }
    if (aln_f.empty() || run.empty()) return fail(tool, "usage: raxmlHPC -f d|e|g -m PROTGAMMAWAG -s aln.phy -n run [-t tree] [-z trees]");
    // two models are built: PROTGAMMAWAG and PROTGAMMAWAGF (the same exchangeabilities with frequencies counted from the
    // alignment); PROTCATWAG, PROTGAMMAIWAG and the other matrices -matrix_eval may pass (PhylogenomicPipeline2.java:260-284:
    // their tables are not in the reference) are different likelihood functions and are refused rather than run as WAG
    // under their name
    if (model_s != "PROTGAMMAWAG" && model_s != "PROTGAMMAWAGF") return fail(tool, "only -m PROTGAMMAWAG and PROTGAMMAWAGF are built, got " + model_s);
    const int pi_mode_s = model_s == "PROTGAMMAWAGF" ? PML_PI_EMPIRICAL : PML_PI_RAXML_3DP;
    if (std::ifstream("RAxML_info." + run)) return fail(tool, "RAxML output files with the run ID <" + run + "> already exist");
    if (f == "b") {                                       // RAxMLRunner.java:453-516: draw the bipartition frequencies of -z trees on -t tree (host only)
        if (tree_f.empty() || trees_f.empty()) return fail(tool, "-f b needs -t tree -z trees");
        const std::string main_tree = read_file(tree_f.c_str());
        std::ifstream tf(trees_f); std::string line; std::vector<std::string> trees;
        while (std::getline(tf, line)) { bool blank = true; for (char c : line) if (!std::isspace((unsigned char)c)) blank = false; if (!blank) trees.push_back(line); }
        std::vector<const char *> tp; for (auto &t : trees) tp.push_back(t.c_str());
        char *out = nullptr;
        if (int rc = pml_support_tree(main_tree.c_str(), (int)tp.size(), tp.data(), 20, &out)) return fail(tool, std::string("-f b: ") + pml_strerror(rc) + " " + pml_last_error(nullptr));
        // RAxML prints percentages; pml_support_tree counts trees
        std::string res; const int nt = (int)tp.size();
        for (const char *p = out; *p; ++p) {
            res += *p;
            if (*p == ')' && std::isdigit((unsigned char)p[1])) { char *e; const long c = std::strtol(p + 1, &e, 10); res += std::to_string(nt ? (int)(0.5 + 100.0 * c / nt) : 0); p = e - 1; }
        }
        pml_free(out);
        std::ofstream("RAxML_bipartitions." + run) << reformat(res.c_str(), 20, true) << "\n";
        std::ofstream("RAxML_info." + run) << "peprml raxmlHPC shim: -f b, " << nt << " trees drawn on " << tree_f << "\n";
        return 0;
    }
    Aln a; std::string err;
    if (!read_phylip(aln_f.c_str(), a, err)) return fail(tool, err);
    pml_ctx *ctx = nullptr; pml_config cfg = {0, 0, 0};
    if (int rc = pml_create(&cfg, &ctx)) return fail(tool, std::string("engine: ") + pml_strerror(rc) + " " + pml_last_error(nullptr));
    std::vector<const char *> np, rp; pml_alignment v = view(a, np, rp);
    pml_model model = {4, 1.0, pi_mode_s};
    std::ofstream info("RAxML_info." + run), logf("RAxML_log." + run);
    info << "peprml raxmlHPC shim (MI355X HIP engine), model " << model_s << ", alignment " << aln_f << "\n";
    int rc = 0;
    if (f == "a") {                                       // RAxMLRunner.java:115-132 with bootstrapReps > 0
        if (bs_reps <= 0) { pml_destroy(ctx); return fail(tool, "-f a needs -x seed -N reps"); }
        pml_result res; char *reps_txt = nullptr;
        rc = pml_bootstrap(ctx, &v, &model, bs_reps, bs_seed, 5, 1e-3, &res, &reps_txt);
        if (!rc) {
            std::ofstream("RAxML_bipartitions." + run) << reformat(res.newick, 20, true) << "\n";   // read at RAxMLRunner.java:302-318
            std::ofstream("RAxML_bootstrap." + run) << reps_txt;
            char b[256]; std::snprintf(b, sizeof b, "Final ML Optimization Likelihood: %.6f\nalpha: %.6f\n", res.lnl, res.alpha);
            info << b;
            // best tree without the support labels
            std::string plain; for (const char *p = res.newick; *p; ++p) { plain += *p; if (*p == ')') { while (std::isdigit((unsigned char)p[1])) ++p; } }
            std::ofstream("RAxML_bestTree." + run) << reformat(plain.c_str(), 20, true) << "\n";
            pml_free(reps_txt); pml_result_free(&res);
        }
    } else if (f == "d" && pars_only) {
        pml_parsimony_opts po = {seed, 20};
        pml_result res; long long mp = 0;
        rc = pml_parsimony(ctx, &v, &po, &res, &mp);
        if (!rc) {
            std::ofstream("RAxML_parsimonyTree." + run) << res.newick << "\n";     // topology only, read at RAxMLRunner.java:338-359
            info << "Parsimony tree length: " << mp << "\n";
            pml_result_free(&res);
        }
    } else if (f == "d") {
        pml_search_opts opts = {1, 1, 5, 1e-3, seed};       // parsimony start (-p seed), NNI + lazy SPR radius 5 ("best rearrangement setting 5")
        pml_result res;
        rc = pml_search(ctx, &v, nullptr, &model, &opts, &res);
        if (!rc) {
            const std::string nw = reformat(res.newick, 20, true);
            std::ofstream("RAxML_result." + run) << nw << "\n"; std::ofstream("RAxML_bestTree." + run) << nw << "\n";
            char b[256]; std::snprintf(b, sizeof b, "Final GAMMA-based Score of best tree %.6f\nalpha: %.6f\nTree-Length: %.6f\n", res.lnl, res.alpha, res.tree_length);
            info << b; logf << "0.0 " << res.lnl << "\n";
            pml_result_free(&res);
        }
    } else if (f == "e") {
        if (tree_f.empty()) { pml_destroy(ctx); return fail(tool, "-f e needs -t tree"); }
        const std::string tr = read_file(tree_f.c_str());
        pml_search_opts opts = {1, 0, 0, 1e-4, 0};
        pml_result res;
        rc = pml_optimize(ctx, &v, tr.c_str(), &model, &opts, &res);
        if (!rc) {
            std::ofstream("RAxML_result." + run) << reformat(res.newick, 20, true) << "\n";
            char b[256]; std::snprintf(b, sizeof b, "Final GAMMA  likelihood: %.6f\nalpha: %.6f\nTree-Length: %.6f\n", res.lnl, res.alpha, res.tree_length);
            info << b; pml_result_free(&res);
        }
    } else if (f == "g") {
        if (trees_f.empty()) { pml_destroy(ctx); return fail(tool, "-f g needs -z trees"); }
        std::ifstream tf(trees_f); std::string line; std::vector<std::string> trees;
        while (std::getline(tf, line)) { bool blank = true; for (char c : line) if (!std::isspace((unsigned char)c)) blank = false; if (!blank) trees.push_back(line); }
        std::ofstream out("RAxML_perSiteLLs." + run);
        out << "  " << trees.size() << "  " << v.nsites << "\n";
        for (size_t i = 0; i < trees.size() && !rc; ++i) {
            pml_search_opts opts = {1, 0, 0, 1e-4, 0};
            pml_result o, r;
            rc = pml_optimize(ctx, &v, trees[i].c_str(), &model, &opts, &o);     // -f g optimises model + lengths per tree
            if (rc) break;
            pml_model m2 = {4, o.alpha, pi_mode_s};
            rc = pml_score(ctx, &v, o.newick, &m2, PML_WANT_SITE_LNL, &r);
            if (!rc) {
                out << "tr" << (i + 1) << "\t";
                char b[64];
                for (int s = 0; s < v.nsites; ++s) { std::snprintf(b, sizeof b, "%.6f ", r.site_lnl[s]); out << b; }
                out << "\n";
                pml_result_free(&r);
            }
            pml_result_free(&o);
        }
    } else { pml_destroy(ctx); return fail(tool, "-f " + f + " is not built (a, b, d, e, g are)"); }
    if (rc) { std::string m = pml_last_error(ctx); pml_destroy(ctx); return fail(tool, m); }
    pml_destroy(ctx);
    return 0;
}
#endif
