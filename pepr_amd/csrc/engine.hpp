// engine.hpp -- device-resident gene batches and the host-side schedulers that drive the kernels.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <string>
#include <vector>

#include "host.hpp"
#include "kernels.h"

namespace pml {

constexpr int MAXTAIL = 4;            // tail requests (evaluate / sumtable+Newton) per gene per run()

enum { K_PMAT = 0, K_NEWVIEW = 1, K_EVALUATE = 2, K_SUMTABLE = 3, K_NEWTON = 4, K_REDUCE = 5, K_HOST_BUILD = 6, K_HOST_WAIT = 7, K_COUNT = 8 };

struct Ctx {
    int device = 0;
    bool profile = false;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;      // second lane of a chained smoothing pass (half the genes each, overlapped)
    std::mutex mu;
    std::string last_error;
    Model model[2];
    bool model_ready[2] = {false, false};
    ModelDev *d_model[2] = {nullptr, nullptr};
    double *d_eigfrags[2] = {nullptr, nullptr};   // 2*PFRAG doubles each
    struct KStat { long long launches = 0; double ms = 0; double bytes = 0; double flops = 0; } stats[K_COUNT];
    long long newton_giveups = 0, newton_reissued = 0, newton_seq_launches = 0;   // k_newton fallback statistics (pml_newton_fallbacks)
    struct Ev { int kind; hipEvent_t a, b; };
    std::vector<Ev> pending;
    std::vector<hipEvent_t> pool;
    // the last destroyed batch's device arena, kept for the next batch: the driver zero-fills fresh allocations at
    // ~40 GB/s (a 111 GiB C4 arena costs 3-6 s), so back-to-back batches (full tree then replicates, successive
    // one-shot calls) reuse it; released by destroy() or when a larger one replaces it
    char *arena_cache = nullptr; size_t arena_cache_bytes = 0;

    int init(int dev, bool prof);
    // a worker context of a grouped search call (api.cpp): its own stream (highest priority class: its own pool of hardware
    // queues), events, statistics and arena cache
    int init_worker(const Ctx &parent);
    bool owns_stream = true;
    hipEvent_t ev_sync = nullptr;
    int sync(hipStream_t s);             // wait for everything THIS context has enqueued on s so far (event-based: other users of a shared stream may enqueue behind it)
    void destroy();
    int ensure_model(int pi_mode);
    hipEvent_t get_event();
    hipStream_t tic_stream = nullptr;    // stream the next tic/toc pair is recorded on (null = stream)
    void tic(int kind, double bytes, double flops = 0);   // record start (profile mode); bytes / flops = SURVEY 8d per-operation figures
    void toc();                          // record stop
    void resolve_events();               // after a stream sync
    int fail(int code, const std::string &msg) { last_error = msg; return code; }
};

struct Gene {
    EncodedAlignment aln;
    Tree tree;
    double alpha = 1.0;
    double rates[NCAT] = {1, 1, 1, 1};
    unsigned rates_epoch = 0;      // bumped by set_alpha (cached score plans refresh a gene's rates only when it moved)
    // device pointers (inside the batch arena)
    uint8_t *d_codes = nullptr;
    double *d_weight = nullptr;
    double *d_clv = nullptr;       // slot_cap slots of 80*mpad doubles
    int *d_scl = nullptr;          // slot_cap slots of mpad ints
    double *d_sumtab[MAXTAIL] = {};    // 80*mpad each
    int *d_sumscl[MAXTAIL] = {};       // mpad each
    double *d_patlnl[MAXTAIL] = {};    // mpad each
    int slot_cap = 0, next_slot = 0;
    std::vector<int> slot_of;      // directed-edge index (v-ntax)*3+k -> slot (-1 = none)
    std::vector<uint8_t> valid;
    std::vector<int> pend_level;   // scratch for collection (-1 = not pending)
    std::vector<Constraint> cons;  // topological constraints of the running search (empty = none)
    std::vector<uint8_t> dirty;    // [node*3+slot]: branch needs re-optimisation (both directions set)
    double *d_len = nullptr;       // [node*3+slot] device copies of branch lengths written by k_newton (chained passes)
    std::vector<uint8_t> len_pending;   // [node*3+slot]: the current length lives in d_len, the host value is stale
    void mark_node(int v) { for (int k = 0; k < 3; ++k) { const int w = tree.nbr[v][k]; if (w < 0) continue; dirty[v * 3 + k] = 1; dirty[w * 3 + tree.slot(w, v)] = 1; } }
    void mark_all() { dirty.assign((size_t)tree.nnodes() * 3, 1); }
    void mark_none() { dirty.assign((size_t)tree.nnodes() * 3, 0); }
};

void det_record(int batch, const Gene &G, char kind, int a, int b, double x, double y, double z);   // PML_DET_LOG diagnostic (engine.cpp)

struct pml_alignment_view { int ntax, nsites; const char *const *names; const char *const *rows; };

// gene alignments encoded ONCE and kept in HBM; replicates (gene subsets) are gathered from them on the device
struct GeneStore {
    struct Item { EncodedAlignment aln; std::vector<int64_t> cmp, diff; uint8_t *d_codes = nullptr; double *d_w = nullptr; };
    Ctx *ctx = nullptr; std::vector<Item> items; char *arena = nullptr;
    int create(Ctx *c, int n, const pml_alignment_view *alns);
    void destroy();
};

constexpr int NNI_PARTS = 6;         // parts an NNI round deals a gene's edges over (four scratch CLVs each)
constexpr int NSCRATCH = 4 * NNI_PARTS > 8 ? 4 * NNI_PARTS : 8;   // extra CLV slots per gene for candidate evaluation (NNI: 4 per part; SPR: 8)
enum { SIDE_TIP = 0, SIDE_MSG = 1, SIDE_SCRATCH = 2, SIDE_CHERRY = 3, SIDE_PITCH = 4 };
// tip node id | directed-edge index (v-ntax)*3+k | scratch slot | directed-edge index of a message whose
// two children are tips ("cherry": never materialised, recomputed from two tip tables where consumed)
struct Side { int kind, id; };
// bv/bq (optional, scratch outputs): the tree branch (node, slot) child c's length belongs to -- lets run() share ONE
// transition-matrix request among all operations of a launch that cross the same branch
struct PendingOp { int gene, out_kind, out_id, level; Side child[2]; double t[2]; int bv[2] = {-1, -1}, bq[2] = {0, 0};
                   bool unstored = false; /* run(): the result stayed in registers (OPF_NO_STORE) and is not valid in memory */
                   bool transient = false; /* caller: only the operation or tail that directly follows reads the result (scratch slots) */
                   int part = 0; /* run(): operations and tails of one gene with different parts are independent of each other and become separate runs
                                    of the launch (their workgroups run side by side): NNI rounds deal the edges of a gene over parts */ };

struct Batch {
    Ctx *ctx = nullptr;
    int pi_mode = 0, ncat = 4, det_id = 0;
    // pi_mode 2 (PROTGAMMAWAGF): every gene has its own eigen-system from its empirical frequencies
    ModelDev *d_gmodel = nullptr; double *d_geig = nullptr;       // [genes] models, [genes][2*PFRAG] eigen-basis fragment sets
    const ModelDev *model_of(int g) const { return d_gmodel ? d_gmodel + g : ctx->d_model[pi_mode]; }
    const double *eig_of(int g) const { return d_geig ? d_geig + (size_t)g * 2 * PFRAG : ctx->d_eigfrags[pi_mode]; }
    int build_gene_models();
    int share = 1;                 // batches working on the device at the same time (groups of one search call): free HBM is divided by it
    double newton_tol = 1e-8;      // Newton stop |dt| < tol: 1e-8 fine, 1e-6 in coarse phases
    std::vector<Gene> genes;
    char *arena = nullptr; size_t arena_bytes = 0;
    // staging (pinned host mirrors + device buffers), grown on demand
    void *h_stage = nullptr; size_t h_cap = 0;
    void *d_stage = nullptr; size_t d_cap = 0;
    double *d_frags = nullptr; size_t frag_cap = 0;      // in fragment sets
    double *d_frags2 = nullptr; size_t frag_cap2 = 0;    // lane 1
    double *d_nsync2 = nullptr; size_t nsync_cap2 = 0;
    int lane = 0;                                        // which stream / buffers the next run() uses (chained passes)
    hipEvent_t ev_stagger = nullptr; bool record_stagger = false;   // lane 1 starts one k_oplist behind lane 0
    double *d_scalars = nullptr; double *h_scalars = nullptr;   // 8 doubles per gene and tail slot: device buffer + pinned host mirror
    size_t scalars_doubles = 0, results_used = 0;
    int fetch_results(bool pooled);                              // enqueue the device -> host copies of the result buffers
    double *d_nsync = nullptr; size_t nsync_cap = 0;             // Newton inter-workgroup sync blocks
    NewtonCtl *d_nctl = nullptr;                                  // [2]: k_newton control block per lane (tickets, abort word)
    // k_newton's exchange gave up (a co-tenant kept slices of a request apart for longer than the wall-clock bound): the
    // affected work is re-issued through the no-exchange SEQ form, and the next `safe_left` Newton launch sets use it from the
    // start (doubling hold-off, so a GPU that stays shared costs one time-out per hold-off period, not one per launch)
    int safe_left = 0, safe_hold = 4; bool safe_now = false;
    unsigned newton_launch_seq = 0;      // launches with Newton tails so far: the exchange granules' tag base (never cleared, tags are unique)
    bool in_retry = false;
    bool newton_safe_mode() { return safe_now || safe_left > 0; }
    void newton_gave_up();
    int clear_abort();
    // cached descriptors of the full-traversal score of ALL genes (topology unchanged): replays skip
    // the tree walk and the descriptor build; transition matrices, CLVs and lnL are recomputed
    struct ReqSrc { int gene, v, q, fold; };      // branch (v, slot q) whose length a P request uses
    struct Plan {
        bool valid = false; unsigned epoch = 0;
        void *h = nullptr, *d = nullptr; size_t bytes = 0;
        size_t o_req = 0, o_ops = 0, o_runs = 0, o_red = 0, nreq = 0, nruns = 0, neval = 0;
        int max_mpad = 0; double algo_bytes = 0, algo_flops = 0; bool any_pitch = false, any_chain = false;
        bool stored = false;                       // recorded with every CLV written (the traversal a search runs) instead of OPF_NO_STORE
        std::vector<ReqSrc> src; std::vector<std::pair<int, int>> outs;
        std::vector<unsigned> rates_seen;          // per gene: rates_epoch the descriptors carry
    } plan;
    long cnt_smooth = 0, cnt_nni = 0, cnt_spr = 0, cnt_eval = 0, cnt_passes = 0;   // run() calls by purpose (PML_TRACE)
    // host wall time by phase, ms (PML_TRACE prints them at the end of a search): where the device waits for the host
    enum { HP_PASS_SETUP, HP_PASS_STEPS, HP_PASS_SYNC, HP_PASS_POST, HP_NNI_BUILD, HP_NNI_RUN, HP_NNI_SELECT, HP_ALPHA_HOST, HP_RUN_SYNCED, HP_N };
    double host_phase_ms[HP_N] = {0};
    unsigned topo_epoch = 0;       // bumped whenever a search may change a topology
    bool score_only_batch = false;
    int replay_plan(double *lnl);
    bool record_plan = false, record_stored = false;
    std::vector<ReqSrc> last_src;
    std::vector<size_t> req_off; std::vector<uint32_t> req_stamp; std::vector<const double *> req_ptr; uint32_t req_launch = 0;   // run(): keyed request table
    std::vector<std::vector<int>> run_tails_of; std::vector<int> run_nparts; std::vector<size_t> run_koff;
    std::vector<std::vector<std::pair<int, int>>> pass_order; std::vector<std::vector<uint8_t>> pass_next;      // smooth_pass scratch
    std::vector<std::vector<std::pair<uint64_t, const double *>>> val_bucket; std::vector<uint32_t> val_stamp;   // run(): requests shared by value
    // chained mode: run() enqueues its copy + kernels and returns WITHOUT synchronising; descriptors are bump-
    // allocated in the staging buffer; branch lengths optimised earlier in the chain are read from Gene::d_len
    bool chain = false; size_t chain_off = 0;
    double *d_lenpool = nullptr;
    double *d_chain = nullptr, *h_chain = nullptr; size_t chain_cap = 0;     // 4 doubles per chained Newton result
    // a step of a chained pass whose upload + launches are issued later, grouped with its neighbours (flush_deferred)
    struct Deferred { size_t base, bytes, o_req, o_ops, o_runs, o_red, o_newt, o_tick, nreq, nruns, neval, nnewton; int nt_reg, nt_stream; bool seq, fused;
                      int max_mpad, newton_maxm, lane; bool any_pitch, any_chain, stagger; double algo_bytes, newton_bytes, algo_flops; };
    std::vector<Deferred> deferred; size_t flush_quota = 1; bool lanes_active = false;
    int flush_deferred();
    int chain_begin(size_t nresults);
    int ensure_results(size_t nresults);   // d_chain (device) / h_chain (pinned mirror): 4 doubles per pooled result
    // pooled sumtables for launches that carry more Newton requests per gene than MAXTAIL (all edges of an NNI round)
    char *d_tailpool = nullptr; size_t tailpool_cap = 0;
    int ensure_tailpool(size_t bytes);
    int chain_sync();                  // wait for everything enqueued; the staging buffer is free again

    int create(Ctx *c, int n, const pml_alignment_view *alns, const char *const *newicks,
               int pi_mode, int ncat, double alpha, bool score_only);
    // one batch gene per replicate = the concatenation of the selected store genes (sorted taxon union, absent
    // taxa = gap rows, MSAConcatenator rules); code matrices are gathered on the device, NJ start trees come from
    // the summed pair counts: no column text is touched again
    int create_replicates(Ctx *c, const GeneStore &store, const std::vector<std::vector<int>> &sel, int pi_mode, int ncat, double alpha);
    int layout(double alpha, bool score_only);
    void destroy();

    void set_alpha(int g, double alpha);
    void invalidate_all(int g);
    void branch_changed(int g, int a, int b);
    // lnL of every active gene at the branch above taxon 0, using/refreshing cached CLVs
    int evaluate(const std::vector<char> &active, double *lnl);
    // invalidate + evaluate; stored: the whole-tree pass writes every CLV (chained children are still READ from registers),
    // i.e. the traversal a search runs after a topology or alpha change -- bench.py's second timed leg
    int score(const std::vector<char> &active, double *lnl, bool stored = false);
    int site_lnl(int g, double *out);
    int root_derivs(double *lnl, double *d1, double *d2);
    // one pass over the DIRTY branches (DFS order); a branch that moves by more than thr flags itself and
    // its neighbours for the next pass
    int smooth_pass(const std::vector<char> &active, std::vector<double> &maxdelta, double thr);
    int opt_alpha(const std::vector<char> &active, double *lnl, double tol = 1e-4);
    int optimize(bool opt_alpha_flag, double eps, double *lnl, const std::vector<char> *mask = nullptr);
    int light_smooth(const std::vector<char> &active, double *lnl);
    int nni_round(const std::vector<char> &active, std::vector<double> &lnl, std::vector<int> &applied);
    int spr_round(const std::vector<char> &active, int radius, std::vector<double> &lnl, std::vector<int> &moves);
    int search(bool nni, int spr_radius, bool opt_alpha_flag, double eps, double *lnl);
    // SH-like local supports (FastTree's default output; FastTreeRunner.java:67-70 without -nosupport): per gene one value
    // per internal edge in nni_round's edge order (u ascending, slot ascending, v > u inner), plus the (u, v) pairs
    // FastTree -gamma: per-pattern likelihoods at the 20 fixed rates (five traversals of four rates), then alpha and the
    // length rescale fitted on that table (k_g20 evaluates, the host steers two alternating Brent searches per gene)
    int gamma20(std::vector<double> &lnl20, std::vector<double> &alpha20, std::vector<double> &rescale20);
    int sh_support(int nboot, unsigned long long seed, std::vector<std::vector<double>> &support, std::vector<std::vector<std::pair<int, int>>> &edges_out);
    int *d_site2pat = nullptr; std::vector<size_t> site2pat_off;
    // FastTree -constraints matrix (names, rows of '0' '1' '-'); start trees that violate it are rebuilt
    int set_constraints(int ncons, int ntax, const char *const *names, const char *const *rows);

    // --- plumbing ---
    int need(int g, int v, int to, std::vector<PendingOp> &ops);   // returns level
    // run the collected newviews, then the tail ops (evaluate or sumtable+newton), one sync
    // a tail is evaluated after `after` newview ops of its gene (-1 = after all of them); `slot`
    // selects one of the gene's MAXTAIL sumtable / per-pattern-lnL / result buffers
    struct Tail { int gene; Side a, b; int mode; double t0; int max_iter; int slot = 0; int after = -1;
                  int bv = -1, bq = 0; /* tree branch (node, slot) t0 comes from: plan replay */
                  double *result_dev = nullptr;               /* chained pass: where k_newton writes (else res(g, slot)) */
                  double *t_dev0 = nullptr, *t_dev1 = nullptr; /* chained pass: d_len entries of the branch */
                  double *patlnl_dev = nullptr;               /* Newton tails: per-pattern lnL at the optimised length */
                  double *sumtab_dev = nullptr;               /* Newton tails: pooled sumtable (80*mpad doubles + mpad ints) instead of the gene's slot buffer */
                  int *scl_dev = nullptr;                     /* MODE_EVALUATE_CAT: where the scaling counts go (patlnl_dev = the 4 x mpad table slice) */
                  const double *result_host = nullptr;        /* host view of result_dev when it is mapped memory (checked after the sync) */
                  int part = 0;                               /* as PendingOp::part */ };
    double *res(int g, int slot = 0) const { return h_scalars + 8 * ((size_t)g * MAXTAIL + slot); }
    Side msg(int g, int node, int toward) const;
    bool is_cherry(int g, int node, int toward) const;
    // 0: real message, 1: cherry (two tips), 2: pitchfork (a cherry and a tip): both are virtual
    int virt_kind(int g, int node, int toward) const;
    bool virtual_pitch = true;         // PML_NO_PITCH=1
    bool virtual_cherries = true;      // PML_NO_CHERRY=1 materialises cherry CLVs like any other (A/B switch)
    int run(std::vector<PendingOp> &ops, const std::vector<Tail> &tails);
    int ensure_stage(size_t bytes);
    int ensure_frags(size_t sets);
    int slot_for(Gene &g, int idx);
};

// parsimony.hip: Fitch parsimony trees (stepwise addition + SPR) for n genes, one launch per device step
int parsimony_batch(Ctx *ctx, int n, const pml_alignment_view *alns, unsigned seed, int radius, std::vector<Tree> &trees_out,
                    std::vector<EncodedAlignment> &alns_out, std::vector<long long> &lengths_out, std::vector<int> &moves_out);

}  // namespace pml
