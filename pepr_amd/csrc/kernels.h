// kernels.h -- device kernel interface of libpeprml (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace pml {

constexpr int NS = 20;                 // amino-acid states
constexpr int NCAT = 4;                // Gamma categories laid out per CLV (ncat=1 replicates)
constexpr int CLV_ROWS = NCAT * NS;    // 80 rows of `mpad` doubles: CLV[cat*20+state][pattern]
constexpr int PFRAG = NCAT * 25 * 16;      // doubles per transition-matrix fragment set (12.8 KB)
constexpr int PAT_PER_WAVE = 32;       // one MFMA chunk: 2 N-tiles of 16 patterns (16 B / lane)
// CLVs and sumtables are TILED in HBM: tile k holds patterns [128k, 128k+128) as 80 rows x 128 doubles = 80 KB contiguous,
// element (row, p) at ((p >> 7) * 80 + row) * 128 + (p & 127).  One k_oplist workgroup / one k_newton slice owns exactly one
// tile, so every CLV it reads or writes is ONE contiguous 80 KB stream (round 1: 80 separate 1 KB pieces 8 KB apart;
// profiles/r02_ubench_hbm.txt: contiguous tiles stream 5.9 TB/s for this 2R:1W mix, scattered 1 KB pieces 4.7-5.2).
constexpr int TILE_PAT = 128;
constexpr size_t clv_doubles(int mpad) { return (size_t)((mpad + TILE_PAT - 1) / TILE_PAT) * TILE_PAT * (NCAT * NS); }
constexpr size_t clv_index(int row, int p) { return ((size_t)(p >> 7) * (NCAT * NS) + (size_t)row) * TILE_PAT + (size_t)(p & (TILE_PAT - 1)); }
constexpr int NCODES = 23;
// tip table: T[c][code][q][kk] (kk padded to 6) = sum_{j in code} P_c[s = 4 kk + q][j]: the five rows a lane
// needs (its q, kk = 0..4) are 40 contiguous bytes of a 48-byte, 16-byte-aligned record -> 3 loads instead of 5
// gathers (17.7 KB per branch)
constexpr int TIPTAB_KK = 6;
constexpr int TIPTAB_DOUBLES = NCAT * NCODES * 4 * TIPTAB_KK;
constexpr int FRAG_STRIDE = TIPTAB_DOUBLES;          // doubles per k_pmat output slot (fragment set or tip table)

// device-resident model constants (one per ctx)
struct ModelDev {
    double eval[NS];
    double U[NS * NS];      // row-major U[s][k]
    double Uinv[NS * NS];   // Uinv[k][j]
    double pi[NS];
    double UinvT[NS * NS];  // UinvT[j][k] = Uinv[k][j]: a column of Uinv contiguous (k_pmat reads it with scalar loads)
};

// request for one transition-matrix fragment set: P(t * rate_c), c = 0..3
struct PmatReq {
    double t;
    double rates[NCAT];
    int kind;               // PM_FRAGS, PM_FRAGS_PI (rows scaled by pi_s: root evaluation), PM_TIPTABLE
    int pad;
    const double *tp;       // non-null: the length is read from device memory (written by an earlier k_newton of
                            // the same stream: chained smoothing pass), `t` is ignored
    const ModelDev *md;     // per-gene model (PROTGAMMAWAGF: empirical frequencies); null = the launch's model
};
enum { PM_FRAGS = 0, PM_FRAGS_PI = 1, PM_TIPTABLE = 2 };

// one side (child) of a CLV operation
//   SK_CLV    : p0 = CLV (80 x mpad doubles) in HBM
//   SK_TIP    : p0 = tip codes (uint8[mpad]); t0 = tip table of its branch (newview only)
//   SK_CHERRY : the child is an inner node whose two other neighbours are tips.  Its CLV is never
//               materialised: operand[c][s] = T0[c][code0][s] * T1[c][code1][s] from the two tip
//               tables (p0/p1 = codes of the two tips, t0/t1 = their tables), then contracted with
//               the fragment set of the branch to the parent like any CLV.
//   SK_PITCH  : the child X is an inner node over a cherry C (tips a,b) and a tip c ("pitchfork", 3 tips):
//               CLV_X[c][s] = (P(t_XC) . (T_a * T_b))[s] * T_c[s] is rebuilt in registers (one extra
//               contraction with the fragment set `f` of branch X-C) and then used like a CLV.
//               p0,p1,p2 = codes of a,b,c; t0,t1,t2 = their tip tables.
enum { SK_CLV = 0, SK_TIP = 1, SK_CHERRY = 2, SK_PITCH = 3 };
constexpr int OPF_NT_STORE = 16;
// Register chaining (k_oplist<9>): a wave owns the same 32 patterns in every operation of its gene, so the result of one
// newview is still in its registers when the next operation of the gene consumes it (post-order: a parent directly follows
// its last-computed child).  OPF_CHAIN_L / OPF_CHAIN_R: that side (kind SK_CLV) is taken from the registers instead of being
// read back; OPF_NO_STORE: the result is consumed that way only and is not written at all (whole-tree scoring).
// OPF_CHAIN_R is honoured on MODE_EVALUATE* operations only (a newview's chained child always goes left, engine.cpp); a
// MODE_SUMTABLE operation leaves its own tile in those registers, so nothing is chained from across one.
constexpr int OPF_CHAIN_L = 32, OPF_CHAIN_R = 64, OPF_NO_STORE = 128;
// Fused branch Newton (k_oplist<11>, round 3): a MODE_SUMTABLE operation that is the LAST operation of its gene in the launch
// does not store its table -- a workgroup's tile of it (128 patterns x 80 rows) stays in the 80 VGPRs per wave that register
// chaining reserves, and the gene's workgroups run makenewz on it in place (one exchange of three sums per evaluation, as in
// k_newton).  Saves the sumtable's write and read (2 of the 5 CLV-sized transfers of a smoothing step), k_newton's launch and ramp.
constexpr int OPF_FUSED_NEWTON = 256;
struct OpSide {
    const void *p0, *p1, *p2;
    const double *t0, *t1, *t2;
    const double *f;
};

// one CLV operation (newview / sumtable / evaluate share the descriptor)
struct NvOp {
    double *out;            // newview: CLV; sumtable: table; evaluate: per-pattern lnL
    OpSide l, r;
    int *out_scl;           // per-pattern scaling counts of the result (may be null for evaluate)
    const int *l_scl;       // null unless the side is SK_CLV
    const int *r_scl;
    const double *pl;       // fragment sets (PFRAG doubles); null for a newview SK_TIP side
    const double *pr;
    int mpad;               // padded pattern count (multiple of 32)
    int flags;              // bits 0-1: left side kind, bits 2-3: right side kind, OPF_*
    int mode;               // MODE_NEWVIEW / MODE_SUMTABLE / MODE_EVALUATE
    int pad;
    const void *aux;        // OPF_FUSED_NEWTON sumtable ops: the NewtonReq (device address) the workgroups of the gene iterate on, else null
};
static_assert(sizeof(NvOp) == 184, "NvOp layout");

// ops [op_begin, op_end) of one gene, in dependency order; executed by every pattern block
struct GeneRun {
    int op_begin, op_end;
};

struct ReduceReq {          // lnL_g = sum_p w[p] * patlnl[p]
    const double *patlnl;
    const double *weight;
    double *out;
    int mpad;
    int pad;
};

struct NewtonReq {          // Newton-Raphson on one branch from its sumtable
    const double *sumtab;   // [80][mpad]
    const double *weight;   // [mpad]
    const int *scl;         // [mpad] combined scaling counts
    double rates[NCAT];
    double t0;
    double tol;             // stop when |dt| < tol
    double *out;            // out[0]=t, out[1]=lnL, out[2]=d1, out[3]=d2 (at returned t)
    double *sync;           // NEWTON_SYNC_DOUBLES 8-byte words: the slices' {tag, value} exchange granules; never cleared: tags are unique
    const ModelDev *md;     // eigenvalues of the gene's model
    unsigned tag_base;      // (launch number << 10): granule tag = tag_base + evaluation number, so a block left by an earlier launch never matches
    unsigned pad0;
    double *t_dev0, *t_dev1; // optional: device-resident copies of the branch length (both directions) for chaining
    double *patlnl;         // optional: per-pattern lnL (scaling applied) at the returned length, [mpad] (SH-like supports)
    int mpad;
    int max_iter;           // 0: derivatives at t0 only
    int ticket0;            // first ticket of this request in its kernel's ticket table (k_newton: slice = ticket - ticket0)
    int pad;
};
constexpr int NEWTON_MAX_SPLIT = 64;    // 128-pattern slices up to 8192 patterns stay register-resident (k_newton)
// the split of a request over workgroups: a function of its pattern count alone (bit-reproducible whatever shares the launch)
constexpr int NEWTON_SLICE_PAT = 128;
__host__ __device__ constexpr int newton_split(int mpad) { return (mpad + NEWTON_SLICE_PAT - 1) / NEWTON_SLICE_PAT < NEWTON_MAX_SPLIT ? ((mpad + NEWTON_SLICE_PAT - 1) / NEWTON_SLICE_PAT < 1 ? 1 : (mpad + NEWTON_SLICE_PAT - 1) / NEWTON_SLICE_PAT) : NEWTON_MAX_SPLIT; }
// register form: a slice IS a tile of the sumtable (80 KB contiguous); beyond 64 tiles the slices grow and stream
__host__ __device__ constexpr int newton_slice(int mpad) { return mpad <= NEWTON_SLICE_PAT * NEWTON_MAX_SPLIT ? NEWTON_SLICE_PAT : ((mpad / 32 + newton_split(mpad) - 1) / newton_split(mpad)) * 32; }
__host__ __device__ constexpr bool newton_reg_form(int mpad) { return newton_slice(mpad) <= NEWTON_SLICE_PAT; }
// per (batch, stream) control block of k_newton, zeroed once at allocation: ticket / done counters of the register-form [0]
// and streaming-form [1] kernels (the last workgroup of a launch re-arms them) and the sticky abort word the host clears
// oticket / odone: k_oplist launches with fused Newton tails claim (gene, tile) by ticket as well, one counter per XCD partition
// dbg: what the first slice that gave up saw (diagnostic, printed under PML_TRACE): {1, slice, S, evaluation, mask of slices
// whose granules had arrived (low / high 32), tag wanted, 0}
struct NewtonCtl { int ticket[2]; int done[2]; int abort; int odone; int pad[2]; int oticket[8]; int dbg[8];
                   unsigned long long n_requests, n_evals; };     // diagnostic totals (PML_TRACE): Newton requests served, evaluations made
constexpr int NEWTON_SYNC_DOUBLES = 2 * NEWTON_MAX_SPLIT * 6;   // two parities x slices x six 8-byte {tag, half a double} granules (three partial sums)

// MODE_EVALUATE_CAT: like MODE_EVALUATE but the four categories are NOT averaged: out[c][p] = sum_s L_c[s] (pi P_c . R_c)[s]
// (plain likelihoods, 4 x mpad doubles) and out_scl[p] = the pattern's scaling count -- the per-site x rate likelihood
// table of FastTree's Gamma20 re-weighting, four rates per traversal
enum { MODE_NEWVIEW = 0, MODE_SUMTABLE = 1, MODE_EVALUATE = 2, MODE_EVALUATE_CAT = 3 };

// FastTree -gamma (FastTreeRunner.java:67-70): lnL of one gene under 20 FIXED rates with category weights w[k]:
//   lnL = sum_p weight[p] * ( ln( sum_k w[k] * table[k][p] * 2^(-256 (cnt[k/4][p] - m_p)) ) - m_p * 256 ln 2 ),  m_p = min_j cnt[j][p]
constexpr int G20_RATES = 20;
struct G20Req {
    const double *table;    // [20][mpad] per-pattern likelihoods, rate k in row k (five MODE_EVALUATE_CAT traversals)
    const int *cnt;         // [5][mpad] scaling counts of the five traversals
    const double *weight;   // [mpad] pattern weights
    double w[G20_RATES];    // category weights
    double *out;            // lnL
    double *patlnl;         // optional [mpad]
    int mpad, pad;
};
void launch_g20(const G20Req *reqs, int n, hipStream_t s);

// one gene's patterns copied into a replicate (jackknife concatenation on the device):
// dst[t][dst_off + p] = rowmap[t] >= 0 ? src[rowmap[t]][p] : gap code, dst_w[dst_off + p] = w[p]
struct GatherSeg {
    const uint8_t *src; const double *w; uint8_t *dst; double *dst_w; const int *rowmap;
    int src_mpad, npat, dst_mpad, dst_off, ntax_dst, pad;
};
void launch_gather(const GatherSeg *segs, int nsegs, int max_npat, hipStream_t s);

// SH-like local support of one split (FastTree's SHSupport; Guindon et al. 2010): per-pattern lnL of the current
// arrangement (l0) and of its two NNI alternatives (l1, l2); nboot resamples of nsites alignment columns drawn with
// the counter hash col(r, j) = mix64((seed+1)*0x9E3779B97F4A7C15 + r*nsites + j) % nsites; a resample supports the
// split when the centred advantage of its best arrangement is smaller than the observed advantage of l0
struct ShReq { const double *l0, *l1, *l2; const int *site2pat; double *out; unsigned long long seed; int nsites, nboot; };
void launch_sh(const ShReq *reqs, int n, hipStream_t s);

// per_request: the requests carry their own models (PmatReq::md), `model` is ignored
void launch_pmat(const ModelDev *model, const PmatReq *reqs, double *frags, int n, hipStream_t s, bool per_request = false);
// constant fragment sets for the eigen-basis transforms used by the sumtable:
//   set 0: x_i = sum_s pi_s U[s][i] A[s]     set 1: y_i = sum_j Uinv[i][j] B[j]
void launch_eigfrags(const ModelDev *model, double *frags2, hipStream_t s);
// n models in an array, n x 2*PFRAG doubles out
void launch_eigfrags_n(const ModelDev *models, double *frags2, int n, hipStream_t s);
// any_pitch: some op has an SK_PITCH side (two more LDS fragment regions are allocated)
// ctl != null: the launch has fused Newton tails (OPF_FUSED_NEWTON): chained variant, (gene, tile) claimed by ticket
void launch_oplist(const NvOp *ops, const GeneRun *runs, int nruns, int max_mpad, bool any_pitch, bool chained, hipStream_t s, NewtonCtl *ctl = nullptr);
// workgroups of the fused-Newton op-list kernel the idle device holds at once (2 per CU)
int fused_oplist_capacity();
// genes of more than 32 tiles are fused in ONE launch with one ticket partition over the device (default; PML_FUSE_BIG=0: only
// when the whole launch is resident at once)
bool fuse_big_genes();
void launch_reduce(const ReduceReq *reqs, int n, hipStream_t s);
// tickets: request index per ticket, register-form tickets [0, nreg) then streaming-form [nreg, nreg + nstream)
void launch_newton(const ModelDev *model, const NewtonReq *reqs, const int *tickets, int nreg, int nstream, NewtonCtl *ctl, hipStream_t s);
// the no-exchange fallback: one workgroup per listed request (register-form requests first), same bits as the split form
void launch_newton_seq(const ModelDev *model, const NewtonReq *reqs, const int *req_list, int nreg, int nstream, NewtonCtl *ctl, hipStream_t s);

}  // namespace pml
