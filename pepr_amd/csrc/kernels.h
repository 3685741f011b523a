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
constexpr int NCODES = 23;

// device-resident model constants (one per ctx)
struct ModelDev {
    double eval[NS];
    double U[NS * NS];      // row-major U[s][k]
    double Uinv[NS * NS];   // Uinv[k][j]
    double pi[NS];
};

// request for one transition-matrix fragment set: P(t * rate_c), c = 0..3
struct PmatReq {
    double t;
    double rates[NCAT];
    int fold_pi;            // 1: rows scaled by pi_s (root evaluation)
    int pad;
};

// one CLV operation (newview / sumtable / evaluate share the descriptor)
struct NvOp {
    double *out;            // newview: CLV; sumtable: table; evaluate: per-pattern lnL
    const void *left;       // CLV (double*) or tip codes (uint8*)
    const void *right;
    int *out_scl;           // per-pattern scaling counts of the result (may be null for evaluate)
    const int *l_scl;       // null for tips
    const int *r_scl;
    const double *pl;       // fragment sets (PFRAG doubles)
    const double *pr;
    int mpad;               // padded pattern count (multiple of 32)
    int flags;              // bit0: left is tip, bit1: right is tip
    int mode;               // MODE_NEWVIEW / MODE_SUMTABLE / MODE_EVALUATE
    int pad;
    double *aux;            // sumtable ops: the Newton sync block to zero (NEWTON_SYNC_DOUBLES), else null
};
static_assert(sizeof(NvOp) == 88, "NvOp layout");

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
    double *sync;           // NEWTON_SYNC_DOUBLES zeroed doubles: arrival counter + per-workgroup partial sums
    int mpad;
    int max_iter;           // 0: derivatives at t0 only
};
constexpr int NEWTON_MAX_SPLIT = 8;
constexpr int NEWTON_SYNC_DOUBLES = 2 + 2 * NEWTON_MAX_SPLIT * 3 + 14;   // 64 doubles = 512 B

enum { MODE_NEWVIEW = 0, MODE_SUMTABLE = 1, MODE_EVALUATE = 2 };

void launch_pmat(const ModelDev *model, const PmatReq *reqs, double *frags, int n, hipStream_t s);
// constant fragment sets for the eigen-basis transforms used by the sumtable:
//   set 0: x_i = sum_s pi_s U[s][i] A[s]     set 1: y_i = sum_j Uinv[i][j] B[j]
void launch_eigfrags(const ModelDev *model, double *frags2, hipStream_t s);
void launch_oplist(const NvOp *ops, const GeneRun *runs, int nruns, int max_mpad, hipStream_t s);
void launch_reduce(const ReduceReq *reqs, int n, hipStream_t s);
void launch_newton(const ModelDev *model, const NewtonReq *reqs, int n, int max_mpad, hipStream_t s);

}  // namespace pml
