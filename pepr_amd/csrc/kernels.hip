// kernels.hip -- hand-written CDNA4 (gfx950) kernels of the likelihood engine.
//
// Data layout in HBM (DESIGN.md "Layout"): a conditional-likelihood vector set (CLV) of one
// directed tree edge is 80 rows x mpad doubles, row = cat*20 + state, patterns contiguous
// (structure-of-arrays), mpad a multiple of 32.  A wave owns 32 consecutive patterns ("chunk"):
// lane (j = lane&15, q = lane>>4) loads 16 B = patterns {2j, 2j+1} of state row 4*kk+q, so a
// wave-load is four 256-byte segments.
//
// The 20x20 transition-matrix x CLV contraction runs on v_mfma_f64_4x4x4_4b_f64 (4 blocks of
// 4x4x4; measured 18 cycles/instruction = 28 flop/clk/SIMD on MI355X versus >=100 cycles for
// v_mfma_f64_16x16x4_f64, profiles/r01_ubench_f64.txt; 20 = 5x4 so no padding is wasted).
// Lane map measured on the device (tools/probe_mfma444.hip): k = lane>>4 for A and B, the
// block is (lane>>2)&3, A row i = lane&3, B/D column j = lane&3, D row i = lane>>4.  With the
// four blocks = four groups of 4 patterns, lane (j16 = lane&15, q = lane>>4) feeds B =
// CLV[state 4kk+q][pattern j16] and receives D = out[state 4st+q][pattern j16]: loads and stores
// use the same address pattern.  The A operand P[4st + (lane&3)][4kk + q] is identical in all
// blocks; the 25 (st,kk) fragments x 4 categories are pre-arranged by k_pmat (12.8 KB per
// branch) and staged once per workgroup and op in LDS.
//
// Execution model: one launch runs a whole dependency-ordered op list per gene.  A workgroup
// owns 128 patterns of one gene and walks that gene's ops sequentially; site patterns are
// independent, so no inter-workgroup synchronisation exists and tree levels overlap freely.
//
// What these kernels replace in the reference: the arithmetic inside the external programs
// spawned at RAxMLRunner.java:147 and FastTreeRunner.java:94 (newview / evaluate / makenewz of
// RAxML 7.2.5, SURVEY.md section 8a-11 iii-v).
#include "kernels.h"

namespace pml {

#define TWO_P256 1.15792089237316195423570985008687907853269984665640564039457584007913129639936e77
#define TWO_M256 8.63616855509444462538635186280017219262570580443837358382049248904e-78
#define LOG_2_256 177.445678223345993274051579105116

__device__ __forceinline__ double mfma4(double a, double b, double c) {
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ unsigned code_mask(unsigned code) {
    // 0..19 single state; 20 = B (N|D); 21 = Z (Q|E); else gap / unknown = all states
    return code < 20u ? (1u << code) : (code == 20u ? 0xCu : (code == 21u ? 0x60u : 0xFFFFFu));
}

// fragment element e = 4*k + i of fragment (c, st, kk) holds M_c[4 st + i][4 kk + k]
__device__ __forceinline__ void frag_decode(int idx, int &c, int &row, int &col) {
    const int e = idx & 15, f = idx >> 4;          // f = c*25 + st*5 + kk
    c = f / 25;
    const int st = (f % 25) / 5, kk = f % 5;
    row = 4 * st + (e & 3); col = 4 * kk + (e >> 2);
}

// ------------------------------------------------------------------------------------------
// k_pmat: P(t r_c) = U diag(exp(lambda t r_c)) U^-1, written directly in MFMA A-fragment order
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pmat(const ModelDev *__restrict__ md,
                                              const PmatReq *__restrict__ reqs,
                                              double *__restrict__ frags) {
    __shared__ double e[NCAT * NS];
    __shared__ double sU[NS * NS], sUi[NS * NS];
    const PmatReq req = reqs[blockIdx.x];
    const int tid = threadIdx.x;
    for (int i = tid; i < NS * NS; i += 256) { sU[i] = md->U[i]; sUi[i] = md->Uinv[i]; }
    if (tid < NCAT * NS) e[tid] = exp(md->eval[tid % NS] * (req.t * req.rates[tid / NS]));
    __syncthreads();
    double *out = frags + (size_t)blockIdx.x * PFRAG;
    for (int idx = tid; idx < PFRAG; idx += 256) {
        int c, s, j;
        frag_decode(idx, c, s, j);
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < NS; ++k) v += sU[s * NS + k] * e[c * NS + k] * sUi[k * NS + j];
        if (v < 0.0) v = 0.0;
        if (req.fold_pi) v *= md->pi[s];
        out[idx] = v;
    }
}

__global__ __launch_bounds__(256) void k_eigfrags(const ModelDev *__restrict__ md,
                                                  double *__restrict__ frags2) {
    for (int idx = threadIdx.x; idx < PFRAG; idx += 256) {
        int c, i, s;
        frag_decode(idx, c, i, s);
        frags2[idx] = md->pi[s] * md->U[s * NS + i];      // x_i = sum_s pi_s U[s][i] A[s]
        frags2[PFRAG + idx] = md->Uinv[i * NS + s];       // y_i = sum_j Uinv[i][j] B[j]
    }
}

// ------------------------------------------------------------------------------------------
// CLV op on one chunk (32 patterns) of one wave.
//   MODE_NEWVIEW : out[c][s] = (P_L,c . L_c)[s] * (P_R,c . R_c)[s], 2^256 rescue, scaling counts
//   MODE_SUMTABLE: same contraction with the eigen-basis matrices (no rescue), counts = l + r
//   MODE_EVALUATE: per-pattern ln( 1/4 sum_c sum_s L_c[s] (pi P_c . R_c)[s] ) - counts*256 ln 2
// ------------------------------------------------------------------------------------------
struct Operand {            // one child's 5 double2 B operands of one category
    double2 v[5];
};

template <bool TIP>
__device__ __forceinline__ void load_operand(Operand &o, const double *__restrict__ base, size_t M, int c, int q, int p,
                                             unsigned m0, unsigned m1) {
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) {
        const int st = kk * 4 + q;
        if (TIP) o.v[kk] = make_double2((m0 >> st) & 1u ? 1.0 : 0.0, (m1 >> st) & 1u ? 1.0 : 0.0);
        else o.v[kk] = *reinterpret_cast<const double2 *>(base + (size_t)(c * NS + st) * M + p);
    }
}

// acc[st][0/1] = sum_kk frag(c,st,kk) x operand(kk)   (even / odd pattern of the lane)
__device__ __forceinline__ void contract(double (&acc)[5][2], const double *__restrict__ frag_c, const Operand &o) {
#pragma unroll
    for (int st = 0; st < 5; ++st) { acc[st][0] = 0.0; acc[st][1] = 0.0; }
#pragma unroll
    for (int kk = 0; kk < 5; ++kk)
#pragma unroll
        for (int st = 0; st < 5; ++st) {
            const double a = frag_c[(st * 5 + kk) * 16];
            acc[st][0] = mfma4(a, o.v[kk].x, acc[st][0]);
            acc[st][1] = mfma4(a, o.v[kk].y, acc[st][1]);
        }
}

template <int MODE, bool LTIP, bool RTIP>
__device__ __forceinline__ void chunk_op(const NvOp &op, const double *__restrict__ sP, int p, int lane) {
    const int q = lane >> 4;
    const size_t M = (size_t)op.mpad;
    const double *__restrict__ Lp = static_cast<const double *>(op.left);
    const double *__restrict__ Rp = static_cast<const double *>(op.right);
    unsigned mL0 = 0, mL1 = 0, mR0 = 0, mR1 = 0;
    if (LTIP) {
        const unsigned cc = *reinterpret_cast<const unsigned short *>(static_cast<const unsigned char *>(op.left) + p);
        mL0 = code_mask(cc & 0xFFu); mL1 = code_mask(cc >> 8);
    }
    if (RTIP) {
        const unsigned cc = *reinterpret_cast<const unsigned short *>(static_cast<const unsigned char *>(op.right) + p);
        mR0 = code_mask(cc & 0xFFu); mR1 = code_mask(cc >> 8);
    }
    // lane's A-fragment element: 4*k + i with k = q, i = lane&3
    const double *fL = sP + (q * 4 + (lane & 3));
    const double *fR = fL + PFRAG;
    double mx0 = 0.0, mx1 = 0.0, site0 = 0.0, site1 = 0.0;

    Operand curL, curR, nxtL, nxtR;
    load_operand<LTIP>(curL, Lp, M, 0, q, p, mL0, mL1);     // evaluate: left side in output layout
    load_operand<RTIP>(curR, Rp, M, 0, q, p, mR0, mR1);
#pragma unroll
    for (int c = 0; c < NCAT; ++c) {
        if (c + 1 < NCAT) {                                 // software prefetch of the next category
            load_operand<LTIP>(nxtL, Lp, M, c + 1, q, p, mL0, mL1);
            load_operand<RTIP>(nxtR, Rp, M, c + 1, q, p, mR0, mR1);
        }
        double aR[5][2];
        contract(aR, fR + c * 25 * 16, curR);
        if (MODE == MODE_EVALUATE) {
#pragma unroll
            for (int st = 0; st < 5; ++st) { site0 += curL.v[st].x * aR[st][0]; site1 += curL.v[st].y * aR[st][1]; }
        } else {
            double aL[5][2];
            contract(aL, fL + c * 25 * 16, curL);
            double *__restrict__ O = op.out;
#pragma unroll
            for (int st = 0; st < 5; ++st) {
                const double o0 = aL[st][0] * aR[st][0], o1 = aL[st][1] * aR[st][1];
                mx0 = fmax(mx0, o0); mx1 = fmax(mx1, o1);
                *reinterpret_cast<double2 *>(O + (size_t)(c * NS + st * 4 + q) * M + p) = make_double2(o0, o1);
            }
        }
        if (c + 1 < NCAT) { curL = nxtL; curR = nxtR; }
    }

    int2 sc = make_int2(0, 0);
    if (q == 0) {
        if (!LTIP) { const int2 a = *reinterpret_cast<const int2 *>(op.l_scl + p); sc.x += a.x; sc.y += a.y; }
        if (!RTIP) { const int2 a = *reinterpret_cast<const int2 *>(op.r_scl + p); sc.x += a.x; sc.y += a.y; }
    }
    if (MODE == MODE_NEWVIEW) {
        mx0 = fmax(mx0, __shfl_xor(mx0, 16)); mx0 = fmax(mx0, __shfl_xor(mx0, 32));
        mx1 = fmax(mx1, __shfl_xor(mx1, 16)); mx1 = fmax(mx1, __shfl_xor(mx1, 32));
        const bool n0 = mx0 < TWO_M256, n1 = mx1 < TWO_M256;
        if (__any(n0 || n1)) {               // rare: numerical rescue of underflowing patterns
            if (n0 || n1) {
                double *O = op.out;
                for (int c = 0; c < NCAT; ++c)
                    for (int st = 0; st < 5; ++st) {
                        double2 *ptr = reinterpret_cast<double2 *>(O + (size_t)(c * NS + st * 4 + q) * M + p);
                        double2 v = *ptr;
                        if (n0) v.x *= TWO_P256;
                        if (n1) v.y *= TWO_P256;
                        *ptr = v;
                    }
            }
        }
        if (q == 0) { sc.x += n0 ? 1 : 0; sc.y += n1 ? 1 : 0; *reinterpret_cast<int2 *>(op.out_scl + p) = sc; }
    } else if (MODE == MODE_SUMTABLE) {
        if (q == 0) *reinterpret_cast<int2 *>(op.out_scl + p) = sc;
    } else {
        site0 += __shfl_xor(site0, 16); site0 += __shfl_xor(site0, 32);
        site1 += __shfl_xor(site1, 16); site1 += __shfl_xor(site1, 32);
        if (q == 0) {
            const double l0 = log(site0 * 0.25) - sc.x * LOG_2_256;
            const double l1 = log(site1 * 0.25) - sc.y * LOG_2_256;
            *reinterpret_cast<double2 *>(op.out + p) = make_double2(l0, l1);
        }
    }
}

template <int MODE>
__device__ __forceinline__ void chunk_dispatch(const NvOp &op, const double *sP, int p, int lane) {
    switch (op.flags & 3) {
        case 0: chunk_op<MODE, false, false>(op, sP, p, lane); break;
        case 1: chunk_op<MODE, true, false>(op, sP, p, lane); break;
        case 2: chunk_op<MODE, false, true>(op, sP, p, lane); break;
        default: chunk_op<MODE, true, true>(op, sP, p, lane); break;
    }
}

// ------------------------------------------------------------------------------------------
// k_oplist: workgroup (gene, pattern block of 128) executes the gene's op list in order.
// ------------------------------------------------------------------------------------------
constexpr int PAT_PER_WG = 4 * PAT_PER_WAVE;   // 128

__global__ __launch_bounds__(256, 2) void k_oplist(const NvOp *__restrict__ ops,
                                                   const GeneRun *__restrict__ runs, int blocks_per_gene) {
    __shared__ double sP[2 * PFRAG];   // 25.6 KB: [left|right][cat][st][kk][16]
    const int gi = blockIdx.x / blocks_per_gene, blk = blockIdx.x % blocks_per_gene;
    const GeneRun run = runs[gi];
    if (run.op_begin >= run.op_end) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mpad = ops[run.op_begin].mpad;            // constant per gene
    if (blk * PAT_PER_WG >= mpad) return;
    const int p = (blk * 4 + wave) * PAT_PER_WAVE + 2 * (lane & 15);
    const bool active = (blk * 4 + wave) * PAT_PER_WAVE < mpad;

    for (int oi = run.op_begin; oi < run.op_end; ++oi) {
        const NvOp op = ops[oi];
        __syncthreads();                 // previous op: LDS reads and global stores complete
        {
            const double2 *gl = reinterpret_cast<const double2 *>(op.pl);
            const double2 *gr = reinterpret_cast<const double2 *>(op.pr);
            double2 *s2 = reinterpret_cast<double2 *>(sP);
            for (int i = tid; i < PFRAG / 2; i += 256) {
                if (op.mode != MODE_EVALUATE) s2[i] = gl[i];
                s2[PFRAG / 2 + i] = gr[i];
            }
        }
        __syncthreads();
        if (active) {
            if (op.mode == MODE_NEWVIEW) chunk_dispatch<MODE_NEWVIEW>(op, sP, p, lane);
            else if (op.mode == MODE_SUMTABLE) chunk_dispatch<MODE_SUMTABLE>(op, sP, p, lane);
            else chunk_dispatch<MODE_EVALUATE>(op, sP, p, lane);
        }
    }
}

// ------------------------------------------------------------------------------------------
// deterministic block reduction (fixed order: wave shuffle tree, then waves 0..3 in order)
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}
template <int N>
__device__ __forceinline__ void block_sum(double (&v)[N], double (*red)[4]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) { const double s = wave_sum(v[i]); if (lane == 0) red[i][wave] = s; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = ((red[i][0] + red[i][1]) + red[i][2]) + red[i][3];
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_reduce(const ReduceReq *__restrict__ reqs) {
    __shared__ double red[1][4];
    const ReduceReq r = reqs[blockIdx.x];
    double acc[1] = {0.0};
    for (int p = threadIdx.x; p < r.mpad; p += 256) {
        const double w = r.weight[p];
        if (w != 0.0) acc[0] += w * r.patlnl[p];
    }
    block_sum<1>(acc, red);
    if (threadIdx.x == 0) *r.out = acc[0];
}

// ------------------------------------------------------------------------------------------
// k_newton: Newton-Raphson on one branch length from its eigen-basis sumtable (RAxML "makenewz",
// SURVEY 8a-11 v).  One workgroup per (gene, branch); the whole iteration runs on the device.
// Control flow is the oracle's eng_newton_branch(), evaluated redundantly by every thread.
// ------------------------------------------------------------------------------------------
#define PML_TMIN 1.0e-6
#define PML_TMAX 34.5

__global__ __launch_bounds__(256) void k_newton(const ModelDev *__restrict__ md,
                                                const NewtonReq *__restrict__ reqs) {
    __shared__ double ex[3][NCAT * NS];
    __shared__ double red[3][4];
    const NewtonReq r = reqs[blockIdx.x];
    const int tid = threadIdx.x, mpad = r.mpad;
    const size_t M = (size_t)mpad;

    auto eval_at = [&](double t, double &L, double &d1, double &d2) {
        if (tid < NCAT * NS) {
            const double lr = md->eval[tid % NS] * r.rates[tid / NS];
            const double e = exp(lr * t);
            ex[0][tid] = e; ex[1][tid] = lr * e; ex[2][tid] = lr * lr * e;
        }
        __syncthreads();
        double acc[3] = {0.0, 0.0, 0.0};
        for (int p = tid; p < mpad; p += 256) {
            const double w = r.weight[p];
            if (w == 0.0) continue;
            double f = 0.0, f1 = 0.0, f2 = 0.0;
#pragma unroll 8
            for (int row = 0; row < CLV_ROWS; ++row) {
                const double x = r.sumtab[(size_t)row * M + p];
                f += x * ex[0][row]; f1 += x * ex[1][row]; f2 += x * ex[2][row];
            }
            const double r1 = f1 / f;
            acc[0] += w * (log(f * 0.25) - r.scl[p] * LOG_2_256);
            acc[1] += w * r1;
            acc[2] += w * (f2 / f - r1 * r1);
        }
        block_sum<3>(acc, red);
        L = acc[0]; d1 = acc[1]; d2 = acc[2];
    };

    double t = r.t0;
    if (r.max_iter > 0) t = t < PML_TMIN ? PML_TMIN : (t > PML_TMAX ? PML_TMAX : t);
    double L, d1, d2;
    eval_at(t, L, d1, d2);
    for (int it = 0; it < r.max_iter; ++it) {
        const double step = (d2 < 0.0) ? -d1 / d2 : (d1 > 0.0 ? t : -0.5 * t);
        double tn = t + step, Ln, n1, n2;
        int bt = 0;
        for (;;) {
            tn = tn < PML_TMIN ? PML_TMIN : (tn > PML_TMAX ? PML_TMAX : tn);
            eval_at(tn, Ln, n1, n2);
            if (Ln >= L - 1e-9 || bt >= 8) break;
            ++bt; tn = 0.5 * (tn + t);
        }
        if (Ln < L - 1e-9) break;
        const double dt = fabs(tn - t);
        t = tn; L = Ln; d1 = n1; d2 = n2;
        if (dt < 1e-8) break;
    }
    if (tid == 0) { r.out[0] = t; r.out[1] = L; r.out[2] = d1; r.out[3] = d2; }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
void launch_pmat(const ModelDev *model, const PmatReq *reqs, double *frags, int n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_pmat, dim3(n), dim3(256), 0, s, model, reqs, frags);
}
void launch_eigfrags(const ModelDev *model, double *frags2, hipStream_t s) {
    hipLaunchKernelGGL(k_eigfrags, dim3(1), dim3(256), 0, s, model, frags2);
}
void launch_oplist(const NvOp *ops, const GeneRun *runs, int nruns, int max_mpad, hipStream_t s) {
    if (nruns <= 0) return;
    const int bpg = (max_mpad + PAT_PER_WG - 1) / PAT_PER_WG;
    hipLaunchKernelGGL(k_oplist, dim3((unsigned)(nruns * bpg)), dim3(256), 0, s, ops, runs, bpg);
}
void launch_reduce(const ReduceReq *reqs, int n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_reduce, dim3(n), dim3(256), 0, s, reqs);
}
void launch_newton(const ModelDev *model, const NewtonReq *reqs, int n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_newton, dim3(n), dim3(256), 0, s, model, reqs);
}

}  // namespace pml
