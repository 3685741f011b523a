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

#include <algorithm>
#include <cstdlib>

namespace pml {

#define TWO_P256 1.15792089237316195423570985008687907853269984665640564039457584007913129639936e77
#define TWO_M256 8.63616855509444462538635186280017219262570580443837358382049248904e-78
#define LOG_2_256 177.445678223345993274051579105116

__device__ __forceinline__ double mfma4(double a, double b, double c) {
#ifdef ABL_NO_MFMA
    return __builtin_fma(a, b, c);       // timing-only ablation: the dependency without the matrix pipe
#endif
    return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ unsigned code_mask(unsigned code) {
    // 0..19 single state; 20 = B (N|D); 21 = Z (Q|E); else gap / unknown = all states
    return code < 20u ? (1u << code) : (code == 20u ? 0xCu : (code == 21u ? 0x60u : 0xFFFFFu));
}

template <int CTRL>           // DPP quad_perm: 0xB1 = [1,0,3,2] (lane ^ 1), 0x4E = [2,3,0,1] (lane ^ 2)
__device__ __forceinline__ double quad_swap(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double quad_sum(double v) { v += quad_swap<0xB1>(v); v += quad_swap<0x4E>(v); return v; }

// fragment element e = 4*k + i of fragment (c, st, kk) holds M_c[4 st + i][4 kk + k]
__device__ __forceinline__ void frag_decode(int idx, int &c, int &row, int &col) {
    const int e = idx & 15, f = idx >> 4;          // f = c*25 + st*5 + kk
    c = f / 25;
    const int st = (f % 25) / 5, kk = f % 5;
    row = 4 * st + (e & 3); col = 4 * kk + (e >> 2);
}

// ------------------------------------------------------------------------------------------
// k_pmat: P(t r_c) = U diag(exp(lambda t r_c)) U^-1 for all four categories of a request, ON THE MATRIX PIPE.
// One wave per request (persistent: a wave walks requests w, w + W, ...).  v_mfma_f64_4x4x4_4b computes four independent
// 4x4x4 products per instruction: the four BLOCKS are the four rate categories, so one instruction adds one k-step of one
// 4x4 tile of P for every category at once:  D_c[i][j] += sum_k (U[4st+i][4kk+k] e_c[4kk+k]) * Uinv[4kk+k][4nt+j].
// 25 tiles x 5 k-steps = 125 instructions per request (18 cycles each), against 32 k f64 FMAs on the vector pipe in the
// round-1 kernel (one 320-thread block per request, two barriers, LDS staging: 0.122 ms for the 12.4 k requests of a C3
// scoring step, 12 % of the step; VALU floor 0.05 ms).  The U / Uinv operands of a lane are the same for every request
// and stay in 100 VGPRs; per request a lane needs 2 exponentials (80 per request, exchanged through a wave-private LDS
// row) and 25 multiplies.  Lane map of the instruction (tools/probe_mfma444.hip): A[i][k]: i = lane&3, k = lane>>4;
// B[k][j]: k = lane>>4, j = lane&3; D[i][j]: i = lane>>4, j = lane&3; block = (lane>>2)&3 throughout.
// Outputs: PM_FRAGS / PM_FRAGS_PI in MFMA A-fragment order for k_oplist; PM_TIPTABLE T[c][code][q][kk] =
// sum_{j in states(code)} P_c[4 kk + q][j] (the contraction of a tip's indicator vector, by lookup).
// ------------------------------------------------------------------------------------------
constexpr int PMAT_THREADS = 256;
// PER_REQ: every request names its own model (PROTGAMMAWAGF: per-gene empirical frequencies, hence per-gene eigen-systems):
// the lane's 50 U / Uinv operands, two eigenvalues and five frequencies are re-read (L2) for each request instead of once
template <bool PER_REQ>
__global__ __launch_bounds__(PMAT_THREADS, 2) void k_pmat(const ModelDev *__restrict__ md0,
                                                          const PmatReq *__restrict__ reqs,
                                                          double *__restrict__ frags, int n) {
    __shared__ double sE[PMAT_THREADS / 64][NCAT * NS];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i4 = lane & 3, k4 = lane >> 4, cb = (lane >> 2) & 3;
    double Au[5][5], Bu[5][5];
    double lam0, lam1, pi_row[5];                // D row = 4 st + (lane>>4)
    // exponentials: lane l owns table rows l and (l < 16) 64 + l; row r = category r / 20, eigenvalue r % 20
    const int cat0 = lane / NS, cat1 = (64 + (lane & 15)) / NS;
    auto load_model = [&](const ModelDev *__restrict__ md) {
#pragma unroll
        for (int a = 0; a < 5; ++a)
#pragma unroll
            for (int b = 0; b < 5; ++b) {
                Au[a][b] = md->U[(4 * a + i4) * NS + 4 * b + k4];          // [st][kk]
                Bu[a][b] = md->Uinv[(4 * a + k4) * NS + 4 * b + i4];       // [kk][nt]
            }
        lam0 = md->eval[lane % NS]; lam1 = md->eval[(64 + (lane & 15)) % NS];
#pragma unroll
        for (int a = 0; a < 5; ++a) pi_row[a] = md->pi[4 * a + k4];
    };
    if (!PER_REQ) load_model(md0);
    const int nwaves = gridDim.x * (PMAT_THREADS / 64);
    for (int rq = blockIdx.x * (PMAT_THREADS / 64) + wv; rq < n; rq += nwaves) {
        const PmatReq &req = reqs[rq];
        if (PER_REQ) load_model(req.md);
        const double tlen = req.tp ? *req.tp : req.t;
        const int kind = req.kind;
        double *sEw = sE[wv];
        sEw[lane] = exp(lam0 * (tlen * req.rates[cat0]));
        if (lane < 16) sEw[64 + lane] = exp(lam1 * (tlen * req.rates[cat1]));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        double ek[5];
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) ek[kk] = sEw[cb * NS + 4 * kk + k4];
        __builtin_amdgcn_wave_barrier();                       // every lane has its five values before the row is rewritten
        double *out = frags + (size_t)rq * FRAG_STRIDE;
        double T[5][5];                                        // [st][nt]: the lane's element of tile (st, nt) in its block's category
#pragma unroll
        for (int st = 0; st < 5; ++st) {
            double a[5];
#pragma unroll
            for (int kk = 0; kk < 5; ++kk) a[kk] = Au[st][kk] * ek[kk];
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) {
                double d = 0.0;
#pragma unroll
                for (int kk = 0; kk < 5; ++kk) d = mfma4(a[kk], Bu[kk][nt], d);
                T[st][nt] = d < 0.0 ? 0.0 : d;
            }
        }
        if (kind != PM_TIPTABLE) {
#pragma unroll
            for (int st = 0; st < 5; ++st) {
                const double scale = kind == PM_FRAGS_PI ? pi_row[st] : 1.0;
#pragma unroll
                for (int nt = 0; nt < 5; ++nt)                 // fragment (c, st, kk = nt), element 4 * (column in tile) + (row in tile)
                    out[((cb * 25 + st * 5 + nt) << 4) + (i4 << 2) + k4] = T[st][nt] * scale;
            }
        } else {
            // T[c][code][q = row in tile][kk = st]: a record is 6 doubles (5 + padding) = three 16-byte stores of one lane
            auto put = [&](int code, double v0, double v1, double v2, double v3, double v4) {
                typedef double dv2 __attribute__((ext_vector_type(2)));
                dv2 *rec = reinterpret_cast<dv2 *>(out + ((cb * NCODES + code) * 4 + k4) * TIPTAB_KK);
                rec[0] = (dv2){v0, v1}; rec[1] = (dv2){v2, v3}; rec[2] = (dv2){v4, 0.0};
            };
#pragma unroll
            for (int nt = 0; nt < 5; ++nt) put(4 * nt + i4, T[0][nt], T[1][nt], T[2][nt], T[3][nt], T[4][nt]);   // plain states: column 4 nt + (lane&3)
            double bz[5], zq[5], any[5];
#pragma unroll
            for (int st = 0; st < 5; ++st) {
                bz[st] = T[st][0] + quad_swap<0xB1>(T[st][0]);             // columns {0,1} / {2,3} of tile 0 pairwise
                zq[st] = T[st][1] + quad_swap<0xD8>(T[st][1]);             // quad_perm [0,2,1,3]: lanes 1 and 2 exchange
                any[st] = quad_sum(T[st][0] + T[st][1] + T[st][2] + T[st][3] + T[st][4]);
            }
            if (i4 == 2) put(20, bz[0], bz[1], bz[2], bz[3], bz[4]);       // B = N | D (columns 2, 3)
            if (i4 == 1) put(21, zq[0], zq[1], zq[2], zq[3], zq[4]);       // Z = Q | E (columns 5, 6)
            if (i4 == 0) put(22, any[0], any[1], any[2], any[3], any[4]);  // gap / unknown: every state
        }
    }
}

__global__ __launch_bounds__(256) void k_eigfrags(const ModelDev *__restrict__ models,
                                                  double *__restrict__ frags2_all) {
    const ModelDev *__restrict__ md = models + blockIdx.x;                 // one model per workgroup
    double *__restrict__ frags2 = frags2_all + (size_t)blockIdx.x * 2 * PFRAG;
    for (int idx = threadIdx.x; idx < PFRAG; idx += 256) {
        int c, i, s;
        frag_decode(idx, c, i, s);
        frags2[idx] = md->pi[s] * md->U[s * NS + i];      // x_i = sum_s pi_s U[s][i] A[s]
        frags2[PFRAG + idx] = md->Uinv[i * NS + s];       // y_i = sum_j Uinv[i][j] B[j]
    }
}

// ------------------------------------------------------------------------------------------
// Branch Newton (makenewz): the pieces k_newton and the fused form inside k_oplist<11> share.  Both forms MUST produce the same
// bits (the unfused + no-exchange form is the fallback of the fused one), so everything from a sumtable row to the Newton
// step is written once, with floating-point contraction pinned where the two call sites could otherwise be compiled differently:
//   per pattern   lane quarter q adds its 20 rows  c*20 + 4*st + q  (c = 0..3 outer, st = 0..4 inner; the rows an MFMA D tile
//                 leaves in that lane) through newton_term, quarters are combined as (q0 + q1) + (q2 + q3);
//   per slice     newton_finish per pattern, lane l of finishing wave A / B owns pattern l / 64 + l, wave_sum tree, A + B;
//   per request   the slices' sums through newton_exchange (fixed shuffle tree over the slices), then newton_drive.
// ------------------------------------------------------------------------------------------
#define PML_TMIN 1.0e-6
#define PML_TMAX 34.5

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
#ifdef PML_OPTIME       // diagnostic build: cycle counters (tools/optime.sh, tools/optime_search.py); see g_optime below
extern __device__ unsigned long long g_optime[256][2];
#endif
__device__ __forceinline__ void st_granule(u64 *p, u64 v) {
    __hip_atomic_store((gu64 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 ld_granule(const u64 *p) {
    return __hip_atomic_load((const gu64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave-uniform double kept in an SGPR pair (the Newton state is identical in every lane; as VGPRs it would cost 16 of 64)
__device__ __forceinline__ double uni(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
// deterministic wave reduction: fixed shuffle tree
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}

// one sumtable row x of a pattern at exp(lambda r t) = e, lambda r = lr:  f += x e,  f' += x e lr,  f'' += x e lr^2
__device__ __forceinline__ void newton_term(double x, double e, double lr, double &f, double &f1, double &f2) {
#pragma clang fp contract(off)
    const double xe = x * e;
    const double xl = xe * lr;
    f = f + xe; f1 = f1 + xl; f2 = __builtin_fma(xl, lr, f2);
}
// a pattern's contribution to (lnL, dlnL/dt, d2lnL/dt2) from its f, f', f'' (weight w != 0, scaling count ss)
__device__ __forceinline__ void newton_finish(double f, double f1, double f2, double w, double ss, double &a0, double &a1, double &a2) {
#pragma clang fp contract(off)
    const double r1 = f1 / f;
    const double lg = log(f * 0.25);
    a0 = w * (lg - ss * LOG_2_256); a1 = w * r1; a2 = w * (f2 / f - r1 * r1);
}

// Cross-workgroup exchange of one evaluation's three partial sums (executed by ONE whole wave of every slice's workgroup).
// Data-tagged granules (cdna_hip_programming.md Guideline 16, form R2): every partial sum travels as two naturally aligned
// 8-byte words {tag, 32 bits of the double}, each written by ONE relaxed agent-scope atomic store and read by relaxed
// agent-scope atomic loads.  A granule is indivisible, so a consumer can never pair a tag with bytes of another evaluation;
// no ordering between different words is needed (no flag, no fence, no s_waitcnt).  A slot is only overwritten two
// evaluations later (parity double buffer), which its producer can reach only after every consumer has finished the
// evaluation in between.  tag = tag_base (unique per launch) + evaluation number, so the block is never cleared.  The sums
// of all slices are combined in a fixed tree order that depends on S alone, S depends on the request alone: bit-reproducible
// whatever else is in the launch.  Returns true when the wait gave up (wall-clock bound or the launch-wide abort word).
__device__ __forceinline__ bool newton_exchange(u64 *gran, int S, int wg, int nevals, unsigned tag_base, double (&tot)[3],
                                                NewtonCtl *ctl, long long timeout_ticks) {
    const int lane = threadIdx.x & 63;
    const u64 want = (u64)(tag_base + (unsigned)nevals + 1u);
    u64 *slot = gran + (size_t)((nevals & 1) * NEWTON_MAX_SPLIT) * 6;
    if (lane < 6) {              // publish: six granules {tag, half of a double}, one lane each
        const int c = lane >> 1;
        const u64 bits = (u64)__double_as_longlong(c == 0 ? tot[0] : (c == 1 ? tot[1] : tot[2]));
        st_granule(slot + wg * 6 + lane, (want << 32) | ((lane & 1) ? (bits >> 32) : (bits & 0xFFFFFFFFull)));
    }
    u64 x[6] = {0, 0, 0, 0, 0, 0};
#ifdef PML_OPTIME
    const long long t_poll0 = clock64();
#endif
    const long long t_start = wall_clock64();
    unsigned polls = 0;
    bool bad = false;
    for (;;) {                   // gather: lane l re-reads slice l's granules until all six carry this evaluation's tag
        bool ok = true;
        if (lane < S) {
#pragma unroll
            for (int k = 0; k < 6; ++k) { x[k] = ld_granule(slot + lane * 6 + k); ok = ok && (x[k] >> 32) == want; }
        }
        if (__all(ok)) break;
        // Back-off: a partner that is only a few microseconds behind (the normal case) is met by the first, dense polls; a slice
        // whose partners have not even STARTED (more workgroups than slots: they start when an earlier gene finishes, tens of
        // microseconds later) must not keep hammering the fabric meanwhile -- agent-scope loads bypass the caches, and a few
        // hundred waves polling back to back slowed the running genes down to the point of a time-out (16 genes of 200 x 5000:
        // 25 of a gene's 39 tiles resident and spinning next to a fully staffed gene)
        ++polls;
        if (polls < 16u) __builtin_amdgcn_s_sleep(1);
        else if (polls < 32u) __builtin_amdgcn_s_sleep(8);
        else if (polls < 64u) __builtin_amdgcn_s_sleep(32);
        else __builtin_amdgcn_s_sleep(127);
        // bounded in wall-clock time; a slice that gives up takes the whole launch (and the stream's later launches) with it
        // through the abort word, so the device drains instead of spinning bound after bound
        if ((polls & 15u) == 0u || timeout_ticks == 0) {
            if (__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { bad = true; break; }
            if (wall_clock64() - t_start > timeout_ticks) {
                const unsigned long long have = __ballot(ok);
                if (lane == 0 && atomicCAS(&ctl->dbg[0], 0, 1) == 0) {
                    ctl->dbg[1] = wg; ctl->dbg[2] = S; ctl->dbg[3] = nevals; ctl->dbg[4] = (int)(have & 0xFFFFFFFFull); ctl->dbg[5] = (int)(have >> 32);
                    ctl->dbg[6] = __hip_atomic_load(&ctl->oticket[blockIdx.x & 7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ctl->dbg[7] = __hip_atomic_load(&ctl->odone, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * 1024 + (int)(blockIdx.x & 7);
                }
                __hip_atomic_store(&ctl->abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bad = true; break;
            }
        }
    }
#ifdef PML_OPTIME        // [246] all gathers, [247] the first gather of a request
    if (lane == 0) { atomicAdd(&g_optime[246][0], (unsigned long long)(clock64() - t_poll0)); atomicAdd(&g_optime[246][1], 1ull); if (nevals == 0) atomicAdd(&g_optime[247][0], (unsigned long long)(clock64() - t_poll0)); }
#endif
#pragma unroll
    for (int i = 0; i < 3; ++i) {   // lanes >= S contribute +0.0 (exact); fixed shuffle tree: a function of S alone
        const u64 bits = ((x[2 * i + 1] & 0xFFFFFFFFull) << 32) | (x[2 * i] & 0xFFFFFFFFull);
        tot[i] = wave_sum(lane < S ? __longlong_as_double((long long)bits) : 0.0);
    }
    return bad;
}

// Newton-Raphson with step control: the oracle's eng_newton_branch().  eval_at(t, L, d1, d2) -> false when the exchange gave up.
// Wave-uniform; every wave of every slice runs it on the same broadcast sums.
template <class Eval>
__device__ __forceinline__ bool newton_drive(double t0, int max_iter, double tol, Eval &&eval_at, double &t, double &L, double &d1, double &d2) {
    t = t0;
    if (max_iter > 0) t = t < PML_TMIN ? PML_TMIN : (t > PML_TMAX ? PML_TMAX : t);
    t = uni(t);
    L = 0.0; d1 = 0.0; d2 = 0.0;
    double tn = t;
    bool first = true, failed = false;
    int it = 0, bt = 0;
    for (;;) {                                          // one evaluation site: initial point, Newton steps and backtracks
        double Ln, n1, n2;
        if (!eval_at(tn, Ln, n1, n2)) failed = true;
        if (first) { first = false; L = Ln; d1 = n1; d2 = n2; }
        else {
            if (!(failed || Ln >= L - 1e-9 || bt >= 8)) { ++bt; tn = 0.5 * (tn + t); tn = uni(tn < PML_TMIN ? PML_TMIN : (tn > PML_TMAX ? PML_TMAX : tn)); continue; }
            if (failed || Ln < L - 1e-9) break;
            const double dt = fabs(tn - t);
            t = tn; L = Ln; d1 = n1; d2 = n2; ++it;
            if (dt < tol) break;
        }
        if (it >= max_iter || failed) break;
        const double step = (d2 < 0.0) ? -d1 / d2 : (d1 > 0.0 ? t : -0.5 * t);
        tn = t + step; bt = 0;
        const bool tiny = fabs(step) < tol && d2 < 0.0;   // converged: take the (sub-tolerance) step unevaluated
        tn = uni(tn < PML_TMIN ? PML_TMIN : (tn > PML_TMAX ? PML_TMAX : tn));
        if (tiny) { t = tn; break; }
    }
    return !failed;
}

// ------------------------------------------------------------------------------------------
// CLV op on one chunk (32 patterns) of one wave.
//   MODE_NEWVIEW : out[c][s] = (P_L,c . L_c)[s] * (P_R,c . R_c)[s], 2^256 rescue, scaling counts
//   MODE_SUMTABLE: same contraction with the eigen-basis matrices (no rescue), counts = l + r
//   MODE_EVALUATE: per-pattern ln( 1/4 sum_c sum_s L_c[s] (pi P_c . R_c)[s] ) - counts*256 ln 2
// ------------------------------------------------------------------------------------------
typedef double dvec2 __attribute__((ext_vector_type(2)));   // native vectors: loadable from address_space(1)
typedef int ivec2 __attribute__((ext_vector_type(2)));
typedef unsigned uvec4 __attribute__((ext_vector_type(4)));
struct Operand {            // one child's 5 two-pattern B operands of one category
    dvec2 v[5];
};

typedef const __attribute__((address_space(1))) char *gcptr;    // explicit GLOBAL pointers: pointers read
typedef __attribute__((address_space(1))) char *gptr;           // from a descriptor are generic -> flat_load
#define GLOBAL_AS __attribute__((address_space(1)))

// CLV operand: base is wave-uniform (SGPR pair), lane_off the lane's 32-bit byte offset
// (q*M + p)*8, so the loads compile to the saddr + voffset form without per-row VGPR addresses.
// CLVs are streamed: every byte is read once per launch, so the loads carry the non-temporal hint (measured, same box,
// rotated order: C3 scoring launch 0.93-1.00 ms plain, 0.92-0.98 nt loads, 0.85-0.86 nt loads + nt stores; C4 shard 10.1 /
// 9.3 / 9.9 ms -- profiles/r02_ab_nontemporal.txt).
__device__ __forceinline__ void load_clv(Operand &o, gcptr base, unsigned lane_off, size_t rowbytes, int c) {
#ifdef ABL_NO_CLV
    { for (int kk = 0; kk < 5; ++kk) o.v[kk] = (dvec2){0.05 + 1e-12 * lane_off, 0.05}; return; }
#endif
#pragma unroll
    for (int kk = 0; kk < 5; ++kk)
        o.v[kk] = __builtin_nontemporal_load(reinterpret_cast<const GLOBAL_AS dvec2 *>(base + (size_t)(c * NS + kk * 4) * rowbytes + lane_off));
}
// tip operand: 0/1 indicator rows from the LDS table T[code][state] (same for every category)
__device__ __forceinline__ void load_tip(Operand &o, const unsigned char *__restrict__ T, unsigned codes, int q) {
    const unsigned char *t0 = T + (codes & 0xFFu) * NS + q, *t1 = T + (codes >> 8) * NS + q;
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) o.v[kk] = (dvec2){(double)t0[kk * 4], (double)t1[kk * 4]};
}

// acc[st][0/1] = sum_kk frag(c,st,kk) x operand(kk)   (even / odd pattern of the lane).
// The A fragments are software-pipelined one k-step ahead and fenced with sched_barrier so that
// hipcc does not hoist all 25 ds_reads (50 VGPRs) in front of the MFMAs.
#ifndef PML_FRAG_AHEAD
#define PML_FRAG_AHEAD 1
#endif
__device__ __forceinline__ void contract(double (&acc)[5][2], const double *__restrict__ frag_c, const Operand &o) {
    // the A fragments are requested PML_FRAG_AHEAD k-steps ahead.  Timing-only ablations (profiles/r03_ablation_k_oplist.txt: C3
    // launch 0.677 ms; fragment reads made free 0.545, tip-table rows free 0.553, CLV loads free 0.645, no per-op barrier 0.687)
    // say the LDS reads cost 20 % -- but two steps ahead (10 more VGPRs, 3 spills) measures 0.677 vs 0.673: it is not the
    // latency of one read that is exposed
#ifdef ABL_NO_LDS
// a DIFFERENT constant per fragment element: with one constant for all of them the five state-tile chains of a contraction are
// identical and the compiler merges them -- four fifths of the MFMAs disappear (the "LDS reads cost 20 %" of the first
// ablation was that, not the reads)
#define FRAG_RD(x) (0.05 + 1.0e-4 * (double)(&(x) - frag_c))
#else
#define FRAG_RD(x) (x)
#endif
    double a[3][5];
#pragma unroll
    for (int st = 0; st < 5; ++st) { acc[st][0] = 0.0; acc[st][1] = 0.0; a[0][st] = FRAG_RD(frag_c[(st * 5) * 16]); }
    if (PML_FRAG_AHEAD >= 2) {
#pragma unroll
        for (int st = 0; st < 5; ++st) a[1][st] = FRAG_RD(frag_c[(st * 5 + 1) * 16]);
    }
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) {
        if (kk + PML_FRAG_AHEAD < 5) {
#pragma unroll
            for (int st = 0; st < 5; ++st) a[(kk + PML_FRAG_AHEAD) % 3][st] = FRAG_RD(frag_c[(st * 5 + kk + PML_FRAG_AHEAD) * 16]);
        }
#pragma unroll
        for (int st = 0; st < 5; ++st) {
            acc[st][0] = mfma4(a[kk % 3][st], o.v[kk].x, acc[st][0]);
            acc[st][1] = mfma4(a[kk % 3][st], o.v[kk].y, acc[st][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// st-outer form for the SECOND side: produces one state tile (2 values) at a time so the caller can
// consume it immediately (multiply with the first side's tile and store): 4 live accumulator
// registers instead of 40.  Fragments of the next tiles are fetched while the current one runs.
#ifndef PML_CONSUME_LATE
#define PML_CONSUME_LATE 0
#endif
template <typename F>
__device__ __forceinline__ void contract_stream(const double *__restrict__ frag_c, const Operand &o, F &&consume) {
    double a[3][5];
    double p0 = 0.0, p1 = 0.0;
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) a[0][kk] = FRAG_RD(frag_c[kk * 16]);
    if (PML_FRAG_AHEAD >= 2) {
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) a[1][kk] = FRAG_RD(frag_c[(5 + kk) * 16]);
    }
#pragma unroll
    for (int st = 0; st < 5; ++st) {
        if (st + PML_FRAG_AHEAD < 5) {
#pragma unroll
            for (int kk = 0; kk < 5; ++kk) a[(st + PML_FRAG_AHEAD) % 3][kk] = FRAG_RD(frag_c[((st + PML_FRAG_AHEAD) * 5 + kk) * 16]);
        }
        double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) {
            acc0 = mfma4(a[st % 3][kk], o.v[kk].x, acc0);
            acc1 = mfma4(a[st % 3][kk], o.v[kk].y, acc1);
#if PML_CONSUME_LATE
            // the previous state tile is consumed (multiply, maximum, store) in the shadow of this tile's MFMAs instead of
            // behind its own, where the wave would wait for the matrix pipe to drain and then issue VALU work with the pipe idle
            if (kk == 0 && st > 0) consume(st - 1, p0, p1);
#endif
        }
#if PML_CONSUME_LATE
        p0 = acc0; p1 = acc1;
#else
        consume(st, acc0, acc1);
#endif
        __builtin_amdgcn_sched_barrier(0);
    }
#if PML_CONSUME_LATE
    consume(4, p0, p1);
#endif
}

// cherry operand: product of the two tips' table rows (tables live in global memory, L2-resident:
// every workgroup of the gene reads the same 2 x 17.7 KB); a lane's five rows are 40 contiguous bytes
struct Rows5 { dvec2 a, b; double c; };
// (ABL_* macros: timing-only ablation builds of tools/ab_ablation.sh -- a source of stalls is replaced by constants to see what it
// costs; results are NOT likelihoods.  Never defined in the product build.)
__device__ __forceinline__ Rows5 load_rows(const double *tab, unsigned code, int c, int q) {
#ifdef ABL_NO_ROWS
    { Rows5 r1; r1.a = (dvec2){0.9, 0.9}; r1.b = (dvec2){0.9, 0.9}; r1.c = 0.9 + 1e-9 * code; return r1; }
#endif
    gcptr p = (gcptr)tab + (size_t)((c * NCODES + code) * 4 + q) * (TIPTAB_KK * 8);
    Rows5 r;
    r.a = *reinterpret_cast<const GLOBAL_AS dvec2 *>(p);
    r.b = *reinterpret_cast<const GLOBAL_AS dvec2 *>(p + 16);
    r.c = *reinterpret_cast<const GLOBAL_AS double *>(p + 32);
    return r;
}
__device__ __forceinline__ void load_cherry(Operand &o, const OpSide &sd, unsigned ca, unsigned cb, int c, int q) {
    const Rows5 a0 = load_rows(sd.t0, ca & 0xFFu, c, q), a1 = load_rows(sd.t0, ca >> 8, c, q);
    const Rows5 b0 = load_rows(sd.t1, cb & 0xFFu, c, q), b1 = load_rows(sd.t1, cb >> 8, c, q);
    o.v[0] = (dvec2){a0.a.x * b0.a.x, a1.a.x * b1.a.x};
    o.v[1] = (dvec2){a0.a.y * b0.a.y, a1.a.y * b1.a.y};
    o.v[2] = (dvec2){a0.b.x * b0.b.x, a1.b.x * b1.b.x};
    o.v[3] = (dvec2){a0.b.y * b0.b.y, a1.b.y * b1.b.y};
    o.v[4] = (dvec2){a0.c * b0.c, a1.c * b1.c};
}
// the same in two halves, so that the rows of the NEXT category can be in flight during the current category's second contraction
// (40 VGPRs of raw rows: affordable since the result tile is written straight into X, round 3)
struct CherryRaw { Rows5 a0, a1, b0, b1; };
__device__ __forceinline__ void issue_cherry(CherryRaw &r, const OpSide &sd, unsigned ca, unsigned cb, int c, int q) {
    r.a0 = load_rows(sd.t0, ca & 0xFFu, c, q); r.a1 = load_rows(sd.t0, ca >> 8, c, q);
    r.b0 = load_rows(sd.t1, cb & 0xFFu, c, q); r.b1 = load_rows(sd.t1, cb >> 8, c, q);
}
__device__ __forceinline__ void finish_cherry(Operand &o, const CherryRaw &r) {
    o.v[0] = (dvec2){r.a0.a.x * r.b0.a.x, r.a1.a.x * r.b1.a.x};
    o.v[1] = (dvec2){r.a0.a.y * r.b0.a.y, r.a1.a.y * r.b1.a.y};
    o.v[2] = (dvec2){r.a0.b.x * r.b0.b.x, r.a1.b.x * r.b1.b.x};
    o.v[3] = (dvec2){r.a0.b.y * r.b0.b.y, r.a1.b.y * r.b1.b.y};
    o.v[4] = (dvec2){r.a0.c * r.b0.c, r.a1.c * r.b1.c};
}
// pitchfork operand for category c: ((F_inner . (T_a * T_b)) * T_c), all in registers
template <bool EARLY = true>
__device__ __forceinline__ void load_pitch(Operand &o, const OpSide &sd, const double *__restrict__ f_inner,
                                           unsigned ca, unsigned cb, unsigned cc, int c, int q) {
    Operand w;
    load_cherry(w, sd, ca, cb, c, q);
    // EARLY: the third tip's rows are requested BEFORE the inner contraction (whose scheduling fences keep everything behind it
    // where the source puts it): behind it, as first written, their L2 round trip was waited for in full, once per category
    // (C3 launch 0.570 -> 0.563 ms).  Not in the fused-Newton variant: 20 more live VGPRs there are 20 more spills (29 -> 49).
    Rows5 r0, r1;
    if (EARLY) { r0 = load_rows(sd.t2, cc & 0xFFu, c, q); r1 = load_rows(sd.t2, cc >> 8, c, q); }
    double v[5][2];
    contract(v, f_inner + c * 25 * 16, w);
    if (!EARLY) { r0 = load_rows(sd.t2, cc & 0xFFu, c, q); r1 = load_rows(sd.t2, cc >> 8, c, q); }
    o.v[0] = (dvec2){v[0][0] * r0.a.x, v[0][1] * r1.a.x};
    o.v[1] = (dvec2){v[1][0] * r0.a.y, v[1][1] * r1.a.y};
    o.v[2] = (dvec2){v[2][0] * r0.b.x, v[2][1] * r1.b.x};
    o.v[3] = (dvec2){v[3][0] * r0.b.y, v[3][1] * r1.b.y};
    o.v[4] = (dvec2){v[4][0] * r0.c, v[4][1] * r1.c};
}

// the chained variants unroll the category loop (X[c] with a static index instead of rotating 80 registers per category:
// C3 scoring launch 0.725 -> 0.693 ms, profiles/r02_ab_chain_unroll.txt); -DPML_CHAIN_UNROLL=0 is the A-B arm
#ifndef PML_CHAIN_UNROLL
#define PML_CHAIN_UNROLL 1
#endif
// One op on one chunk (32 patterns) of one wave.  All branches on op.* are wave-uniform.
//   MODE_NEWVIEW : out[c][s] = (P_L,c . L_c)[s] * (P_R,c . R_c)[s], 2^256 rescue, scaling counts
//   MODE_SUMTABLE: same contraction with the eigen-basis matrices (no rescue), counts = l + r
//   MODE_EVALUATE: per-pattern ln( 1/4 sum_c sum_s L_c[s] (pi P_c . R_c)[s] ) - counts*256 ln 2
#ifndef PML_TIPLOOK
#define PML_TIPLOOK 1
#endif
template <bool PREFETCH, bool CHAIN, bool FUSE>
__device__ __forceinline__ void chunk_op(const NvOp &op, const double *__restrict__ sP, const unsigned char *__restrict__ sT,
                                         int p, int lane, Operand (&X)[4], ivec2 &xsc) {
    // `op` refers to the descriptor in global memory (wave-uniform): fields are fetched by scalar loads
    // where they are used instead of being held in ~34 SGPRs for the whole op
    const int q = lane >> 4;
    // tiled CLV layout (kernels.h): the chunk's tile starts (p >> 7) * 80 KB into the CLV, rows are 1 KB apart
    constexpr size_t rowbytes = (size_t)TILE_PAT * 8;
    const size_t tabrow = (size_t)op.mpad * 8;                 // row pitch of the per-category likelihood table (MODE_EVALUATE_CAT)
    const unsigned lane_off = (unsigned)(p >> 7) * (unsigned)(CLV_ROWS * rowbytes) + (unsigned)((size_t)q * rowbytes) + (unsigned)(p & (TILE_PAT - 1)) * 8u;
    const int lk = op.flags & 3, rk = (op.flags >> 2) & 3;
    const bool nt_store = (op.flags & OPF_NT_STORE) != 0;      // wave-uniform: the result is not read again soon (host's call)
    const int mode = op.mode;
    // register chaining (kernels.h OPF_CHAIN_*): X[0..3] = the four categories of the wave's last newview result, xsc its counts
    const bool chL = CHAIN && (op.flags & OPF_CHAIN_L) != 0, chR = CHAIN && (op.flags & OPF_CHAIN_R) != 0;
    // fused branch Newton: the sumtable tile is not stored, it stays in X for newton_fused (its counts in xsc)
    const bool fusedN = FUSE && (op.flags & OPF_FUSED_NEWTON) != 0;
    const bool keep = !(CHAIN && (op.flags & OPF_NO_STORE) != 0) && !fusedN;
    constexpr bool TIPLOOK = PML_TIPLOOK != 0;
    constexpr bool PF_L = PREFETCH && !CHAIN;       // the chained variants prefetch the right side only (the chained child is the left one; registers)
    // (a plain tip side goes through the MFMA with its 0/1 indicator operand: the matrix pipe has slack and
    // table gathers for it measured slower)
    // The descriptor fields the category loop needs are read ONCE, here, and pinned in SGPRs: a scalar load in the middle of a
    // contraction costs far more than its own latency -- SMEM returns out of order, so the compiler has to wait for it with
    // s_waitcnt lgkmcnt(0), which also drains every LDS fragment read that was requested ahead (78 such drains per operation
    // before; the fragment reads showed up as 20 % of the launch in the ablations, profiles/r03_ablation_k_oplist.txt)
#ifndef PML_PIN_DESC
#define PML_PIN_DESC 1
#endif
    OpSide sdl = op.l, sdr = op.r;
    const int *l_scl = op.l_scl, *r_scl = op.r_scl; int *out_scl = op.out_scl; double *outp = op.out;
#if PML_PIN_DESC
#define PML_PIN(x) asm volatile("" : "+s"(x))
    PML_PIN(sdl.p0); PML_PIN(sdl.p1); PML_PIN(sdl.p2); PML_PIN(sdl.t0); PML_PIN(sdl.t1); PML_PIN(sdl.t2); PML_PIN(sdl.f);
    PML_PIN(sdr.p0); PML_PIN(sdr.p1); PML_PIN(sdr.p2); PML_PIN(sdr.t0); PML_PIN(sdr.t1); PML_PIN(sdr.t2); PML_PIN(sdr.f);
    PML_PIN(l_scl); PML_PIN(r_scl); PML_PIN(out_scl); PML_PIN(outp);
#endif
    gcptr Lp = (gcptr)sdl.p0, Rp = (gcptr)sdr.p0;
    gptr O = (gptr)outp;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void *)outp, 0, 0x7FFFFFFF, 0x00020000);   // raw buffer over the output CLV
    // lane's A-fragment element: 4*k + i with k = q, i = lane&3
    const double *fL = sP + (q * 4 + (lane & 3));
    const double *fR = fL + PFRAG;
    // running maxima of the result entries of the lane's two patterns, kept as the HIGH WORDS of the doubles: the entries are
    // non-negative, so their order is the order of their bit patterns, and x < 2^-256 <=> hi(x) < hi(2^-256) because the low word
    // of 2^-256 is zero -- one integer max per entry instead of a double-precision one (a wave issues one of those per 12 cycles)
    unsigned mx0 = 0u, mx1 = 0u;
    double site0 = 0.0, site1 = 0.0;
    unsigned cl = 0, cl2 = 0, cl3 = 0, cr = 0, cr2 = 0, cr3 = 0;      // tip codes of the lane's two patterns
    // inner fragments of a pitchfork side: ONE extra LDS region, owned by the left side if it is a
    // pitchfork, else by the right; if both are, the right side reads its set from global memory (rare)
    const double *fLi = fL + 2 * PFRAG;
    // (two code paths, not one pointer chosen at run time: a pointer that may be global OR shared is a generic one, and every
    // fragment read through it a FLAT load, which counts on both vmcnt and lgkmcnt -- each k-step of the inner contraction then
    // waited for everything in flight: a right-hand pitchfork cost 11 k cycles more than a right-hand cherry for 100 more MFMAs)
    const bool both_pitch = lk == SK_PITCH && rk == SK_PITCH;
    const double *fRg = sdr.f + (q * 4 + (lane & 3));
#ifdef ABL_NO_CODES      // timing-only ablation: tip codes made up from the pattern index instead of loaded
#define CODE_LD(ptr) ((((unsigned)p * 5u + 1u + (unsigned)(size_t)(ptr)) % 20u) | ((((unsigned)p * 3u + 2u) % 20u) << 8))
#else
#define CODE_LD(ptr) (*reinterpret_cast<const GLOBAL_AS unsigned short *>(ptr))
#endif
    if (lk != SK_CLV) cl = CODE_LD(Lp + p);
    if (lk >= SK_CHERRY) cl2 = CODE_LD((gcptr)sdl.p1 + p);
    if (lk == SK_PITCH) cl3 = CODE_LD((gcptr)sdl.p2 + p);
    if (rk != SK_CLV) cr = CODE_LD(Rp + p);
    if (rk >= SK_CHERRY) cr2 = CODE_LD((gcptr)sdr.p1 + p);
    if (rk == SK_PITCH) cr3 = CODE_LD((gcptr)sdr.p2 + p);

    // A tip side with plain amino-acid codes needs no contraction: (P e_a)[s] = P[s][a] is an element of the fragment set
    // already in LDS (A-fragment order: P[4 st + i][4 kk + k] at (st*5 + kk)*16 + k*4 + i) -- 5 LDS reads per pattern and
    // category instead of 25 reads + 50 MFMAs, the same bits (the MFMA adds exact zeros).  Ambiguity codes and gaps (sums of
    // columns) keep the MFMA path for the whole wave, so a pattern's bits do not depend on which path its wave took.
    const bool lookL = TIPLOOK && mode < MODE_EVALUATE && lk == SK_TIP && !__any((cl & 0xFFu) >= 20u || (cl >> 8) >= 20u);
    const bool lookR = TIPLOOK && mode < MODE_EVALUATE && rk == SK_TIP && !__any((cr & 0xFFu) >= 20u || (cr >> 8) >= 20u);
    const double *tL0 = sP + ((cl & 0xFFu) >> 2) * 16 + (cl & 3u) * 4 + q, *tL1 = sP + ((cl >> 8) >> 2) * 16 + ((cl >> 8) & 3u) * 4 + q;
    const double *tR0 = sP + PFRAG + ((cr & 0xFFu) >> 2) * 16 + (cr & 3u) * 4 + q, *tR1 = sP + PFRAG + ((cr >> 8) >> 2) * 16 + ((cr >> 8) & 3u) * 4 + q;
    Operand curL, curR, nxtL, nxtR;
    CherryRaw rawR;
    if (lk == SK_TIP && !lookL) load_tip(curL, sT, cl, q);
    else if (lk == SK_CLV && !chL) load_clv(curL, Lp, lane_off, rowbytes, 0);      // evaluate: left side in output layout
    if (rk == SK_TIP && !lookR) load_tip(curR, sT, cr, q);
    else if (rk == SK_CLV && !chR) load_clv(curR, Rp, lane_off, rowbytes, 0);
    constexpr int CAT_UNROLL = (CHAIN && PML_CHAIN_UNROLL) ? NCAT : 1;
#pragma unroll CAT_UNROLL
    for (int c = 0; c < NCAT; ++c) {
        Operand Y;                                            // CHAIN: this category of the result
        // (fully unrolled: a chained side is contracted straight out of X[c] -- its own code path below -- instead of being copied
        // into the operand registers first: 10 v_mov_b64 per category and side, at the 12 cycles a wave pays per double-precision
        // VALU instruction, tools/ubench_f64.hip)
        constexpr bool DIRECT = CHAIN && CAT_UNROLL == NCAT;        // results are written straight into X[c]
#ifndef PML_DIRECT_FUSE
#define PML_DIRECT_FUSE 0
#endif
        constexpr bool DIRECT_IN = DIRECT && (!FUSE || PML_DIRECT_FUSE);    // chained operands are read straight from X[c]
        if (CHAIN && !DIRECT_IN) {
            if (CAT_UNROLL == NCAT) { if (chL) curL = X[c]; if (chR) curR = X[c]; }
            else { if (chL) curL = X[0]; if (chR) curR = X[0]; }     // X is rotated once per category: X[0] is category c
        }
        if (PREFETCH && c + 1 < NCAT) {                       // software prefetch of the next category
            if (PF_L && lk == SK_CLV && !chL) load_clv(nxtL, Lp, lane_off, rowbytes, c + 1);
            if (rk == SK_CLV && !chR) load_clv(nxtR, Rp, lane_off, rowbytes, c + 1);
        }
        // PML_PF_ROWS (A-B arms, both measured slower, profiles/r03_kernel_steps.txt): 1 = a right-hand cherry's rows requested one
        // category ahead (40 VGPRs of raw rows live through the second contraction: 123 spills), 2 = requested in front of the
        // left contraction of the same category and multiplied behind it (63 spills, 0.60 -> 0.72 ms)
#ifndef PML_PF_ROWS
#define PML_PF_ROWS 0
#endif
        constexpr bool PF_ROWS = PML_PF_ROWS && DIRECT && !FUSE;
        const bool pfR = PF_ROWS && rk == SK_CHERRY && lk == SK_CLV && mode < MODE_EVALUATE;     // right-hand cherry next to a CLV: rows requested early
        // (first thing in the category: the 40 raw-row registers die here, before the left side builds its own operand)
#if PML_PF_ROWS == 1
        if (pfR) { if (c == 0) issue_cherry(rawR, sdr, cr, cr2, 0, q); finish_cherry(curR, rawR); }
#else
        // PML_PF_ROWS == 2: the rows of THIS category are requested here and multiplied behind the left contraction, whose 50 MFMAs
        // cover most of their L2 round trip; nothing stays live across categories
        if (pfR) issue_cherry(rawR, sdr, cr, cr2, c, q);
#endif
        if (lk == SK_CHERRY) load_cherry(curL, sdl, cl, cl2, c, q);
        else if (lk == SK_PITCH) load_pitch<!FUSE>(curL, sdl, fLi, cl, cl2, cl3, c, q);
        if (pfR) {}
        else if (rk == SK_CHERRY) load_cherry(curR, sdr, cr, cr2, c, q);
        else if (rk == SK_PITCH) { if (both_pitch) load_pitch<!FUSE>(curR, sdr, fRg, cr, cr2, cr3, c, q); else load_pitch<!FUSE>(curR, sdr, fLi, cr, cr2, cr3, c, q); }
        if (mode >= MODE_EVALUATE) {
            if (DIRECT_IN && chL) contract_stream(fR + c * 25 * 16, curR, [&](int st, double y0, double y1) { site0 += X[c].v[st].x * y0; site1 += X[c].v[st].y * y1; });
            else if (DIRECT_IN && chR) contract_stream(fR + c * 25 * 16, X[c], [&](int st, double y0, double y1) { site0 += curL.v[st].x * y0; site1 += curL.v[st].y * y1; });
            else contract_stream(fR + c * 25 * 16, curR, [&](int st, double y0, double y1) {
                site0 += curL.v[st].x * y0; site1 += curL.v[st].y * y1;
            });
            if (mode == MODE_EVALUATE_CAT) {          // this category's likelihood goes out on its own (row c of the table slice)
                double a0 = site0, a1 = site1;
                a0 += __shfl_xor(a0, 16); a0 += __shfl_xor(a0, 32);
                a1 += __shfl_xor(a1, 16); a1 += __shfl_xor(a1, 32);
                if (q == 0) *reinterpret_cast<GLOBAL_AS dvec2 *>(O + (size_t)c * tabrow + 8 * p) = (dvec2){a0, a1};
                site0 = 0.0; site1 = 0.0;
            }
        } else {
            double aL[5][2];
            if (lookL) {
#pragma unroll
                for (int st = 0; st < 5; ++st) { aL[st][0] = tL0[c * 400 + st * 80]; aL[st][1] = tL1[c * 400 + st * 80]; }
            } else if (DIRECT_IN && chL) contract(aL, fL + c * 25 * 16, X[c]);
            else contract(aL, fL + c * 25 * 16, curL);
#if PML_PF_ROWS == 1
            if (pfR && c + 1 < NCAT) issue_cherry(rawR, sdr, cr, cr2, c + 1, q);
#else
            if (pfR) finish_cherry(curR, rawR);
#endif
            auto emit = [&](int st, double y0, double y1) {
                const double o0 = aL[st][0] * y0, o1 = aL[st][1] * y1;
                mx0 = max(mx0, (unsigned)__double2hiint(o0)); mx1 = max(mx1, (unsigned)__double2hiint(o1));
                // fully unrolled: the result tile goes straight into X[c] (the left contraction of this category, the only reader
                // of the old X[c], is complete) -- for sumtable operations too, which therefore END a chain (engine.cpp: nothing is
                // chained from across a sumtable tail)
                if (CHAIN) { if (DIRECT) X[c].v[st] = (dvec2){o0, o1}; else Y.v[st] = (dvec2){o0, o1}; }
                if (!keep) return;
#ifdef ABL_NO_STORE
                return;
#endif
                // buffer stores: the cache policy is an immediate of the instruction, so the two policies are two instructions
                // under a wave-uniform branch (an if/else of a plain and a __builtin_nontemporal_store is merged by hipcc
                // into ONE plain store)
                const uvec4 bits = __builtin_bit_cast(uvec4, ((dvec2){o0, o1}));
                const int soff = (c * NS + st * 4) * (int)rowbytes;
                if (nt_store) __builtin_amdgcn_raw_buffer_store_b128(bits, orsrc, lane_off, soff, 2);      // aux 2 = nt
                else __builtin_amdgcn_raw_buffer_store_b128(bits, orsrc, lane_off, soff, 0);
            };
            if (lookR) {
#pragma unroll
                for (int st = 0; st < 5; ++st) emit(st, tR0[c * 400 + st * 80], tR1[c * 400 + st * 80]);
            } else contract_stream(fR + c * 25 * 16, curR, emit);
        }
        if (CHAIN && !DIRECT) {
            if (mode == MODE_NEWVIEW) { X[0] = X[1]; X[1] = X[2]; X[2] = X[3]; X[3] = Y; }
            else if (chL || chR) { const Operand t = X[0]; X[0] = X[1]; X[1] = X[2]; X[2] = X[3]; X[3] = t; }   // a tail leaves X as it was
        }
        if (c + 1 < NCAT) {
            if (PREFETCH) { if (PF_L && lk == SK_CLV && !chL) curL = nxtL; if (rk == SK_CLV && !chR) curR = nxtR; }
            if (!PF_L) { if (lk == SK_CLV && !chL) load_clv(curL, Lp, lane_off, rowbytes, c + 1); }
            if (!PREFETCH) {
                if (rk == SK_CLV && !chR) load_clv(curR, Rp, lane_off, rowbytes, c + 1);
            }
        }
    }

    ivec2 sc = {0, 0};
    if (q == 0) {
        if (lk == SK_CLV) { if (chL) sc += xsc; else sc += *reinterpret_cast<const GLOBAL_AS ivec2 *>((gcptr)l_scl + 4 * p); }
        if (rk == SK_CLV) { if (chR) sc += xsc; else sc += *reinterpret_cast<const GLOBAL_AS ivec2 *>((gcptr)r_scl + 4 * p); }
    }
    if (mode == MODE_NEWVIEW) {
        mx0 = max(mx0, (unsigned)__shfl_xor((int)mx0, 16)); mx0 = max(mx0, (unsigned)__shfl_xor((int)mx0, 32));
        mx1 = max(mx1, (unsigned)__shfl_xor((int)mx1, 16)); mx1 = max(mx1, (unsigned)__shfl_xor((int)mx1, 32));
        constexpr unsigned HI_TWO_M256 = (1023u - 256u) << 20;
        const bool n0 = mx0 < HI_TWO_M256, n1 = mx1 < HI_TWO_M256;
        if (__any(n0 || n1)) {               // rare: numerical rescue of underflowing patterns
            if (CHAIN) {
                const double f0 = n0 ? TWO_P256 : 1.0, f1 = n1 ? TWO_P256 : 1.0;
#pragma unroll
                for (int c = 0; c < NCAT; ++c)
#pragma unroll
                    for (int k = 0; k < 5; ++k) { X[c].v[k].x *= f0; X[c].v[k].y *= f1; }
            }
            if (keep && (n0 || n1)) {
#pragma unroll 1
                for (int r = 0; r < NCAT * 5; ++r) {
                    GLOBAL_AS dvec2 *ptr = reinterpret_cast<GLOBAL_AS dvec2 *>(O + (size_t)((r / 5) * NS + (r % 5) * 4) * rowbytes + lane_off);
                    dvec2 v = *ptr;
                    if (n0) v.x *= TWO_P256;
                    if (n1) v.y *= TWO_P256;
                    *ptr = v;
                }
            }
        }
        if (q == 0) { sc.x += n0 ? 1 : 0; sc.y += n1 ? 1 : 0; if (keep) *reinterpret_cast<GLOBAL_AS ivec2 *>((gptr)out_scl + 4 * p) = sc; }
        if (CHAIN) xsc = sc;
    } else if (mode == MODE_SUMTABLE || mode == MODE_EVALUATE_CAT) {
        if (fusedN) xsc = sc;
        else if (q == 0) *reinterpret_cast<GLOBAL_AS ivec2 *>((gptr)out_scl + 4 * p) = sc;
    } else {
        site0 += __shfl_xor(site0, 16); site0 += __shfl_xor(site0, 32);
        site1 += __shfl_xor(site1, 16); site1 += __shfl_xor(site1, 32);
        if (q == 0) {
            const double l0 = log(site0 * 0.25) - sc.x * LOG_2_256;
            const double l1 = log(site1 * 0.25) - sc.y * LOG_2_256;
            *reinterpret_cast<GLOBAL_AS dvec2 *>(O + 8 * p) = (dvec2){l0, l1};
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_oplist: workgroup (gene, pattern block of 128) executes the gene's op list in order.
// VARIANT bit0: low-register form (no category prefetch)
//         bit1: double-buffered fragment staging by LDS-DMA (one barrier per op)
// ------------------------------------------------------------------------------------------
constexpr int PAT_PER_WG = 4 * PAT_PER_WAVE;   // 128
constexpr int TIPTAB = NCODES * NS;            // 460 bytes (0/1 indicators)

__device__ __forceinline__ void stage_frags_dma(const NvOp &op, double *dst, int lane, int wave) {
    // 2*PFRAG doubles = 25 KiB = 25 wave-instructions of 1 KiB (16 B per lane)
    for (int i = wave; i < 25; i += 4) {
        const int e = i * 128 + lane * 2;      // element index of this lane's 16 bytes in [left|right]
        const double *g = (e < PFRAG) ? (op.pl ? op.pl : op.pr) + e : (op.pr ? op.pr : op.pl) + (e - PFRAG);
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)g, (__attribute__((address_space(3))) void *)(dst + i * 128), 16, 0, 0);
    }
}

#ifndef PML_CHAIN_WAVES
#define PML_CHAIN_WAVES 2
#endif
// left | right | inner (pitchfork) fragment sets of one op into the three consecutive LDS regions at dst.
// A set is 12.5 rows of 1 KiB (one LDS-DMA instruction of 16 B per lane each); a wave takes one chunk of each set -- rows 4k..4k+3
// through the instruction's immediate offset (it advances the global AND the LDS address), or the half row 12 -- with k rotated from
// set to set so that the waves issue 9 to 12 instructions each.  The descriptor's pointers are read once, every address is a
// wave-uniform base + lane * 16: about three instructions per KiB.  (The first version walked `for (i = wave; i < 25; i += 4)` over
// the left | right pair with a per-lane choice of the source: the compiler re-read the descriptor from memory and waited for it
// INSIDE the loop and spent ~25 instructions per KiB -- seen in the ISA after the single-wave ablations had shown that the kernel
// is bound by what a wave issues, not by memory: C3 launch 0.595 -> 0.572 ms.  Issuing the staging from inside the category loop
// instead of in front of the operation, so that the operation's own first loads are not queued behind it, measured SLOWER: 0.580-0.586.)
__device__ __forceinline__ void dma_chunk(const double *set, double *lds_set, int k, int lane) {
    const GLOBAL_AS char *src = (const GLOBAL_AS char *)set + k * 4096 + lane * 16;
    __attribute__((address_space(3))) char *dst = (__attribute__((address_space(3))) char *)lds_set + k * 4096;
    if (k < 3) {
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)src, (__attribute__((address_space(3))) void *)dst, 16, 1024, 0);
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)src, (__attribute__((address_space(3))) void *)dst, 16, 2048, 0);
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)src, (__attribute__((address_space(3))) void *)dst, 16, 3072, 0);
    } else if (lane < 32) __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
}
__device__ __forceinline__ void stage_frags_dma3(const NvOp &op, double *dst, int lane, int wave) {
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const double *pl = op.pl, *pr = op.pr;
    const int lk = op.flags & 3, rk = (op.flags >> 2) & 3;
    const double *inner = (lk == SK_PITCH) ? op.l.f : (rk == SK_PITCH) ? op.r.f : nullptr;
    dma_chunk(pl ? pl : pr, dst, wv, lane);
    dma_chunk(pr ? pr : pl, dst + PFRAG, (wv + 1) & 3, lane);
    if (inner != nullptr) dma_chunk(inner, dst + 2 * PFRAG, (wv + 2) & 3, lane);
}

// Fused branch Newton (OPF_FUSED_NEWTON): the workgroups of one gene -- one 128-pattern tile of the sumtable each, in X --
// iterate makenewz in place.  Same per-pattern order, finishing lanes, wave sums, exchange and step control as k_newton
// (shared helpers above), hence its bits: waves 0 / 1 are k_newton's service waves A / B (80 exponentials per evaluation by
// threads 0..79, logs and divisions of patterns l / 64 + l, wave 1 the exchange), all four waves add their patterns' rows.
// `scratch` = the current parity's fragment region (the eigen-basis fragments the sumtable operation has finished with).
struct FusedShared {
    double exl[NCAT * NS][2];
    double red[3];
    double fb[3][PAT_PER_WAVE * 4];
    double ss[PAT_PER_WAVE * 4];
    double bc[4];
};
static_assert(sizeof(FusedShared) <= PFRAG * sizeof(double), "fused Newton scratch fits one fragment region");
__device__ __forceinline__ void newton_fused(const NvOp &op, double *scratch, const Operand (&X)[4], const ivec2 xsc,
                                             bool active, int blk, NewtonCtl *ctl, long long timeout_ticks) {
    const NewtonReq &r = *reinterpret_cast<const NewtonReq *>(op.aux);
    FusedShared &sh = *reinterpret_cast<FusedShared *>(scratch);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4, pl = wave * PAT_PER_WAVE + 2 * (lane & 15);
    const int mpad = r.mpad, S = newton_split(mpad);
    const ModelDev *__restrict__ md = r.md;
    u64 *gran = reinterpret_cast<u64 *>(r.sync);
    __syncthreads();                       // every wave has finished the sumtable operation: its fragments may be overwritten
    if (q == 0) { sh.ss[pl] = active ? (double)xsc.x : 0.0; sh.ss[pl + 1] = active ? (double)xsc.y : 0.0; }
    if (tid == 0) sh.bc[3] = 0.0;
    double sw = 0.0;                       // finishing lanes (waves 0, 1): weight of pattern wave * 64 + lane of the tile
    if (wave < 2) { const int pf = blk * (PAT_PER_WAVE * 4) + wave * 64 + lane; if (pf < mpad) sw = r.weight[pf]; }
    int nevals = 0;
    auto eval_at = [&](double t, double &L, double &d1, double &d2) -> bool {
        if (tid < NCAT * NS) {
            const double lr = md->eval[tid % NS] * r.rates[tid / NS];
            sh.exl[tid][0] = exp(lr * t); sh.exl[tid][1] = lr;
        }
        __syncthreads();
        double fa = 0.0, fa1 = 0.0, fa2 = 0.0, fb = 0.0, fb1 = 0.0, fb2 = 0.0;
#pragma unroll
        for (int c = 0; c < NCAT; ++c)
#pragma unroll
            for (int st = 0; st < 5; ++st) {
                const int row = c * NS + 4 * st + q;
                const double e = sh.exl[row][0], lr = sh.exl[row][1];
                newton_term(X[c].v[st].x, e, lr, fa, fa1, fa2);
                newton_term(X[c].v[st].y, e, lr, fb, fb1, fb2);
            }
        // quarters: (q0 + q1) + (q2 + q3), as quad_sum combines them in k_newton
        fa += __shfl_xor(fa, 16); fa += __shfl_xor(fa, 32); fa1 += __shfl_xor(fa1, 16); fa1 += __shfl_xor(fa1, 32); fa2 += __shfl_xor(fa2, 16); fa2 += __shfl_xor(fa2, 32);
        fb += __shfl_xor(fb, 16); fb += __shfl_xor(fb, 32); fb1 += __shfl_xor(fb1, 16); fb1 += __shfl_xor(fb1, 32); fb2 += __shfl_xor(fb2, 16); fb2 += __shfl_xor(fb2, 32);
        if (q == 0) {
            sh.fb[0][pl] = fa; sh.fb[1][pl] = fa1; sh.fb[2][pl] = fa2;
            sh.fb[0][pl + 1] = fb; sh.fb[1][pl + 1] = fb1; sh.fb[2][pl + 1] = fb2;
        }
        __syncthreads();
        double tot[3] = {0.0, 0.0, 0.0};
        if (wave < 2) {
            double a0 = 0.0, a1 = 0.0, a2 = 0.0;
            if (sw != 0.0) {
                const int j = wave * 64 + lane;
                newton_finish(sh.fb[0][j], sh.fb[1][j], sh.fb[2][j], sw, sh.ss[j], a0, a1, a2);
            }
            tot[0] = wave_sum(a0); tot[1] = wave_sum(a1); tot[2] = wave_sum(a2);
            if (wave == 0 && lane == 0) { sh.red[0] = tot[0]; sh.red[1] = tot[1]; sh.red[2] = tot[2]; }
        }
        __syncthreads();
        if (wave == 1) {
#pragma unroll
            for (int i = 0; i < 3; ++i) tot[i] = sh.red[i] + __shfl(tot[i], 0);
            bool bad = false;
            if (S > 1) bad = newton_exchange(gran, S, blk, nevals, r.tag_base, tot, ctl, timeout_ticks);
            if (lane == 0) { sh.bc[0] = tot[0]; sh.bc[1] = tot[1]; sh.bc[2] = tot[2]; if (bad) sh.bc[3] = 1.0; }
        }
        __syncthreads();
        ++nevals;
        L = uni(sh.bc[0]); d1 = uni(sh.bc[1]); d2 = uni(sh.bc[2]);
        return sh.bc[3] == 0.0;
    };
    double t, L, d1, d2;
    const bool ok = newton_drive(r.t0, r.max_iter, r.tol, eval_at, t, L, d1, d2);
    if (tid == 64 && blk == 0) {           // wave 1, lane 0 of the gene's first tile
        atomicAdd(&ctl->n_requests, 1ull); atomicAdd(&ctl->n_evals, (unsigned long long)nevals);
        if (!ok) { t = r.t0; L = __builtin_nan(""); }      // exchange gave up: the host re-issues the step unfused (no-exchange form)
        r.out[0] = t; r.out[1] = L; r.out[2] = d1; r.out[3] = d2;
        if (r.t_dev0) { *r.t_dev0 = t; *r.t_dev1 = t; }
    }
}

#ifdef PML_OPTIME
__device__ unsigned long long g_optime[256][2];
} // namespace pml
extern "C" void pml_abl_optime(unsigned long long *out, int reset) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(pml::g_optime), sizeof(unsigned long long) * 512);
    if (reset) { static unsigned long long z[512]; (void)hipMemcpyToSymbol(HIP_SYMBOL(pml::g_optime), z, sizeof z); }
}
namespace pml {
#endif
template <int VARIANT>
// VARIANT 5 = variant 1 compiled for 3 waves/SIMD (168 VGPRs, no spills)
// VARIANT 9 = variant 1 with register chaining (kernels.h OPF_CHAIN_*): 80 more live VGPRs, 2 waves/SIMD
// VARIANT 11 = 9 + double-buffered fragment staging by LDS-DMA, three regions per parity (2 workgroups per CU leave 80 KB each)
// VARIANT 15 = 11 + fused branch Newton (OPF_FUSED_NEWTON) and ticketed slots: the launches of the search that carry Newton
//              tails; a variant of its own so that the Newton code (exp, log, exchange) costs the scoring kernel no register
__global__ __launch_bounds__(256, (VARIANT == 1) ? 4 : (VARIANT >= 8) ? PML_CHAIN_WAVES : 3) void k_oplist(
        const NvOp *__restrict__ ops, const GeneRun *__restrict__ runs, int nruns, int blocks_per_gene, int any_pitch,
        NewtonCtl *ctl, long long timeout_ticks) {
#ifdef PML_OPTIME
    const long long t_life0 = clock64();
#endif
    constexpr bool PREFETCH = !(VARIANT & 1);
    constexpr bool CHAIN = VARIANT >= 8;                  // 8..11, 15: bit 0 / bit 1 as above, three LDS regions per parity
    constexpr bool FUSE = VARIANT == 15;
    constexpr bool DBUF3 = CHAIN && (VARIANT & 2) != 0;
    constexpr bool DBUF = ((VARIANT & 2) != 0 && VARIANT < 4) || DBUF3;
    constexpr int PARITY_STRIDE = (DBUF3 ? 3 : 2) * PFRAG;
    Operand X[4]; ivec2 xsc = {0, 0};
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < 5; ++k) X[c].v[k] = (dvec2){0.0, 0.0};
    // dynamic LDS: [left|right] fragments (x2 when double-buffered) [+ left-inner|right-inner fragments of
    // pitchfork sides when the launch has any] + the float tip-indicator table
    extern __shared__ double sP[];
    const int nfrag_regions = DBUF3 ? 6 : (DBUF ? 4 : 2) + (any_pitch ? 1 : 0);
    unsigned char *sT = reinterpret_cast<unsigned char *>(sP + nfrag_regions * PFRAG);
    // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share
    // an XCD and its L2), so all pattern blocks of one gene get the same blockIdx % 8: the gene's
    // transition-matrix fragments are then fetched into ONE L2 instead of eight (speed only).
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // blocks_per_gene < 0 (ticketed launches of genes too large for two of them per XCD): ONE partition, genes in plain order
    const bool one_part = blocks_per_gene < 0;
    if (one_part) blocks_per_gene = -blocks_per_gene;
    const int xcd = one_part ? 0 : (blockIdx.x & 7);
    int slot = one_part ? (int)blockIdx.x : (int)(blockIdx.x >> 3);
    // Launches with fused Newton tails (ctl != null): the gene's workgroups exchange sums, i.e. WAIT for each other, so the
    // slot is not tied to blockIdx but claimed by ticket when the workgroup starts (one counter per XCD partition, which keeps
    // a gene's tiles on one XCD): whoever holds a ticket is running, and so are the holders of all lower tickets of the
    // partition -- every gene but the newest is fully staffed (kernels.hip "FORWARD PROGRESS" at k_newton).
    const bool ticketed = FUSE && ctl != nullptr;
    if (ticketed) {
        __shared__ int s_slot;
        if (tid == 0) s_slot = __hip_atomic_fetch_add(&ctl->oticket[xcd], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        slot = s_slot;
    }
    auto leave = [&]() {                   // the workgroup that finishes last re-arms the ticket counters for the stream's next launch
        if (!ticketed) return;
        __syncthreads();
        if (tid == 0) {
            const int d = __hip_atomic_fetch_add(&ctl->odone, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == (int)gridDim.x - 1) {
#pragma unroll
                for (int k = 0; k < 8; ++k) __hip_atomic_store(&ctl->oticket[k], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ctl->odone, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    const int gi = one_part ? slot / blocks_per_gene : xcd + 8 * (slot / blocks_per_gene), blk = slot % blocks_per_gene;
    if (gi >= nruns) { leave(); return; }
    const GeneRun run = runs[gi];
    if (run.op_begin >= run.op_end) { leave(); return; }
    const int mpad = ops[run.op_begin].mpad;            // constant per gene
    if (blk * PAT_PER_WG >= mpad) { leave(); return; }
    const int p = (blk * 4 + wave) * PAT_PER_WAVE + 2 * (lane & 15);
    const bool active = (blk * 4 + wave) * PAT_PER_WAVE < mpad;

#ifdef PML_OPTIME
    const long long t_start1 = clock64();        // slot claimed, run descriptor read
#endif
    for (int i = tid; i < TIPTAB; i += 256) sT[i] = (unsigned char)((code_mask(i / NS) >> (i % NS)) & 1u);
    if (DBUF) { if (DBUF3) stage_frags_dma3(ops[run.op_begin], sP, lane, wave); else stage_frags_dma(ops[run.op_begin], sP, lane, wave); }
    __syncthreads();
#ifdef PML_OPTIME
    const long long t_start2 = clock64();        // tip table filled, first fragment sets staged and landed
    long long t_newton = 0, t_bar = 0, t_ops = 0;
#endif
    for (int oi = run.op_begin; oi < run.op_end; ++oi) {
        const NvOp &op = ops[oi];
        const double *buf = sP;
        if (DBUF) {
            const int par = (oi - run.op_begin) & 1;
            buf = sP + par * PARITY_STRIDE;
            if (oi + 1 < run.op_end) {
                if (DBUF3) stage_frags_dma3(ops[oi + 1], sP + (par ^ 1) * PARITY_STRIDE, lane, wave);
                else stage_frags_dma(ops[oi + 1], sP + (par ^ 1) * PARITY_STRIDE, lane, wave);
            }
        } else {
            if (oi > run.op_begin) __syncthreads();      // previous op: LDS reads and global stores complete
            const double2 *gl = reinterpret_cast<const double2 *>(op.pl);
            const double2 *gr = reinterpret_cast<const double2 *>(op.pr);
            double2 *s2 = reinterpret_cast<double2 *>(sP);
            for (int i = tid; i < PFRAG / 2; i += 256) {
                if (op.mode < MODE_EVALUATE && op.pl) s2[i] = gl[i];
                if (op.pr) s2[PFRAG / 2 + i] = gr[i];
            }
            if (any_pitch) {                 // inner fragment sets of pitchfork sides
                const int lk = op.flags & 3, rk = (op.flags >> 2) & 3;
                const double2 *il = reinterpret_cast<const double2 *>(op.l.f), *ir = reinterpret_cast<const double2 *>(op.r.f);
                for (int i = tid; i < PFRAG / 2; i += 256) {
                    if (lk == SK_PITCH) s2[PFRAG + i] = il[i];
                    else if (rk == SK_PITCH) s2[PFRAG + i] = ir[i];
                }
            }
            __syncthreads();
        }
#ifdef PML_OPTIME         // diagnostic build (tools/optime.sh): shader-clock cycles per operation, by the kinds of its sides
        const long long t_op0 = clock64();
#endif
#ifndef ABL_NO_OP        // timing-only ablation: the shell alone (descriptor reads, fragment staging, barriers)
        if (active) chunk_op<PREFETCH, CHAIN, FUSE>(op, buf, sT, p, lane, X, xsc);
#endif
#ifdef PML_OPTIME
        if (lane == 0 && active) {
            const int k = (op.flags & 15) | ((op.mode & 3) << 4) | ((op.flags & (OPF_CHAIN_L | OPF_CHAIN_R)) ? 64 : 0) | ((op.flags & OPF_NO_STORE) ? 128 : 0);
            atomicAdd(&g_optime[k][0], (unsigned long long)(clock64() - t_op0)); atomicAdd(&g_optime[k][1], 1ull);
        }
        t_ops += clock64() - t_op0;
        const long long t_n0 = clock64();
#endif
        if (FUSE && (op.flags & OPF_FUSED_NEWTON) != 0) {
            // a launch-wide abort (an exchange of this stream gave up earlier) is honoured before waiting on anybody
            if (__hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                const NewtonReq &r = *reinterpret_cast<const NewtonReq *>(op.aux);
                if (tid == 0 && blk == 0) { r.out[0] = r.t0; r.out[1] = __builtin_nan(""); r.out[2] = 0.0; r.out[3] = 0.0; }
            } else newton_fused(op, const_cast<double *>(buf), X, xsc, active, blk, ctl, timeout_ticks);
        }
#ifdef PML_OPTIME
        const long long t_b0 = clock64(); t_newton += t_b0 - t_n0;
#endif
#ifndef ABL_NO_BARRIER
        if (DBUF) __syncthreads();           // next fragments landed (vmcnt(0) + barrier), stores done
#endif
#ifdef PML_OPTIME
        t_bar += clock64() - t_b0;
#endif
    }
#ifdef PML_OPTIME
    if (lane == 0 && active) {
        const int b = FUSE ? 230 : 238;           // [b] lifetime, [b+1] claim + descriptor, [b+2] table + first staging, [b+3] operations, [b+4] Newton, [b+5] barrier
        const long long t_end = clock64();
        atomicAdd(&g_optime[b][0], (unsigned long long)(t_end - t_life0)); atomicAdd(&g_optime[b][1], 1ull);
        atomicAdd(&g_optime[b + 1][0], (unsigned long long)(t_start1 - t_life0)); atomicAdd(&g_optime[b + 2][0], (unsigned long long)(t_start2 - t_start1));
        atomicAdd(&g_optime[b + 3][0], (unsigned long long)t_ops); atomicAdd(&g_optime[b + 4][0], (unsigned long long)t_newton); atomicAdd(&g_optime[b + 5][0], (unsigned long long)t_bar);
    }
#endif
    leave();
}

// ------------------------------------------------------------------------------------------
// k_oplist16 (round 3): the chained op-list kernel with ONE pattern per lane -- a wave owns 16 patterns instead of 32.
// Why: k_oplist<11> needs 256 VGPRs (the chained result alone is 80), i.e. 2 waves per SIMD, and its MFMA bursts are separated
// by L2 / HBM round trips that two waves cannot hide (matrix pipe 34 % busy, waves 47 % in s_waitcnt).  With one pattern per lane
// every per-pattern quantity halves (chained result 40 VGPRs, operands 10, left result 10): the kernel fits 128 VGPRs = 4 waves
// per SIMD with register chaining KEPT.  A workgroup is still one 128-pattern tile, now 8 waves (512 threads) sharing the staged
// fragment sets; two workgroups per CU = the same 256 patterns in flight per CU, as 16 waves instead of 8.  The price: an A
// fragment read from LDS feeds one MFMA instead of two (LDS reads = MFMA count: the LDS pipe is as busy as the matrix pipe).
// Arithmetic per pattern is unchanged (the four blocks of v_mfma_f64_4x4x4_4b are independent 4x4x4 products): same bits.
// RESULT (same box, rotated order, profiles/r03_ab_k_oplist16.txt): SLOWER -- 0.797 ms per C3 scoring launch against 0.676 ms of
// k_oplist<11> (stored traversal 0.93 vs 0.77).  The kernel does not fit its 128 VGPRs (46 spills, in prologue and epilogue), the
// fragment reads cannot be software-pipelined inside that budget (each k-step waits for its five ds_reads: 480 s_waitcnt per
// 500 MFMAs), and twice the LDS reads per MFMA are exposed instead of hidden.  Kept as an A-B arm (PML_CHAIN_VARIANT=16), not used.
// Lane (j = lane & 15, q = lane >> 4): B operand = CLV[state 4 kk + q][pattern j], D = out[state 4 st + q][pattern j].
// ------------------------------------------------------------------------------------------
struct Operand1 { double v[5]; };

__device__ __forceinline__ void load_clv1(Operand1 &o, gcptr base, unsigned lane_off, size_t rowbytes, int c) {
#pragma unroll
    for (int kk = 0; kk < 5; ++kk)
        o.v[kk] = __builtin_nontemporal_load(reinterpret_cast<const GLOBAL_AS double *>(base + (size_t)(c * NS + kk * 4) * rowbytes + lane_off));
}
__device__ __forceinline__ void load_tip1(Operand1 &o, const unsigned char *__restrict__ T, unsigned code, int q) {
    const unsigned char *t0 = T + code * NS + q;
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) o.v[kk] = (double)t0[kk * 4];
}
__device__ __forceinline__ void contract1(double (&acc)[5], const double *__restrict__ frag_c, const Operand1 &o) {
    // no software pipelining of the fragment reads: the ten registers of a second fragment column push the kernel further over
    // its 128-VGPR budget (58 spills instead of 46) and measured slower (0.827 vs 0.797 ms per C3 launch)
#pragma unroll
    for (int st = 0; st < 5; ++st) acc[st] = 0.0;
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) {
        double a[5];
#pragma unroll
        for (int st = 0; st < 5; ++st) a[st] = frag_c[(st * 5 + kk) * 16];
#pragma unroll
        for (int st = 0; st < 5; ++st) acc[st] = mfma4(a[st], o.v[kk], acc[st]);
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <typename F>
__device__ __forceinline__ void contract_stream1(const double *__restrict__ frag_c, const Operand1 &o, F &&consume) {
#pragma unroll
    for (int st = 0; st < 5; ++st) {
        double a[5];
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) a[kk] = frag_c[(st * 5 + kk) * 16];
        double acc = 0.0;
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) acc = mfma4(a[kk], o.v[kk], acc);
        consume(st, acc);
        __builtin_amdgcn_sched_barrier(0);
    }
}
__device__ __forceinline__ void load_cherry1(Operand1 &o, const OpSide &sd, unsigned ca, unsigned cb, int c, int q) {
    const Rows5 a = load_rows(sd.t0, ca, c, q), b = load_rows(sd.t1, cb, c, q);
    o.v[0] = a.a.x * b.a.x; o.v[1] = a.a.y * b.a.y; o.v[2] = a.b.x * b.b.x; o.v[3] = a.b.y * b.b.y; o.v[4] = a.c * b.c;
}
__device__ __forceinline__ void load_pitch1(Operand1 &o, const OpSide &sd, const double *__restrict__ f_inner,
                                            unsigned ca, unsigned cb, unsigned cc, int c, int q) {
    Operand1 w;
    load_cherry1(w, sd, ca, cb, c, q);
    double v[5];
    contract1(v, f_inner + c * 25 * 16, w);
    const Rows5 r = load_rows(sd.t2, cc, c, q);
    o.v[0] = v[0] * r.a.x; o.v[1] = v[1] * r.a.y; o.v[2] = v[2] * r.b.x; o.v[3] = v[3] * r.b.y; o.v[4] = v[4] * r.c;
}

// one op on one chunk of 16 patterns of one wave (the twin of chunk_op<false, true, false>)
__device__ __forceinline__ void chunk_op1(const NvOp &op, const double *__restrict__ sP, const unsigned char *__restrict__ sT,
                                          int p, int lane, Operand1 (&X)[4], int &xsc) {
    const int q = lane >> 4;
    constexpr size_t rowbytes = (size_t)TILE_PAT * 8;
    const size_t tabrow = (size_t)op.mpad * 8;
    const unsigned lane_off = (unsigned)(p >> 7) * (unsigned)(CLV_ROWS * rowbytes) + (unsigned)((size_t)q * rowbytes) + (unsigned)(p & (TILE_PAT - 1)) * 8u;
    const int lk = op.flags & 3, rk = (op.flags >> 2) & 3;
    const bool nt_store = (op.flags & OPF_NT_STORE) != 0;
    const int mode = op.mode;
    const bool chL = (op.flags & OPF_CHAIN_L) != 0, chR = (op.flags & OPF_CHAIN_R) != 0;
    const bool keep = (op.flags & OPF_NO_STORE) == 0;
    gcptr Lp = (gcptr)op.l.p0, Rp = (gcptr)op.r.p0;
    gptr O = (gptr)op.out;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void *)op.out, 0, 0x7FFFFFFF, 0x00020000);
    const double *fL = sP + (q * 4 + (lane & 3));
    const double *fR = fL + PFRAG;
    double mx = 0.0, site = 0.0;
    unsigned cl = 0, cl2 = 0, cl3 = 0, cr = 0, cr2 = 0, cr3 = 0;
    const double *fLi = fL + 2 * PFRAG;
    const double *fRi = (lk == SK_PITCH) ? op.r.f + (q * 4 + (lane & 3)) : fL + 2 * PFRAG;
    if (lk != SK_CLV) cl = *reinterpret_cast<const GLOBAL_AS unsigned char *>(Lp + p);
    if (lk >= SK_CHERRY) cl2 = *reinterpret_cast<const GLOBAL_AS unsigned char *>((gcptr)op.l.p1 + p);
    if (lk == SK_PITCH) cl3 = *reinterpret_cast<const GLOBAL_AS unsigned char *>((gcptr)op.l.p2 + p);
    if (rk != SK_CLV) cr = *reinterpret_cast<const GLOBAL_AS unsigned char *>(Rp + p);
    if (rk >= SK_CHERRY) cr2 = *reinterpret_cast<const GLOBAL_AS unsigned char *>((gcptr)op.r.p1 + p);
    if (rk == SK_PITCH) cr3 = *reinterpret_cast<const GLOBAL_AS unsigned char *>((gcptr)op.r.p2 + p);
    // plain tip sides by look-up of the fragment set's column (same rule as chunk_op: the whole wave or not at all)
    const bool lookL = mode < MODE_EVALUATE && lk == SK_TIP && !__any(cl >= 20u);
    const bool lookR = mode < MODE_EVALUATE && rk == SK_TIP && !__any(cr >= 20u);
    const double *tL = sP + (cl >> 2) * 16 + (cl & 3u) * 4 + q;
    const double *tR = sP + PFRAG + (cr >> 2) * 16 + (cr & 3u) * 4 + q;
    Operand1 curL, curR;
    if (lk == SK_TIP && !lookL) load_tip1(curL, sT, cl, q);
    else if (lk == SK_CLV && !chL) load_clv1(curL, Lp, lane_off, rowbytes, 0);
    if (rk == SK_TIP && !lookR) load_tip1(curR, sT, cr, q);
    else if (rk == SK_CLV && !chR) load_clv1(curR, Rp, lane_off, rowbytes, 0);
#pragma unroll
    for (int c = 0; c < NCAT; ++c) {
        Operand1 Y;
        if (chL) curL = X[c];
        if (chR) curR = X[c];
        if (lk == SK_CHERRY) load_cherry1(curL, op.l, cl, cl2, c, q);
        else if (lk == SK_PITCH) load_pitch1(curL, op.l, fLi, cl, cl2, cl3, c, q);
        if (rk == SK_CHERRY) load_cherry1(curR, op.r, cr, cr2, c, q);
        else if (rk == SK_PITCH) load_pitch1(curR, op.r, fRi, cr, cr2, cr3, c, q);
        if (mode >= MODE_EVALUATE) {
            contract_stream1(fR + c * 25 * 16, curR, [&](int st, double y) { site += curL.v[st] * y; });
            if (mode == MODE_EVALUATE_CAT) {
                double a = site;
                a += __shfl_xor(a, 16); a += __shfl_xor(a, 32);
                if (q == 0) *reinterpret_cast<GLOBAL_AS double *>(O + (size_t)c * tabrow + 8 * p) = a;
                site = 0.0;
            }
        } else {
            double aL[5];
            if (lookL) {
#pragma unroll
                for (int st = 0; st < 5; ++st) aL[st] = tL[c * 400 + st * 80];
            } else contract1(aL, fL + c * 25 * 16, curL);
            auto emit = [&](int st, double y) {
                const double o = aL[st] * y;
                mx = fmax(mx, o);
                Y.v[st] = o;
                if (!keep) return;
                typedef unsigned uvec2 __attribute__((ext_vector_type(2)));
                const uvec2 bits = __builtin_bit_cast(uvec2, o);
                const int soff = (c * NS + st * 4) * (int)rowbytes;
                if (nt_store) __builtin_amdgcn_raw_buffer_store_b64(bits, orsrc, lane_off, soff, 2);
                else __builtin_amdgcn_raw_buffer_store_b64(bits, orsrc, lane_off, soff, 0);
            };
            if (lookR) {
#pragma unroll
                for (int st = 0; st < 5; ++st) emit(st, tR[c * 400 + st * 80]);
            } else contract_stream1(fR + c * 25 * 16, curR, emit);
        }
        if (mode == MODE_NEWVIEW) X[c] = Y;
        if (c + 1 < NCAT) {
            if (lk == SK_CLV && !chL) load_clv1(curL, Lp, lane_off, rowbytes, c + 1);
            if (rk == SK_CLV && !chR) load_clv1(curR, Rp, lane_off, rowbytes, c + 1);
        }
    }
    int sc = 0;
    if (q == 0) {
        if (lk == SK_CLV) { if (chL) sc += xsc; else sc += *reinterpret_cast<const GLOBAL_AS int *>((gcptr)op.l_scl + 4 * p); }
        if (rk == SK_CLV) { if (chR) sc += xsc; else sc += *reinterpret_cast<const GLOBAL_AS int *>((gcptr)op.r_scl + 4 * p); }
    }
    if (mode == MODE_NEWVIEW) {
        mx = fmax(mx, __shfl_xor(mx, 16)); mx = fmax(mx, __shfl_xor(mx, 32));
        const bool n0 = mx < TWO_M256;
        if (__any(n0)) {                     // rare: numerical rescue of underflowing patterns
            const double f0 = n0 ? TWO_P256 : 1.0;
#pragma unroll
            for (int c = 0; c < NCAT; ++c)
#pragma unroll
                for (int k = 0; k < 5; ++k) X[c].v[k] *= f0;
            if (keep && n0) {
#pragma unroll 1
                for (int r = 0; r < NCAT * 5; ++r) {
                    GLOBAL_AS double *ptr = reinterpret_cast<GLOBAL_AS double *>(O + (size_t)((r / 5) * NS + (r % 5) * 4) * rowbytes + lane_off);
                    *ptr = *ptr * TWO_P256;
                }
            }
        }
        if (q == 0) { sc += n0 ? 1 : 0; if (keep) *reinterpret_cast<GLOBAL_AS int *>((gptr)op.out_scl + 4 * p) = sc; }
        xsc = sc;
    } else if (mode == MODE_SUMTABLE || mode == MODE_EVALUATE_CAT) {
        if (q == 0) *reinterpret_cast<GLOBAL_AS int *>((gptr)op.out_scl + 4 * p) = sc;
    } else {
        site += __shfl_xor(site, 16); site += __shfl_xor(site, 32);
        if (q == 0) *reinterpret_cast<GLOBAL_AS double *>(O + 8 * p) = log(site * 0.25) - sc * LOG_2_256;
    }
}

constexpr int OPL16_THREADS = 512;                     // 8 waves x 16 patterns = one 128-pattern tile
__global__ __launch_bounds__(OPL16_THREADS, 4) void k_oplist16(const NvOp *__restrict__ ops, const GeneRun *__restrict__ runs, int nruns,
                                                               int blocks_per_gene) {
    Operand1 X[4]; int xsc = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int k = 0; k < 5; ++k) X[c].v[k] = 0.0;
    extern __shared__ double sP[];                      // 2 parities x [left | right | inner] fragment sets + the tip-indicator table
    constexpr int PARITY_STRIDE = 3 * PFRAG;
    unsigned char *sT = reinterpret_cast<unsigned char *>(sP + 6 * PFRAG);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int gi = xcd + 8 * (slot / blocks_per_gene), blk = slot % blocks_per_gene;
    if (gi >= nruns) return;
    const GeneRun run = runs[gi];
    if (run.op_begin >= run.op_end) return;
    const int mpad = ops[run.op_begin].mpad;
    if (blk * TILE_PAT >= mpad) return;
    const int p = blk * TILE_PAT + wave * 16 + (lane & 15);
    const bool active = blk * TILE_PAT + wave * 16 < mpad;
    auto stage = [&](const NvOp &op, double *dst) {     // the three fragment sets of one op by LDS-DMA: 37.5 x 1 KiB wave-instructions
        for (int i = wave; i < 25; i += OPL16_THREADS / 64) {
            const int e = i * 128 + lane * 2;
            const double *g = (e < PFRAG) ? (op.pl ? op.pl : op.pr) + e : (op.pr ? op.pr : op.pl) + (e - PFRAG);
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)g, (__attribute__((address_space(3))) void *)(dst + i * 128), 16, 0, 0);
        }
        const int lk = op.flags & 3, rk = (op.flags >> 2) & 3;
        const double *inner = (lk == SK_PITCH) ? op.l.f : (rk == SK_PITCH) ? op.r.f : nullptr;
        if (inner == nullptr) return;
        for (int i = wave; i < 13; i += OPL16_THREADS / 64) {
            const int e = i * 128 + lane * 2;
            if (e < PFRAG) __builtin_amdgcn_global_load_lds((const GLOBAL_AS void *)(inner + e), (__attribute__((address_space(3))) void *)(dst + 2 * PFRAG + i * 128), 16, 0, 0);
        }
    };
    for (int i = tid; i < TIPTAB; i += OPL16_THREADS) sT[i] = (unsigned char)((code_mask(i / NS) >> (i % NS)) & 1u);
    stage(ops[run.op_begin], sP);
    __syncthreads();
    for (int oi = run.op_begin; oi < run.op_end; ++oi) {
        const NvOp &op = ops[oi];
        const int par = (oi - run.op_begin) & 1;
        const double *buf = sP + par * PARITY_STRIDE;
        if (oi + 1 < run.op_end) stage(ops[oi + 1], sP + (par ^ 1) * PARITY_STRIDE);
        if (active) chunk_op1(op, buf, sT, p, lane, X, xsc);
        __syncthreads();                                 // next fragments landed (vmcnt(0) + barrier), stores done
    }
}

// ------------------------------------------------------------------------------------------
// deterministic block reduction (fixed order: wave shuffle tree, then waves in index order)
// ------------------------------------------------------------------------------------------
template <int N, int WAVES>
__device__ __forceinline__ void block_sum(double (&v)[N], double (*red)[WAVES]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) { const double s = wave_sum(v[i]); if (lane == 0) red[i][wave] = s; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double s = red[i][0];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) s += red[i][w];
        v[i] = s;
    }
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
    block_sum<1, 4>(acc, red);
    if (threadIdx.x == 0) *r.out = acc[0];
}

// ------------------------------------------------------------------------------------------
// k_newton: Newton-Raphson on one branch length from its eigen-basis sumtable (RAxML "makenewz",
// SURVEY 8a-11 v).  The whole iteration runs on the device.  A single CU streams the 640 B/pattern
// sumtable at only ~70 GB/s (one CU's L2 rate), so a request is SPLIT over S workgroups (pattern
// slices of <= 128 patterns whose sumtable rows stay in registers for the whole iteration) that
// exchange three partial sums per evaluation through global memory.
//
// Exchange = data-tagged granules (cdna_hip_programming.md Guideline 16, form R2): every partial sum
// travels as two naturally aligned 8-byte words {tag = evaluation number, 32 bits of the double},
// each written by ONE relaxed agent-scope atomic store and read by relaxed agent-scope atomic loads.
// A granule is indivisible, so a consumer can never pair a tag with bytes of another evaluation;
// no ordering between different words is needed (no flag, no fence, no s_waitcnt).  A slot is only
// overwritten two evaluations later (parity double buffer), which its producer can reach only after
// every consumer has finished the evaluation in between.  The sums of all slices are combined in a
// fixed tree order that depends on S alone, S depends on the request alone: bit-reproducible
// whatever else is in the launch.  Every workgroup of a request executes the same evaluations
// because all see identical sums.
//
// FORWARD PROGRESS (round 3; rounds 1-2 relied on workgroups being dispatched in grid order, which holds
// for ONE queue of ONE process only: with co-tenants the XCDs fill with spinning slices of different
// requests whose partners are bound for another, equally full XCD -- the recorded time-out of four
// ranks on one GPU).  Slices are no longer tied to blockIdx: a workgroup that
// STARTS takes the next ticket of the launch (one atomic add) and the ticket names (request, slice)
// through a host-built table in request-major order.  Whoever holds ticket k is running, and so is
// (or has finished) every holder of a ticket < k: all requests but the newest are fully staffed and
// finish without waiting for anything that has not started; the newest waits only for workgroups the
// dispatcher can place in ANY free slot on ANY XCD.  Waiting never depends on the order in which
// the hardware dispatches, only on it dispatching at all.
// Belt and braces: the wait is bounded in WALL-CLOCK time (s_memrealtime, 100 MHz); the first slice
// that gives up raises the launch-wide abort word, which every spinning slice polls and every later
// k_newton launch of the stream checks on entry, so the device drains at once instead of spinning
// its bound per resident batch; affected requests report lnL = NaN and the host re-issues them
// through k_newton<..., SEQ = true>: ONE workgroup walks the request's slices in turn through the
// same per-slice code and combines the slice sums with the same shuffle tree -- no exchange, the
// bits of the split form (engine.cpp: Batch::run / smooth_pass; tests/test_gpu_newton_fallback.py).
// Control flow is the oracle's eng_newton_branch().
// ------------------------------------------------------------------------------------------
constexpr int NEWTON_THREADS = 512;           // 8 waves, four lanes per pattern -> 128 patterns per workgroup
constexpr int NEWTON_WAVES = NEWTON_THREADS / 64;
constexpr int NEWTON_SLICE = NEWTON_WAVES * 16;           // patterns per register-resident slice
constexpr int NEWTON_ROWS = CLV_ROWS / 4;     // sumtable rows per lane (20 doubles = 40 VGPRs)
static_assert(NEWTON_SLICE == NEWTON_SLICE_PAT, "kernels.h newton_split / newton_slice describe this kernel's slices");

struct NewtonShared {
    double exl[NCAT * NS][2];                 // (exp(lambda_i r_k t), lambda_i r_k)
    double red[3][NEWTON_WAVES];              // streaming form: every wave's sums; register form: [.][0] = service wave A's
    double fb[3][NEWTON_SLICE];               // register form: per pattern f, f', f'' (the service waves finish them)
    double xs[2][NEWTON_ROWS][64];            // register form: the two service waves' sumtable rows (LDS instead of VGPRs)
    double part[3][NEWTON_MAX_SPLIT];         // SEQ form: the slices' partial sums (what the split form exchanges)
    double bc[4];
    int ticket;
};

// Wave specialisation (ROLE).  Every wave owns 16 patterns (lanes 4j..4j+3 own pattern j, each 20 of its 80 sumtable
// rows, read once from HBM/L2 instead of once per evaluation; quarters combined with two DPP quad swaps).  Waves 0-5
// (ROLE 0) keep their rows in 40 VGPRs and do nothing else.  Waves 6 and 7 (ROLE 1, 2: the SERVICE waves) keep theirs in
// LDS and also do the scalar work of an evaluation: 40 of the 80 f64 exponentials each, then the log and the two divisions
// of 64 patterns each, wave 7 the cross-workgroup exchange.  The roles are separate instantiations of one body, so the
// register-hungry exp / log never meet 40 live sumtable registers: the kernel fits 8 waves per SIMD = 4 workgroups per CU
// = every slice of 128 C3 genes resident at once.  All waves run the same Newton control flow on the same broadcast sums
// and meet at the same barriers per evaluation.  Slices > 128 patterns (genes of more than 8192 patterns) stream
// their rows from L2 in every evaluation (REG = false, a second kernel with a 128-VGPR budget).
// SEQ: the no-exchange fallback -- this ONE workgroup is every slice of the request in turn (rows re-read per evaluation).
template <bool REG, int ROLE, bool SEQ>
__device__ __forceinline__ void newton_body(const NewtonReq &r, NewtonShared &sh, NewtonCtl *ctl,
                                            int S, int wg, int slice, long long timeout_ticks) {
    constexpr bool SVC = ROLE != 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mpad = r.mpad;
    const ModelDev *__restrict__ md = r.md;
    u64 *gran = reinterpret_cast<u64 *>(r.sync);       // [parity 2][slice NEWTON_MAX_SPLIT][6] granules
    int nevals = 0;
    double xr[NEWTON_ROWS]; const int sub = tid & 3;
    double sw = 0.0, ss = 0.0;             // service waves: weight and scaling count of the pattern the lane finishes
    // lane quarter `sub` owns rows c*20 + 4*st + sub (i = c*5 + st): the rows an MFMA D tile leaves in that quarter, so the
    // fused form (k_oplist<11>) adds the same rows in the same order
    auto row_of = [&](int i) { return (i / 5) * NS + 4 * (i % 5) + sub; };
    // the rows / weights of slice s (register form): once for the split form, per evaluation for SEQ
    auto load_slice = [&](int s) {
        const int p_begin = s * slice, p_end = min(mpad, p_begin + slice);
        // lanes beyond the slice read its last pattern (a valid address) and carry weight 0: no per-load branches
        const int p = p_begin + (tid >> 2), pc = min(p, p_end - 1);
        const double *col = r.sumtab + clv_index(sub, pc);          // tiled sumtable: rows of a tile are 128 doubles apart
        if (SVC) {
#pragma unroll 4
            for (int i = 0; i < NEWTON_ROWS; ++i) sh.xs[ROLE - 1][i][lane] = col[(size_t)((i / 5) * NS + 4 * (i % 5)) * TILE_PAT];      // read back by the same lane only
            const int pf = p_begin + 64 * (ROLE - 1) + lane;
            sw = 0.0; ss = 0.0;
            if (pf < p_end) { sw = r.weight[pf]; ss = (double)r.scl[pf]; }
        } else {
#pragma unroll
            for (int i = 0; i < NEWTON_ROWS; ++i) xr[i] = col[(size_t)((i / 5) * NS + 4 * (i % 5)) * TILE_PAT];
        }
    };
    if (REG && !SEQ) load_slice(wg);
    auto fill_exl = [&](double t) {        // service wave A: rows 0..39, B: rows 40..79
        if (lane < NCAT * NS / 2) {
            const int k = (ROLE - 1) * (NCAT * NS / 2) + lane;
            const double lr = md->eval[k % NS] * r.rates[k / NS];
            sh.exl[k][0] = exp(lr * t); sh.exl[k][1] = lr;
        }
    };
    // the three sums of slice s at the current sh.exl (ends with them in tot[] of service wave B, all waves past the same barriers)
    auto slice_sums = [&](int s, double (&tot)[3]) {
        const int p_begin = s * slice, p_end = min(mpad, p_begin + slice);
        if (REG) {
            if (SEQ) load_slice(s);
            double f = 0.0, f1 = 0.0, f2 = 0.0;
#pragma unroll
            for (int i = 0; i < NEWTON_ROWS; ++i) {
                const int row = row_of(i);
                newton_term(SVC ? sh.xs[ROLE - 1][i][lane] : xr[i], sh.exl[row][0], sh.exl[row][1], f, f1, f2);
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // at most 4 LDS pairs in flight
            }
            f = quad_sum(f); f1 = quad_sum(f1); f2 = quad_sum(f2);
            // the log and the two divisions per pattern are left to the service waves (their temporaries would not
            // fit beside the 40 sumtable registers at 8 waves per SIMD)
            if (sub == 0) { sh.fb[0][tid >> 2] = f; sh.fb[1][tid >> 2] = f1; sh.fb[2][tid >> 2] = f2; }
        } else {
            // streaming form (slices of more than 128 patterns): a thread owns whole patterns, rows in index order
            double acc[3] = {0.0, 0.0, 0.0};
            for (int p = p_begin + tid; p < p_end; p += NEWTON_THREADS) {
                const double w = r.weight[p];
                if (w == 0.0) continue;
                double f = 0.0, f1 = 0.0, f2 = 0.0;
#pragma unroll 8
                for (int row = 0; row < CLV_ROWS; ++row) newton_term(r.sumtab[clv_index(row, p)], sh.exl[row][0], sh.exl[row][1], f, f1, f2);
                double a0, a1, a2;
                newton_finish(f, f1, f2, w, (double)r.scl[p], a0, a1, a2);
                acc[0] += a0; acc[1] += a1; acc[2] += a2;
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) { const double sm = wave_sum(acc[i]); if (lane == 0) sh.red[i][wave] = sm; }
        }
        __syncthreads();
        tot[0] = tot[1] = tot[2] = 0.0;
        if (SVC && REG) {                    // lane l of service wave A / B finishes pattern l / 64 + l of the slice
            double a0 = 0.0, a1 = 0.0, a2 = 0.0;
            if (sw != 0.0) {
                const int j = 64 * (ROLE - 1) + lane;
                newton_finish(sh.fb[0][j], sh.fb[1][j], sh.fb[2][j], sw, ss, a0, a1, a2);
            }
            tot[0] = wave_sum(a0); tot[1] = wave_sum(a1); tot[2] = wave_sum(a2);
            if (ROLE == 1 && lane == 0) { sh.red[0][0] = tot[0]; sh.red[1][0] = tot[1]; sh.red[2][0] = tot[2]; }
        }
        __syncthreads();
        if (ROLE == 2) {                     // service wave B: the slice's sums in a fixed order
            if (REG) {
#pragma unroll
                for (int i = 0; i < 3; ++i) tot[i] = sh.red[i][0] + __shfl(tot[i], 0);
            } else {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    double sm = sh.red[i][0];
#pragma unroll
                    for (int w = 1; w < NEWTON_WAVES; ++w) sm += sh.red[i][w];
                    tot[i] = sm;
                }
            }
        }
    };

    auto eval_at = [&](double t, double &L, double &d1, double &d2) -> bool {
        if (SVC) fill_exl(t);
        __syncthreads();
        double tot[3] = {0.0, 0.0, 0.0};
        if (SEQ) {
            for (int s = 0; s < S; ++s) {
                slice_sums(s, tot);
                if (ROLE == 2 && lane == 0) { sh.part[0][s] = tot[0]; sh.part[1][s] = tot[1]; sh.part[2][s] = tot[2]; }
                __syncthreads();             // sh.fb / sh.red / sh.xs are reused by the next slice
            }
        } else slice_sums(wg, tot);
        if (ROLE == 2) {
            bool bad = false;
            if (S > 1) {
                if (SEQ) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) tot[i] = wave_sum(lane < S ? sh.part[i][lane] : 0.0);      // the split form's tree
                } else bad = newton_exchange(gran, S, wg, nevals, r.tag_base, tot, ctl, timeout_ticks);
            }
            if (lane == 0) { sh.bc[0] = tot[0]; sh.bc[1] = tot[1]; sh.bc[2] = tot[2]; if (bad) sh.bc[3] = 1.0; }
        }
        __syncthreads();
        ++nevals;
        L = uni(sh.bc[0]); d1 = uni(sh.bc[1]); d2 = uni(sh.bc[2]);
        return sh.bc[3] == 0.0;
    };

    double t, L, d1, d2;
    const bool ok = newton_drive(r.t0, r.max_iter, r.tol, eval_at, t, L, d1, d2);
    if (ROLE == 2 && lane == 0 && wg == 0) {
        atomicAdd(&ctl->n_requests, 1ull); atomicAdd(&ctl->n_evals, (unsigned long long)nevals);
        if (!ok) { t = r.t0; L = __builtin_nan(""); }      // exchange gave up: reported, the host re-issues the request (SEQ form)
        r.out[0] = t; r.out[1] = L; r.out[2] = d1; r.out[3] = d2;
        if (r.t_dev0) { *r.t_dev0 = t; *r.t_dev1 = t; }
    }
    if (r.patlnl != nullptr) {                  // per-pattern lnL at the returned length (every slice its own patterns)
        __syncthreads();
        if (SVC) fill_exl(t);
        __syncthreads();
        for (int s = SEQ ? 0 : wg; s < (SEQ ? S : wg + 1); ++s) {
            const int p_begin = s * slice, p_end = min(mpad, p_begin + slice);
            if (REG) {
                if (SEQ) load_slice(s);
                double f = 0.0;
#pragma unroll
                for (int i = 0; i < NEWTON_ROWS; ++i) f += (SVC ? sh.xs[ROLE - 1][i][lane] : xr[i]) * sh.exl[row_of(i)][0];
                f = quad_sum(f);
                if (sub == 0) sh.fb[0][tid >> 2] = f;
                __syncthreads();
                if (SVC) {                           // the logs, again by the service waves
                    const int j = 64 * (ROLE - 1) + lane, p = p_begin + j;
                    if (p < p_end) r.patlnl[p] = (sw != 0.0) ? log(sh.fb[0][j] * 0.25) - ss * LOG_2_256 : 0.0;
                }
                if (SEQ) __syncthreads();
            } else {
                for (int p = p_begin + tid; p < p_end; p += NEWTON_THREADS) {
                    double f = 0.0;
                    if (r.weight[p] != 0.0) {
                        for (int row = 0; row < CLV_ROWS; ++row) f += r.sumtab[clv_index(row, p)] * sh.exl[row][0];
                        f = log(f * 0.25) - r.scl[p] * LOG_2_256;
                    }
                    r.patlnl[p] = f;
                }
            }
        }
    }
}

// (the streaming form is a second kernel so that its loads do not cost the register-resident form its 8 waves per
// SIMD; each kernel has its own ticket table)
// SEQ = false: one workgroup per TICKET (ticket -> request through `ticket_req`, slice = ticket - first ticket of the request);
// SEQ = true : one workgroup per entry of `ticket_req` = one whole request (the host lists the requests to re-issue)
template <bool REG, bool SEQ>
__global__ __launch_bounds__(NEWTON_THREADS, (REG && !SEQ) ? 8 : 4) void k_newton(const ModelDev *__restrict__ md,
                                                                                  const NewtonReq *__restrict__ reqs,
                                                                                  const int *__restrict__ ticket_req, NewtonCtl *ctl,
                                                                                  long long timeout_ticks) {
    __shared__ NewtonShared sh;
    constexpr int K = REG ? 0 : 1;
    if (threadIdx.x == 0) {
        sh.bc[3] = 0.0;
        sh.ticket = SEQ ? (int)blockIdx.x : __hip_atomic_fetch_add(&ctl->ticket[K], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // a launch of a stream whose earlier launch gave up does not start spinning again: it reports NaN at once
        if (!SEQ && __hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) sh.bc[3] = 2.0;
    }
    __syncthreads();
    int ticket = sh.ticket;
    if (!SEQ && (unsigned)ticket >= gridDim.x) {      // cannot happen with armed counters; never index the tables with it
        if (threadIdx.x == 0) __hip_atomic_store(&ctl->abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const NewtonReq &r = reqs[ticket_req[ticket]];    // by reference: a private copy would live in scratch (rates[] is indexed dynamically)
    const int mpad = r.mpad, wg = SEQ ? 0 : ticket - r.ticket0;
    // the split depends on the request alone (not on what else is in the launch): results are reproducible
    // whatever the batch composition
    const int S = newton_split(mpad), slice = newton_slice(mpad);
    if (sh.bc[3] == 2.0) {
        if (threadIdx.x == 0 && wg == 0) { r.out[0] = r.t0; r.out[1] = __builtin_nan(""); r.out[2] = 0.0; r.out[3] = 0.0; }
    } else {
        const int wave = threadIdx.x >> 6;
        if (wave == NEWTON_WAVES - 1) newton_body<REG, 2, SEQ>(r, sh, ctl, S, wg, slice, timeout_ticks);
        else if (wave == NEWTON_WAVES - 2) newton_body<REG, 1, SEQ>(r, sh, ctl, S, wg, slice, timeout_ticks);
        else newton_body<REG, 0, SEQ>(r, sh, ctl, S, wg, slice, timeout_ticks);
    }
    if (!SEQ) {                                   // the workgroup that finishes last re-arms the ticket counter for the stream's next launch
        __syncthreads();
        if (threadIdx.x == 0) {
            const int d = __hip_atomic_fetch_add(&ctl->done[K], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == (int)gridDim.x - 1) {
                __hip_atomic_store(&ctl->ticket[K], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ctl->done[K], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_gather: build a replicate's code matrix from gene matrices already resident in HBM (byte copies,
// HBM-bound; one thread per pattern, rows walked in a loop so stores of a wavefront are contiguous)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather(const GatherSeg *__restrict__ segs) {
    const GatherSeg g = segs[blockIdx.x];
    const int p = blockIdx.y * 256 + threadIdx.x;
    if (p >= g.npat) return;
    g.dst_w[g.dst_off + p] = g.w[p];
    for (int t = 0; t < g.ntax_dst; ++t) {
        const int row = g.rowmap[t];
        g.dst[(size_t)t * g.dst_mpad + g.dst_off + p] = row >= 0 ? g.src[(size_t)row * g.src_mpad + p] : (uint8_t)(NCODES - 1);
    }
}

// ------------------------------------------------------------------------------------------
// k_sh: SH-like local support of one split per workgroup (see ShReq).  Thread = resamples tid, tid+256, ...;
// the three per-pattern vectors are a few KB and stay in L1/L2.  Integer hash + gathers: latency/ALU-bound, tiny.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
    return z;
}
__global__ __launch_bounds__(256) void k_sh(const ShReq *__restrict__ reqs) {
    __shared__ double red[3][4];
    const ShReq r = reqs[blockIdx.x];
    const int tid = threadIdx.x;
    double orig[3] = {0.0, 0.0, 0.0};
    for (int j = tid; j < r.nsites; j += 256) { const int p = r.site2pat[j]; orig[0] += r.l0[p]; orig[1] += r.l1[p]; orig[2] += r.l2[p]; }
    block_sum<3, 4>(orig, red);
    const double delta = orig[0] - fmax(orig[1], orig[2]);
    double cnt[1] = {0.0};
    if (delta > 0.0) {
        const unsigned long long base = (r.seed + 1ull) * 0x9E3779B97F4A7C15ull, ns = (unsigned long long)r.nsites;
        for (int b = tid; b < r.nboot; b += 256) {
            double s0 = 0.0, s1 = 0.0, s2 = 0.0;
            const unsigned long long k0 = base + (unsigned long long)b * ns;
            for (int j = 0; j < r.nsites; ++j) {
                const int p = r.site2pat[(int)(mix64(k0 + (unsigned long long)j) % ns)];
                s0 += r.l0[p]; s1 += r.l1[p]; s2 += r.l2[p];
            }
            s0 -= orig[0]; s1 -= orig[1]; s2 -= orig[2];            // centred
            const double best = fmax(s0, fmax(s1, s2));
            // advantage of the best arrangement over the better of the other two
            const double second = (best == s0) ? fmax(s1, s2) : (best == s1 ? fmax(s0, s2) : fmax(s0, s1));
            if (best - second < delta) cnt[0] += 1.0;
        }
    }
    __shared__ double red1[1][4];
    block_sum<1, 4>(cnt, red1);
    if (tid == 0) *r.out = (r.nboot > 0) ? cnt[0] / (double)r.nboot : 0.0;
}

// ------------------------------------------------------------------------------------------
// k_g20: FastTree's Gamma20 likelihood of one gene per workgroup from its 20 x mpad table (G20Req).  16 B per rate and
// pattern read once per evaluation of (alpha, rescale); the table of a C3 gene is 160 KB (L2-resident across the fit's
// ~50 evaluations): latency-bound, tiny.  Fixed-order reduction.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_g20(const G20Req *__restrict__ reqs) {
    __shared__ double red[1][4];
    __shared__ double w[G20_RATES];
    const G20Req &r = reqs[blockIdx.x];
    if (threadIdx.x < G20_RATES) w[threadIdx.x] = r.w[threadIdx.x];
    __syncthreads();
    const size_t M = (size_t)r.mpad;
    double acc[1] = {0.0};
    for (int p = threadIdx.x; p < r.mpad; p += 256) {
        const double wt = r.weight[p];
        double l = 0.0;
        if (wt != 0.0) {
            int m = r.cnt[p];
#pragma unroll
            for (int j = 1; j < G20_RATES / 4; ++j) m = min(m, r.cnt[(size_t)j * M + p]);
            double sum = 0.0;
#pragma unroll
            for (int j = 0; j < G20_RATES / 4; ++j) {
                const int d = r.cnt[(size_t)j * M + p] - m;            // a traversal rescued more often holds 2^(256 d) x larger numbers
                const double f = ldexp(1.0, -256 * d);
#pragma unroll
                for (int c = 0; c < 4; ++c) sum += w[4 * j + c] * r.table[(size_t)(4 * j + c) * M + p] * f;
            }
            l = log(sum) - m * LOG_2_256;
            acc[0] += wt * l;
        }
        if (r.patlnl) r.patlnl[p] = l;
    }
    block_sum<1, 4>(acc, red);
    if (threadIdx.x == 0) *r.out = acc[0];
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
void launch_g20(const G20Req *reqs, int n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_g20, dim3((unsigned)n), dim3(256), 0, s, reqs);
}
void launch_sh(const ShReq *reqs, int n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_sh, dim3((unsigned)n), dim3(256), 0, s, reqs);
}
void launch_gather(const GatherSeg *segs, int nsegs, int max_npat, hipStream_t s) {
    if (nsegs <= 0 || max_npat <= 0) return;
    hipLaunchKernelGGL(k_gather, dim3((unsigned)nsegs, (unsigned)((max_npat + 255) / 256)), dim3(256), 0, s, segs);
}
void launch_pmat(const ModelDev *model, const PmatReq *reqs, double *frags, int n, hipStream_t s, bool per_request) {
    if (n <= 0) return;
    const int per_block = PMAT_THREADS / 64;                  // one wave per request, persistent beyond 2 waves per SIMD
    const int blocks = std::min((n + per_block - 1) / per_block, 512);
    if (per_request) hipLaunchKernelGGL(k_pmat<true>, dim3(blocks), dim3(PMAT_THREADS), 0, s, model, reqs, frags, n);
    else hipLaunchKernelGGL(k_pmat<false>, dim3(blocks), dim3(PMAT_THREADS), 0, s, model, reqs, frags, n);
}
void launch_eigfrags(const ModelDev *model, double *frags2, hipStream_t s) {
    hipLaunchKernelGGL(k_eigfrags, dim3(1), dim3(256), 0, s, model, frags2);
}
void launch_eigfrags_n(const ModelDev *models, double *frags2, int n, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_eigfrags, dim3((unsigned)n), dim3(256), 0, s, models, frags2);
}
static int oplist_variant() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("PML_OPLIST_VARIANT"); v = e ? atoi(e) : 1; }
    return v;
}
static long long newton_timeout_ticks();
static void launch_oplist_one(const NvOp *ops, const GeneRun *runs, int nruns, int bpg, bool one_part, bool any_pitch, bool chained, hipStream_t s, NewtonCtl *ctl);
int fused_oplist_capacity();
bool fuse_big_genes();
void launch_oplist(const NvOp *ops, const GeneRun *runs, int nruns, int max_mpad, bool any_pitch, bool chained, hipStream_t s, NewtonCtl *ctl) {
    if (nruns <= 0) return;
    const int bpg = (max_mpad + PAT_PER_WG - 1) / PAT_PER_WG;
    if (!ctl) { launch_oplist_one(ops, runs, nruns, bpg, false, any_pitch, chained, s, nullptr); return; }
    // Launches with fused Newton tails: the workgroups of a gene WAIT for each other, and slots are claimed by ticket (whoever
    // holds a ticket is running, and so is every holder of a lower ticket of its partition: all genes but the newest are fully
    // staffed).  What is left to the hardware is to place the newest gene's missing tiles when slots come free -- and that it
    // does NOT do reliably when they are bound to one XCD (workgroup b runs on XCD b % 8: tools/ubench_xcc.hip): with one ticket
    // partition per XCD, tools/ubench_ticket.hip stalls for good with gangs of 63 (always), 35 and 40 (now and then) although
    // slots are free, never with gangs <= 32 (an XCD always has room for 32 of these workgroups, one per CU), and 16 genes of
    // 200 x 5000 (35 tiles) did the same in the engine.  With ONE partition over the whole device -- a gang's members on any
    // XCD -- no configuration ever stalled (72 configurations, gangs 33..64, up to 5x oversubscribed:
    // profiles/r03_ubench_ticket_dispatch.txt; the engine: C4 shard, 2457 workgroups on 512 slots).  Hence:
    //   bpg <= 32: one ticket partition per XCD, which keeps a gene's tiles -- and its fragment sets -- on one L2 (C3);
    //   bpg  > 32: one partition over the device (a C4 shard searches in 10.65 s against 11.0 s un-fused).
    // PML_FUSE_BIG=0 restores the conservative rule for the second case (fused only when the whole launch is resident at once).
    if (bpg <= 32) { launch_oplist_one(ops, runs, nruns, bpg, false, any_pitch, true, s, ctl); return; }
    if (fuse_big_genes()) { launch_oplist_one(ops, runs, nruns, bpg, true, any_pitch, true, s, ctl); return; }
    const int cap = fused_oplist_capacity();
    const int genes_per_launch = std::max(1, cap / bpg);
    for (int off = 0; off < nruns; off += genes_per_launch)
        launch_oplist_one(ops, runs + off, std::min(genes_per_launch, nruns - off), bpg, true, any_pitch, true, s, ctl);
}
bool fuse_big_genes() {
    static const bool on = !(std::getenv("PML_FUSE_BIG") && std::atoi(std::getenv("PML_FUSE_BIG")) == 0);
    return on;
}
int fused_oplist_capacity() {
    static const int cap = [] {
        int dev = 0, cus = 256, nb = 2; hipDeviceProp_t p;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
        const size_t lds15 = (size_t)6 * PFRAG * sizeof(double) + 512;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_oplist<15>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds15);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void *>(k_oplist<15>), 256, lds15) != hipSuccess || nb < 1) nb = 1;
        if (nb > 2) nb = 2;                                  // 256 VGPRs: two waves per SIMD whatever the query says
        return cus * nb;
    }();
    return cap;
}
static void launch_oplist_one(const NvOp *ops, const GeneRun *runs, int nruns, int bpg_in, bool one_part, bool any_pitch, bool chained, hipStream_t s, NewtonCtl *ctl) {
    const long long to = newton_timeout_ticks();
    const int bpg = one_part ? -bpg_in : bpg_in;
    const dim3 grid((unsigned)(one_part ? nruns * bpg_in : ((nruns + 7) / 8) * 8 * bpg_in)), block(256);
    int v = oplist_variant();
    if (any_pitch && (v == 2 || v == 3)) v = 1;          // pitchfork regions are not combined with double buffering
    if (chained) {                                       // the descriptors carry OPF_CHAIN_* flags: only these variants honour them
        static const int cv = std::getenv("PML_CHAIN_VARIANT") ? std::atoi(std::getenv("PML_CHAIN_VARIANT")) : 11;
        v = cv == 9 ? 9 : (cv == 16 ? 16 : (cv == 10 ? 10 : 11));
        if (ctl) v = 15;
    }
    if (v == 16) {                                       // one pattern per lane, 8 waves per tile (k_oplist16)
        const size_t lds16 = (size_t)6 * PFRAG * sizeof(double) + 512;
        static const hipError_t big16 = hipFuncSetAttribute(reinterpret_cast<const void *>(k_oplist16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16);
        if (big16 == hipSuccess) {
            hipLaunchKernelGGL(k_oplist16, grid, dim3(OPL16_THREADS), lds16, s, ops, runs, nruns, bpg_in);
            return;
        }
        v = 11;
    }
    const bool dbuf = (v == 2 || v == 3);
    size_t lds = (size_t)((dbuf ? 4 : 2) + (any_pitch ? 1 : 0)) * PFRAG * sizeof(double) + 512;   // 38.9 KB with pitchforks: 4 per CU
    if (v == 11 || v == 15 || v == 10) {
        lds = (size_t)6 * PFRAG * sizeof(double) + 512;      // 77.3 KB: two workgroups per CU, which is what its 256 VGPRs allow anyway
        static const hipError_t big11 = [&] {
            const hipError_t a = hipFuncSetAttribute(reinterpret_cast<const void *>(k_oplist<11>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            const hipError_t b = hipFuncSetAttribute(reinterpret_cast<const void *>(k_oplist<15>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_oplist<10>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            return a != hipSuccess ? a : b;
        }();
        if (big11 != hipSuccess && v == 11) {                // no 77 KB of dynamic LDS: the single-buffered chained variant (25.6-38.4 KB) does the same work
            v = 9; lds = (size_t)(2 + (any_pitch ? 1 : 0)) * PFRAG * sizeof(double) + 512;
        }
    }
    const int ap = any_pitch ? 1 : 0;
    switch (v) {
        case 0: hipLaunchKernelGGL(k_oplist<0>, grid, block, lds, s, ops, runs, nruns, bpg, ap, ctl, to); break;
        case 2: hipLaunchKernelGGL(k_oplist<2>, grid, block, lds, s, ops, runs, nruns, bpg, ap, ctl, to); break;
        case 3: hipLaunchKernelGGL(k_oplist<3>, grid, block, lds, s, ops, runs, nruns, bpg, ap, ctl, to); break;
        case 5: hipLaunchKernelGGL(k_oplist<5>, grid, block, lds, s, ops, runs, nruns, bpg, ap, ctl, to); break;
        case 9: hipLaunchKernelGGL(k_oplist<9>, grid, block, lds, s, ops, runs, nruns, bpg, ap, ctl, to); break;
        case 10: hipLaunchKernelGGL(k_oplist<10>, grid, block, lds, s, ops, runs, nruns, bpg, ap, ctl, to); break;
        case 11: hipLaunchKernelGGL(k_oplist<11>, grid, block, lds, s, ops, runs, nruns, bpg, ap, ctl, to); break;
        case 15: hipLaunchKernelGGL(k_oplist<15>, grid, block, lds, s, ops, runs, nruns, bpg, ap, ctl, to); break;
        default: hipLaunchKernelGGL(k_oplist<1>, grid, block, lds, s, ops, runs, nruns, bpg, ap, ctl, to); break;
    }
}
void launch_reduce(const ReduceReq *reqs, int n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_reduce, dim3(n), dim3(256), 0, s, reqs);
}
static long long newton_timeout_ticks() {
    // wall-clock bound of one exchange wait, in 100 MHz ticks; PML_NEWTON_TIMEOUT_US is the test hook (0 = give up at the
    // first unsuccessful poll: forces the SEQ fallback for every request that is split)
    static const long long v = [] { const char *e = std::getenv("PML_NEWTON_TIMEOUT_US"); return e ? std::atoll(e) * 100LL : 2000000LL * 100LL; }();
    return v;
}
void launch_newton(const ModelDev *model, const NewtonReq *reqs, const int *tickets, int nreg, int nstream, NewtonCtl *ctl, hipStream_t s) {
    // one workgroup per ticket; `tickets` lists the register-form tickets first, then the streaming-form ones (engine.cpp
    // builds it in request order).  No chunking, no co-residency requirement: see FORWARD PROGRESS above.
    const long long to = newton_timeout_ticks();
    if (nreg > 0) hipLaunchKernelGGL((k_newton<true, false>), dim3((unsigned)nreg), dim3(NEWTON_THREADS), 0, s, model, reqs, tickets, ctl, to);
    if (nstream > 0) hipLaunchKernelGGL((k_newton<false, false>), dim3((unsigned)nstream), dim3(NEWTON_THREADS), 0, s, model, reqs, tickets + nreg, ctl, to);
}
void launch_newton_seq(const ModelDev *model, const NewtonReq *reqs, const int *req_list, int nreg, int nstream, NewtonCtl *ctl, hipStream_t s) {
    if (nreg > 0) hipLaunchKernelGGL((k_newton<true, true>), dim3((unsigned)nreg), dim3(NEWTON_THREADS), 0, s, model, reqs, req_list, ctl, 0LL);
    if (nstream > 0) hipLaunchKernelGGL((k_newton<false, true>), dim3((unsigned)nstream), dim3(NEWTON_THREADS), 0, s, model, reqs, req_list + nreg, ctl, 0LL);
}

}  // namespace pml
