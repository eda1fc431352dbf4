// Assembly stages of the gato hot path for gfx950: CSR -> dense scatter (A1), Schur complement +
// block-Jacobi blocks + gamma (A2), symmetric-stair off-diagonals (A3), dz back-substitution (A9).
//
// Replaces gato_convert_kkt_format / csr_to_custom_G / csr_to_custom_C (src/gato_schur.cuh:674-756),
// gato_form_schur_jacobi[_inner] (:13-494), gato_form_ss[_inner] (:497-670), gato_compute_dz[_inner]
// (:758-879) and the block helpers they call (src/gato_utils.cuh: invertMatrix :468-586,
// mat_mat_prod :609-659, mat_vec_prod :595-606, gato_ATx :664-679, store/load_block_bd :44-119).
//
// Design: one wavefront (64 lanes) per knot (per matrix in the inversion kernel), every Q_k / R_k inverted
// exactly once into a separate inverse buffer (the reference inverts each Q_k twice and overwrites G_dense in
// place while neighbouring blocks still read it - SURVEY.md D3).
//  * Gauss-Jordan without pivoting, the reference's elimination order (gato_utils.cuh:468-586), REGISTER
//    resident: lane c holds column c of the augmented [A | I]; per pivot the pivot column is broadcast with
//    v_readlane (the pivot lane is a compile-time constant after unrolling) and every lane updates its column
//    with n FMAs - no LDS, no barriers (the reference pays two __syncthreads and an LDS round trip per pivot).
//  * the S x S x S products (phi = A Q^-1, theta = phi A^T + B R^-1 B^T, the four stair products) run on the
//    matrix cores: v_mfma_f32_16x16x4_f32 / v_mfma_f64_16x16x4_f64 tiles with LDS operands (S = 14 pads to one
//    16 x 16 tile, S = 32 is 2 x 2 tiles).  f32 MFMA is a k-ordered fmaf chain, i.e. the reference's own
//    accumulation order.  This is the only place of the hot path where a dense contraction exists.
// Boundary fixes D1, D2, D4.
#include "gato_common.h"

namespace gato {
namespace {

constexpr int WAVE = 64;

// ---- N4: blocks handed over directly: copy G_dense adding rho on the diagonals of Q_k and R_k (what
// csr_to_custom_G does to structurally present diagonal entries, gato_schur.cuh:697,:700)
template <typename T, int S, int C>
__global__ void add_rho_kernel(const T *__restrict__ G_in, T rho, T *__restrict__ Gd, size_t n_per_sys, int B)
{
    constexpr int SS = S * S, CC = C * C;
    const size_t total = n_per_sys * B;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t w = (i % n_per_sys) % (SS + CC);
        const bool diag = w < SS ? (w % (S + 1) == 0) : ((w - SS) % (C + 1) == 0);
        Gd[i] = G_in[i] + (diag ? rho : (T)0);
    }
}

// ---- block helpers on LDS operands, executed by one wavefront ---------------------------------
__device__ __forceinline__ void wave_sync() { __syncthreads(); }   // blockDim == 64: one wave

template <typename T>
__device__ __forceinline__ void copy_in(T *dst, const T *__restrict__ src, int n, int lane)
{
    for (int i = lane; i < n; i += WAVE) dst[i] = src[i];
}

// out(m) = A(m x n) x (mat_vec_prod, :595-606)
template <typename T, int m, int n>
__device__ __forceinline__ void mv(T *out, const T *A, const T *x, int lane)
{
    for (int r = lane; r < m; r += WAVE) {
        T res = (T)0;
#pragma unroll
        for (int c = 0; c < n; ++c) res = gato::fmaT(A[r + c * m], x[c], res);   // explicit: every kernel contracts alike
        out[r] = res;
    }
}
// out(n) = A(m x n)^T x (gato_ATx, :664-679)
template <typename T, int m, int n>
__device__ __forceinline__ void mTv(T *out, const T *A, const T *x, int lane)
{
    for (int i = lane; i < n; i += WAVE) {
        T res = (T)0;
#pragma unroll
        for (int t = 0; t < m; ++t) res = gato::fmaT(A[i * m + t], x[t], res);
        out[i] = res;
    }
}

// ---- register-resident Gauss-Jordan -----------------------------------------------------------------
__device__ __forceinline__ float readlane_c(float v, int l)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ double readlane_c(double v, int l)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// lane c < n holds column c of A, lane n + c holds column c of I; on return lane n + c holds column c of A^-1.
// Same arithmetic as invertMatrix (gato_utils.cuh:468-495): pivot row *= 1/pv, other rows -= col[r]/pv * row.
template <typename T, int n>
__device__ __forceinline__ void gj_inverse_reg(T (&col)[n])
{
#pragma unroll
    for (int p = 0; p < n; ++p) {
        const T pvinv = (T)1 / readlane_c(col[p], p);
        const T prow = col[p] * pvinv;                       // this lane's element of the scaled pivot row
#pragma unroll
        for (int r = 0; r < n; ++r) {
            if (r != p) {
                const T f = readlane_c(col[r], p);           // A[r][p], wave-uniform
                col[r] = gato::fmaT(-f, prow, col[r]);
            }
        }
        col[p] = prow;
    }
}

// Inverts the n x n column-major matrix at src (LDS or global) into dst (column-major), one wave, n <= 32.
template <typename T, int n>
__device__ __forceinline__ void invert_to(const T *src, T *dst, int lane, T scale)
{
    static_assert(2 * n <= WAVE, "augmented matrix must fit one wavefront");
    T col[n];
#pragma unroll
    for (int r = 0; r < n; ++r) col[r] = lane < n ? src[lane * n + r] : (T)(lane - n == r);
    gj_inverse_reg<T, n>(col);
    if (lane >= n && lane < 2 * n) {
#pragma unroll
        for (int r = 0; r < n; ++r) dst[(lane - n) * n + r] = scale * col[r];
    }
}

// ---- small GEMMs on the matrix cores --------------------------------------------------------------------
template <typename T> struct Mfma;
template <> struct Mfma<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    __device__ static __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ int row(int lane, int j) { return 4 * (lane >> 4) + j; }       // C/D: row = 4*(lane/16)+reg
};
template <> struct Mfma<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    __device__ static __forceinline__ acc_t mma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    __device__ static __forceinline__ int row(int lane, int j) { return (lane >> 4) + 4 * j; }       // f64: row = lane/16 + 4*reg
};

// acc tile (mt, nt) += A(M x KD, col-major, ld M) * B   where B is KD x N col-major (ld KD), or, if TB, given as
// N x KD col-major (ld N) and used transposed.  Operands in LDS; lanes outside the matrix feed zeros.
template <typename T, int M, int KD, int N, bool TB>
__device__ __forceinline__ typename Mfma<T>::acc_t mfma_tile(const T *A, const T *B, int mt, int nt, int lane,
                                                             typename Mfma<T>::acc_t acc)
{
    const int i = mt * 16 + (lane & 15), jn = nt * 16 + (lane & 15);
#pragma unroll
    for (int ks = 0; ks < (KD + 3) / 4; ++ks) {
        const int kk = ks * 4 + (lane >> 4);
        const T a = (i < M && kk < KD) ? A[kk * M + i] : (T)0;
        const T b = (jn < N && kk < KD) ? (TB ? B[kk * N + jn] : B[jn * KD + kk]) : (T)0;
        acc = Mfma<T>::mma(a, b, acc);
    }
    return acc;
}

// out(M x N, col-major) = sign * (A B [+ A2 B2^T ...]) : generic driver taking a per-element epilogue
template <typename T, int M, int N, typename TileFn, typename StoreFn>
__device__ __forceinline__ void mfma_for_tiles(int lane, TileFn tile, StoreFn store)
{
#pragma unroll
    for (int mt = 0; mt < (M + 15) / 16; ++mt) {
#pragma unroll
        for (int nt = 0; nt < (N + 15) / 16; ++nt) {
            typename Mfma<T>::acc_t acc = {0, 0, 0, 0};
            acc = tile(mt, nt, acc);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = mt * 16 + Mfma<T>::row(lane, j), c = nt * 16 + (lane & 15);
                if (r < M && c < N) store(r, c, acc[j]);
            }
        }
    }
}

// ---- A2a: invert every Q_k and R_k once: one wavefront per matrix, register resident ----------------------
template <typename T, int S, int C>
__global__ __launch_bounds__(256) void invert_G_kernel(const T *__restrict__ Gd, T *__restrict__ Ginv, int K,
                                                       BatchStride bs)
{
    constexpr int SS = S * S, CC = C * C;
    Gd += blockIdx.y * bs.g; Ginv += blockIdx.y * bs.g;
    const int lane = threadIdx.x & 63;
    const int nmat = 2 * K - 1;                                   // Q_0, R_0, Q_1, ..., Q_{K-1}
    const int m_hi = 2 * bs.k_hi - 1 < nmat ? 2 * bs.k_hi - 1 + (bs.k_hi < K ? 1 : 0) : nmat;   // Q and R of the knots [k_lo, k_hi)
    for (int m = 2 * bs.k_lo + blockIdx.x * 4 + (threadIdx.x >> 6); m < m_hi; m += gridDim.x * 4) {
        const int k = m >> 1;
        const size_t off = (size_t)k * (SS + CC);
        if ((m & 1) == 0) invert_to<T, S>(Gd + off, Ginv + off, lane, (T)1);
        else invert_to<T, C>(Gd + off + SS, Ginv + off + SS, lane, (T)1);
    }
}

// XCD-aware workgroup -> (system, knot) map for the one-wave-per-knot launches.  Workgroups are dealt round-robin over
// the 8 XCDs in launch order, and each XCD has its own L2: with the plain map the knots k-1, k, k+1 - which share
// Q_{k-1}^-1 (Schur), Pinv[k+-1].main and S[k+1].left (stair), lambda_{k+1} (dz) - sit on three different L2s and every
// shared block comes from HBM once per reader.  Here the workgroups of one XCD take CONSECUTIVE knots, so the neighbour's
// blocks are L2 hits.  A placement hint only: any dispatch order gives the same results.
__device__ __forceinline__ void xcd_knot_map(int &sys, int &kidx)
{
    const unsigned total = gridDim.x * gridDim.y, L = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned t8 = total & ~7u;
    const unsigned M = L < t8 ? (L & 7u) * (t8 >> 3) + (L >> 3) : L;
    sys = (int)(M / gridDim.x);
    kidx = (int)(M % gridDim.x);
}

// ---- A2b: Schur blocks, block-Jacobi main blocks, gamma (gato_schur.cuh:13-460) -----------------
// waves per SIMD the register allocation must leave room for (one-wave workgroups: latency-bound launches want many knots in flight)
#ifndef GATO_SCHUR_WAVES_F64
#define GATO_SCHUR_WAVES_F64 4
#endif
#ifndef GATO_SCHUR_WAVES_F32
#define GATO_SCHUR_WAVES_F32 5
#endif
#ifndef GATO_GATHER_WAVES_F64
#define GATO_GATHER_WAVES_F64 1
#endif
#ifndef GATO_GATHER_WAVES_F32
#define GATO_GATHER_WAVES_F32 1
#endif
template <typename T, int S> struct AsmWaves { static constexpr int v = S <= 16 ? (sizeof(T) == 8 ? GATO_SCHUR_WAVES_F64 : GATO_SCHUR_WAVES_F32) : 1; };
template <typename T, int S> struct GatherWaves { static constexpr int v = S <= 16 ? (sizeof(T) == 8 ? GATO_GATHER_WAVES_F64 : GATO_GATHER_WAVES_F32) : 1; };

template <typename T, int S, int C>
__global__ __launch_bounds__(WAVE, (AsmWaves<T, S>::v)) void schur_kernel(const T *__restrict__ Gd, const T *__restrict__ Ginv,
                                                     const T *__restrict__ Cd, const T *__restrict__ g,
                                                     const T *__restrict__ c, int K, T *__restrict__ Sbd,
                                                     T *__restrict__ Pbd, T *__restrict__ gamma, BatchStride bs)
{
    constexpr int n = S + C, SS = S * S, CC = C * C, SC = S * C;
    int sysi, kidx;
    xcd_knot_map(sysi, kidx);
    Gd += sysi * bs.g; Ginv += sysi * bs.g; Cd += sysi * bs.c; g += sysi * bs.n;
    c += sysi * bs.sk; Sbd += sysi * bs.bd; Pbd += sysi * bs.bd; gamma += sysi * bs.sk;
    // Only what one product hands to the next goes through LDS (phi, BR, theta); the operands that come from memory
    // (A, B, the inverses, q) are read from there by the lanes that feed them to the matrix cores / the FMAs: 4.5 KB of
    // LDS per wave instead of 10.4 KB, i.e. twice the knots in flight per CU - these launches are latency bound.
    // (S > 16: several MFMA tiles share every operand element, there the operands are staged in LDS as before.)
    constexpr bool STAGE = S > 16;
    __shared__ T sPhi[SS], sBR[SC], sTh[SS];
    __shared__ T sv[3 * S];
    __shared__ T stage[STAGE ? 3 * SS + SC + CC + 2 * S + C : 1];
    const int lane = threadIdx.x;
    for (int k = bs.k_lo + kidx; k < bs.k_hi; k += gridDim.x) {
        T *Sk = Sbd + (size_t)k * 3 * SS;
        T *Pk = Pbd + (size_t)k * 3 * SS;
        wave_sync();
        if (k == 0) {                                                        // :26-147
            const T *sQik = Ginv, *sq = g;
            mv<T, S, S>(sv, sQik, sq, lane);
            for (int i = lane; i < SS; i += WAVE) {
                Sk[i] = (T)0;                                                // S[0].left: unused (:157-165)
                Sk[SS + i] = -sQik[i];                                       // :120-126
                Pk[i] = (T)0;
                Pk[SS + i] = -Gd[i];                                         // :75-81
                Pk[2 * SS + i] = (T)0;
                if (K == 1) Sk[2 * SS + i] = (T)0;
            }
            wave_sync();
            for (int i = lane; i < S; i += WAVE) gamma[i] = c[i] - sv[i];    // :131-146, + c_0 (D4)
            continue;
        }
        const size_t gm = (size_t)(k - 1) * (SS + CC), gk = (size_t)k * (SS + CC);
        const size_t cm = (size_t)(k - 1) * (SS + SC);
        const T *sA = Cd + cm, *sB = Cd + cm + SS;                           // :189-196
        const T *sQim = Ginv + gm, *sRim = Ginv + gm + SS, *sQik = Ginv + gk;
        const T *sq = g + (size_t)(k - 1) * n;                               // q_{k-1}, r_{k-1}, and q_k n further (u_{k-1} is C long)
        if constexpr (STAGE) {
            T *a_ = stage, *b_ = a_ + SS, *qm_ = b_ + SC, *qk_ = qm_ + SS, *rm_ = qk_ + SS, *q_ = rm_ + CC;
            copy_in(a_, sA, SS, lane); copy_in(b_, sB, SC, lane); copy_in(qm_, sQim, SS, lane); copy_in(qk_, sQik, SS, lane);
            copy_in(rm_, sRim, CC, lane); copy_in(q_, sq, 2 * S + C, lane);
            sA = a_; sB = b_; sQim = qm_; sQik = qk_; sRim = rm_; sq = q_;
            wave_sync();
        }
        // phi = A Q_{k-1}^-1 (:277-285), BR = B R_{k-1}^-1 (:293-301): matrix cores, results to LDS and S[k].left
        mfma_for_tiles<T, S, S>(lane,
            [&](int mt, int nt, typename Mfma<T>::acc_t acc) { return mfma_tile<T, S, S, S, false>(sA, sQim, mt, nt, lane, acc); },
            [&](int r, int cc, T v) {
                sPhi[cc * S + r] = v;
                Sk[cc * S + r] = -v;                                         // S[k].left = -phi      :388-394
                Sk[2 * SS - 3 * SS + r * S + cc] = -v;                       // S[k-1].right = -phi^T :443-455
            });
        mfma_for_tiles<T, S, C>(lane,
            [&](int mt, int nt, typename Mfma<T>::acc_t acc) { return mfma_tile<T, S, C, C, false>(sB, sRim, mt, nt, lane, acc); },
            [&](int r, int cc, T v) { sBR[cc * S + r] = v; });
        mv<T, S, S>(sv, sQik, sq + n, lane);                                 // Q_k^-1 q_k           :306-310
        wave_sync();
        mv<T, S, S>(sv + S, sPhi, sq, lane);                                 // phi q_{k-1}          :316-320
        mv<T, S, C>(sv + 2 * S, sBR, sq + S, lane);                          // BR r_{k-1}           :324-328
        // theta = phi A^T + Q_k^-1 + BR B^T (:342-384): both products accumulate in the same MFMA tile
        mfma_for_tiles<T, S, S>(lane,
            [&](int mt, int nt, typename Mfma<T>::acc_t acc) {
                acc = mfma_tile<T, S, S, S, true>(sPhi, sA, mt, nt, lane, acc);
                return mfma_tile<T, S, C, S, true>(sBR, sB, mt, nt, lane, acc);
            },
            [&](int r, int cc, T v) {
                const T th = v + sQik[cc * S + r];
                sTh[cc * S + r] = th;
                Sk[SS + cc * S + r] = -th;                                   // S[k].main   :398-404
            });
        wave_sync();
        for (int i = lane; i < S; i += WAVE) {
            T gt = sv[i] - c[(size_t)k * S + i];                             // :311-313
            gt += sv[2 * S + i] + sv[S + i];                                 // :336-338
            gamma[(size_t)k * S + i] = -gt;                                  // :435-438
        }
        // block-Jacobi off-diagonals are zero; when the stair launch follows (bs.stair_follows) it writes Pinv[k].left and,
        // for k < K-1, Pinv[k].right itself, so only the last knot's right block is zeroed here
        if (!bs.stair_follows || k == K - 1) {
            for (int i = lane; i < SS; i += WAVE) {
                if (!bs.stair_follows) Pk[i] = (T)0;
                Pk[2 * SS + i] = (T)0;
                if (k == K - 1) Sk[2 * SS + i] = (T)0;                       // last right: unused (:166-174)
            }
        }
        invert_to<T, S>(sTh, Pk + SS, lane, (T)-1);                          // Pinv[k].main = -theta^-1  :407-422
    }
}

// ---- point-Jacobi preconditioner: the reference's build with BLOCK_J_PRECON = SS_PRECON = 0 (gato_defines.h:9-10;
// gato_schur.cuh:424-428): Pinv[k].main = diag(1 / S[k].main_ii), every other entry of the block row zero (the
// reference leaves them as cudaMalloc returned them; zero is what it means).
template <typename T, int S>
__global__ __launch_bounds__(256) void point_jacobi_kernel(const T *__restrict__ Sbd, T *__restrict__ Pbd, int K, BatchStride bs)
{
    constexpr int SS = S * S;
    Sbd += blockIdx.y * bs.bd; Pbd += blockIdx.y * bs.bd;
    const size_t total = (size_t)3 * SS * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % (3 * SS));
        const bool diag = w >= SS && w < 2 * SS && (w - SS) % (S + 1) == 0;
        Pbd[i] = diag ? (T)1 / Sbd[i] : (T)0;
    }
}

// ---- A3: symmetric stair (gato_schur.cuh:497-649) ------------------------------------------------
template <typename T, int S>
__global__ __launch_bounds__(WAVE) void ss_kernel(const T *__restrict__ Sbd, T *__restrict__ Pbd, int K, BatchStride bs)
{
    constexpr int SS = S * S;
    int sysi, kidx;
    xcd_knot_map(sysi, kidx);
    Sbd += sysi * bs.bd; Pbd += sysi * bs.bd;
    // operands straight from memory, only the intermediate products go through LDS (see schur_kernel); the left and the
    // right pair are independent chains: both first products, then both second products
    constexpr bool STAGE = S > 16;                                           // as in schur_kernel
    __shared__ T sT[2][SS];
    __shared__ T stage[STAGE ? 5 * SS : 1];
    const int lane = threadIdx.x;
    for (int k = bs.k_lo + kidx; k < bs.k_hi; k += gridDim.x) {
        T *Pk = Pbd + (size_t)k * 3 * SS;
        const T *sPm = Pk + SS;
        const bool has_l = k > 0, has_r = k < K - 1;                         // right pair for k < K-1 only (D1)
        const T *xl = Sbd + (size_t)k * 3 * SS, *xr = Sbd + (size_t)(k + 1) * 3 * SS;      // S[k].left, S[k+1].left
        const T *pl = Pbd + (size_t)(k - 1) * 3 * SS + SS, *pr = Pbd + (size_t)(k + 1) * 3 * SS + SS;   // Pinv[k-1].main, Pinv[k+1].main
        wave_sync();
        if constexpr (STAGE) {
            copy_in(stage, sPm, SS, lane);
            if (has_l) { copy_in(stage + SS, xl, SS, lane); copy_in(stage + 2 * SS, pl, SS, lane); }
            if (has_r) { copy_in(stage + 3 * SS, xr, SS, lane); copy_in(stage + 4 * SS, pr, SS, lane); }
            sPm = stage; xl = stage + SS; pl = stage + 2 * SS; xr = stage + 3 * SS; pr = stage + 4 * SS;
            wave_sync();
        }
        if (has_l) {                                                         // :578-611
            const T *sX = xl;
            mfma_for_tiles<T, S, S>(lane,
                [&](int mt, int nt, typename Mfma<T>::acc_t acc) { return mfma_tile<T, S, S, S, false>(sPm, sX, mt, nt, lane, acc); },
                [&](int r, int cc, T v) { sT[0][cc * S + r] = v; });
        }
        if (has_r) {                                                         // :614-648
            const T *sX = xr;                                                // used transposed
            mfma_for_tiles<T, S, S>(lane,
                [&](int mt, int nt, typename Mfma<T>::acc_t acc) { return mfma_tile<T, S, S, S, true>(sPm, sX, mt, nt, lane, acc); },
                [&](int r, int cc, T v) { sT[1][cc * S + r] = v; });
        }
        wave_sync();
        if (has_l) {
            const T *sPn = pl;
            mfma_for_tiles<T, S, S>(lane,
                [&](int mt, int nt, typename Mfma<T>::acc_t acc) { return mfma_tile<T, S, S, S, false>(sT[0], sPn, mt, nt, lane, acc); },
                [&](int r, int cc, T v) { Pk[cc * S + r] = -v; });
        }
        if (has_r) {
            const T *sPn = pr;
            mfma_for_tiles<T, S, S>(lane,
                [&](int mt, int nt, typename Mfma<T>::acc_t acc) { return mfma_tile<T, S, S, S, false>(sT[1], sPn, mt, nt, lane, acc); },
                [&](int r, int cc, T v) { Pk[2 * SS + cc * S + r] = -v; });
        }
    }
}

// ---- A1 + A2 + A3 in ONE launch ------------------------------------------------------------------------------
// A small solve is bound by dependent launches, not by work: memset, scatter, inversions, Schur and stair were five
// launches of ~5 us each in front of a PCG loop of ~230 us (IIWA 14/7/50, profiles/).  Here the workgroup of knot k
// produces everything the stage kernels write for knot k PLUS the two stair blocks that couple knots k-1 and k
// (Pinv[k].left, Pinv[k-1].right).  Those need theta_{k-1}^-1, which the workgroup RECOMPUTES from knot k-2's blocks
// instead of waiting for its neighbour: twice the arithmetic, no inter-workgroup hand-off (a flag hand-off across
// XCDs measured 5-8 us here, more than the whole recomputation), no co-residency requirement, any K.
//  1. gather the CSR entries of knots k-2..k (rows are contiguous, so the entries are one contiguous index range:
//     a thread per ENTRY, row found by bisection of the row pointers held in LDS - two dependent global loads deep,
//     coalesced) into dense LDS blocks; knot k's blocks are written out once (no memset);
//  2. invert Q_{k-2}, R_{k-2}, Q_{k-1}, R_{k-1}, Q_k, R_k on six wavefronts at the same time (register Gauss-Jordan);
//  3. phi/theta of knots k-1 and k with the MFMA tiles spread over the waves; gamma, S[k], Pinv[k].main;
//  4. the two stair blocks.
// Per block the instruction sequence is that of invert_G_kernel / schur_kernel / ss_kernel: results are bit-identical
// (tests/test_gpu_parity.py::test_fused_assembly_is_bit_identical_to_the_stage_kernels).
template <int NT, typename T>
__device__ __forceinline__ void copy_nt(T *dst, const T *__restrict__ src, int n, int tid)
{
    for (int i = tid; i < n; i += NT) dst[i] = src[i];
}

// MFMA tiles of an M x N result spread over the workgroup's waves: tile t runs on wave (first + t) % NW
template <typename T, int M, int N, int NW, typename TileFn, typename StoreFn>
__device__ __forceinline__ void mfma_tiles_on_waves(int wave, int lane, int first, TileFn tile, StoreFn store)
{
    constexpr int NTN = (N + 15) / 16;
#pragma unroll
    for (int mt = 0; mt < (M + 15) / 16; ++mt) {
#pragma unroll
        for (int nt = 0; nt < NTN; ++nt) {
            if ((first + mt * NTN + nt) % NW != wave) continue;                 // wave-uniform
            typename Mfma<T>::acc_t acc = {0, 0, 0, 0};
            acc = tile(mt, nt, acc);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = mt * 16 + Mfma<T>::row(lane, j), c = nt * 16 + (lane & 15);
                if (r < M && c < N) store(r, c, acc[j]);
            }
        }
    }
}

template <typename T, int S, int C>
struct AsmLds {                                   // element offsets into the dynamic LDS block
    static constexpr int n = S + C, SS = S * S, CC = C * C, SC = S * C;
    static constexpr int Q = 0;                   // 3 x SS   Q_{k-2}, Q_{k-1}, Q_k: raw, then inverted in place
    static constexpr int R = Q + 3 * SS;          // 3 x CC
    static constexpr int AB = R + 3 * CC;         // 2 x (SS + SC)   [A | B] of block rows k-2, k-1
    static constexpr int PHI = AB + 2 * (SS + SC);   // 2 x SS  phi of knots k-1, k
    static constexpr int BR = PHI + 2 * SS;       // 2 x SC
    static constexpr int TH = BR + 2 * SC;        // 2 x SS   theta, then -theta^-1 in place (Pinv main of knots k-1, k)
    static constexpr int TT = TH + 2 * SS;        // 2 x SS   stair temporaries
    static constexpr int VQ = TT + 2 * SS;        // q_{k-1}, r_{k-1}, q_k
    static constexpr int VV = VQ + 2 * S + C;     // 3 x S
    static constexpr int ELEMS = (VV + 3 * S + 3) / 4 * 4;
    static constexpr int PTR_G = 3 * n + 1, PTR_C = 2 * S + 1;
    static constexpr size_t BYTES = sizeof(T) * ELEMS + sizeof(int) * (PTR_G + PTR_C);
};

// largest i in [0, nrows) with ptr[i] <= e  (ptr[0] <= e < ptr[nrows])
__device__ __forceinline__ int row_of_entry(const int *ptr, int nrows, int e)
{
    int lo = 0, hi = nrows;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (ptr[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
}

// ---- A1 (+ the inversions of A2): one workgroup per knot ----------------------------------------------------------
// The reference walks each CSR row with one thread (gato_schur.cuh:674-743): a chain of dependent loads per entry and
// scattered 4-byte stores into pre-zeroed outputs.  Here the workgroup of knot k gathers the knot's entries - G rows
// of knot k and C row-block k+1 are contiguous index ranges - a thread per ENTRY, into dense LDS blocks and writes
// Q_k, R_k, [A_k | B_k] out whole: no memset, coalesced stores, the batch shares the index arrays.  With Ginv given,
// two wavefronts then invert Q_k and R_k from LDS (what invert_G_kernel does after a round trip through HBM).
// Same arithmetic and same destination per entry as the reference.
template <typename T, int S, int C, int NT>
__global__ __launch_bounds__(NT, (GatherWaves<T, S>::v)) void gather_kernel(const int *__restrict__ G_row, const int *__restrict__ G_col,
                                                    const T *__restrict__ G_val, const int *__restrict__ C_row,
                                                    const int *__restrict__ C_col, const T *__restrict__ C_val, int K, T rho,
                                                    T *__restrict__ Gd, T *__restrict__ Cd, T *__restrict__ Ginv, BatchStride bs)
{
    constexpr int n = S + C, SS = S * S, CC = C * C, SC = S * C, ABS = SS + SC;
    static_assert(NT % WAVE == 0, "whole waves");
    __shared__ T blk[SS + CC + ABS];                                          // Q_k | R_k | A_k | B_k
    __shared__ int sPtrG[n + 1], sPtrC[S + 1];
    // COLLISIONS: a CSR row may hold the same column twice (or two columns that fold onto one slot, col % n).  The reference walks a
    // row with ONE thread, so the LAST entry in storage order wins (gato_schur.cuh:689-702, :733-741); a thread per entry needs a
    // rule.  Fast path: every entry also sets its slot's bit (one LDS atomic), and only a workgroup that sees a bit set twice
    // resolves owners: atomicMax of the entry index per slot, then the winners store again.  Valid inputs never take that path.
    constexpr int NSLOT = SS + CC + ABS;
    __shared__ unsigned smask[(NSLOT + 31) / 32];
    __shared__ int sown[NSLOT];
    __shared__ int s_coll;
    T *sQ = blk, *sR = blk + SS, *sAB = blk + SS + CC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    G_val += blockIdx.y * bs.nnzG; C_val += blockIdx.y * bs.nnzC;            // batch: shared structure, own values
    Gd += blockIdx.y * bs.g; Cd += blockIdx.y * bs.c;
    if (Ginv) Ginv += blockIdx.y * bs.g;
    for (int k = bs.k_lo + blockIdx.x; k < bs.k_hi; k += gridDim.x) {
        const bool last = k == K - 1;
        const int r0 = k * n, nrG = last ? S : n;
        const int c0 = (k + 1) * S, nrC = last ? 0 : S;
        __syncthreads();
        for (int i = tid; i <= nrG; i += NT) sPtrG[i] = G_row[r0 + i];
        if (nrC) for (int i = tid; i <= nrC; i += NT) sPtrC[i] = C_row[c0 + i];
        for (int i = tid; i < SS + CC + ABS; i += NT) blk[i] = (T)0;
        for (int i = tid; i < (NSLOT + 31) / 32; i += NT) smask[i] = 0u;
        if (tid == 0) s_coll = 0;
        __syncthreads();
        const int eG0 = sPtrG[0], nG = sPtrG[nrG] - eG0;
        const int eC0 = nrC ? sPtrC[0] : 0, nC = nrC ? sPtrC[nrC] - eC0 : 0;
        // destination slot in blk of entry t (given its column and value), -1 if the entry is dropped; v = the value stored
        auto slot_of = [&](int t, int colv, T valv, T &v) -> int {
            if (t < nG) {                                                  // csr_to_custom_G, gato_schur.cuh:674-704
                const int isr = row_of_entry(sPtrG, nrG, eG0 + t);
                const int isc = colv % n;
                v = valv + (colv == r0 + isr ? rho : (T)0);
                if (isc < S) return isr < S ? isc * S + isr : -1;
                return isr >= S ? SS + (isc - S) * C + (isr - S) : -1;
            }
            const int i = row_of_entry(sPtrC, nrC, eC0 + (t - nG));        // csr_to_custom_C, :707-743
            v = valv;
            return colv / n <= k ? SS + CC + (colv % n) * S + i : -1;
        };
        constexpr int U = 4;
        for (int t0 = tid; t0 < nG + nC; t0 += NT * U) {
            int col[U];
            T val[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u * NT;
                if (t < nG) { col[u] = G_col[eG0 + t]; val[u] = G_val[eG0 + t]; }
                else if (t < nG + nC) { col[u] = C_col[eC0 + (t - nG)]; val[u] = C_val[eC0 + (t - nG)]; }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = t0 + u * NT;
                if (t < nG + nC) {
                    T v;
                    const int sl = slot_of(t, col[u], val[u], v);
                    if (sl >= 0) {
                        blk[sl] = v;
                        const unsigned bit = 1u << (sl & 31);
                        if (atomicOr(&smask[sl >> 5], bit) & bit) s_coll = 1;
                    }
                }
            }
        }
        __syncthreads();
        if (s_coll) {                                                      // workgroup-uniform; never taken for duplicate-free rows
            for (int i = tid; i < NSLOT; i += NT) sown[i] = -1;
            __syncthreads();
            for (int pass = 0; pass < 2; ++pass) {
                for (int t = tid; t < nG + nC; t += NT) {
                    const int colv = t < nG ? G_col[eG0 + t] : C_col[eC0 + (t - nG)];
                    const T valv = t < nG ? G_val[eG0 + t] : C_val[eC0 + (t - nG)];
                    T v;
                    const int sl = slot_of(t, colv, valv, v);
                    if (sl < 0) continue;
                    // (within a row storage order = index order; entries of different rows never share a slot)
                    if (pass == 0) atomicMax(&sown[sl], t);
                    else if (sown[sl] == t) blk[sl] = v;
                }
                __syncthreads();
            }
        }
        __syncthreads();
        const size_t gk = (size_t)k * (SS + CC);
        for (int i = tid; i < SS + (last ? 0 : CC); i += NT) Gd[gk + i] = blk[i];
        if (!last) for (int i = tid; i < ABS; i += NT) Cd[(size_t)k * ABS + i] = sAB[i];
        if (Ginv) {
            if (wave == 0) invert_to<T, S>(sQ, Ginv + gk, lane, (T)1);
            if (wave == (NT > WAVE ? 1 : 0) && !last) invert_to<T, C>(sR, Ginv + gk + SS, lane, (T)1);
        }
    }
}

template <typename T, int S, int C, int NT>
__global__ __launch_bounds__(NT) void assemble_kernel(AsmArgs a, int K, BatchStride bs)
{
    typedef AsmLds<T, S, C> L;
    constexpr int n = S + C, SS = S * S, CC = C * C, SC = S * C, NW = NT / WAVE, ABS = SS + SC;
    static_assert(NW >= 6, "six inversions run side by side");
    constexpr int TILES = ((S + 15) / 16) * ((S + 15) / 16);
    extern __shared__ __align__(16) unsigned char lds_raw[];
    T *lds = (T *)lds_raw;
    T *sQ = lds + L::Q, *sR = lds + L::R, *sAB = lds + L::AB, *sPhi = lds + L::PHI, *sBR = lds + L::BR;
    T *sTh = lds + L::TH, *sT = lds + L::TT, *sq = lds + L::VQ, *sv = lds + L::VV;
    int *sPtrG = (int *)(lds + L::ELEMS), *sPtrC = sPtrG + L::PTR_G;
    // collisions of the CSR scatter (see gather_kernel): slot bits, owners, flag
    constexpr int NSLOT = 3 * SS + 3 * CC + 2 * ABS;
    __shared__ unsigned smask[(NSLOT + 31) / 32];
    __shared__ int sown[NSLOT];
    __shared__ int s_coll;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sys = blockIdx.y;
    const T *g = (const T *)a.g + sys * bs.n, *c = (const T *)a.c + sys * bs.sk;
    T *Gd = (T *)a.Gd + sys * bs.g, *Cd = (T *)a.Cd + sys * bs.c, *Ginv = (T *)a.Ginv + sys * bs.g;
    T *Sbd = (T *)a.Sbd + sys * bs.bd, *Pbd = (T *)a.Pbd + sys * bs.bd, *gamma = (T *)a.gamma + sys * bs.sk;
    const T rho = (T)a.rho;
    // optional second copy of S and Pinv for the one-workgroup PCG kernels (PcgLaunch::imgS): entry (row r of knot kk, column
    // cc of block b) at img[(b S + cc) ld + kk S + r] - a lane of those kernels then loads its 3S entries with unit stride
    // across the wave.  One system only (the caller passes nullptr otherwise).
    T *const iS = (T *)a.imgS, *const iP = (T *)a.imgP;
    const size_t ild = (size_t)a.img_ld;
    auto imS = [&](int kk, int b, int r, int cc, T v) { if (iS) iS[(size_t)(b * S + cc) * ild + (size_t)kk * S + r] = v; };
    auto imP = [&](int kk, int b, int r, int cc, T v) { if (iP) iP[(size_t)(b * S + cc) * ild + (size_t)kk * S + r] = v; };
    // this knot's operands (sub-knot 1) and the previous knot's (sub-knot 0)
    T *sQk = sQ + 2 * SS, *sRk = sR + 2 * CC, *sQm = sQ + SS, *sRm = sR + CC, *sQmm = sQ, *sRmm = sR;
    T *sA1 = sAB + ABS, *sB1 = sA1 + SS, *sA0 = sAB, *sB0 = sAB + SS;
    T *sPhi1 = sPhi + SS, *sPhi0 = sPhi, *sBR1 = sBR + SC, *sBR0 = sBR, *sTh1 = sTh + SS, *sTh0 = sTh;
    int n_stamp = 0;
    auto stamp = [&]() {
        if (a.stamps && blockIdx.x == (K > 2 ? 2 : 0) && sys == 0 && tid == 0) a.stamps[n_stamp++] = __builtin_amdgcn_s_memrealtime();
    };
    stamp();

    for (int k = blockIdx.x; k < K; k += gridDim.x) {
        const bool first = k == 0, last = k == K - 1;
        const bool full0 = k >= 2;                                           // theta_{k-1} is a Schur block (k-1 >= 1)
        const size_t gk = (size_t)k * (SS + CC);
        const size_t cm = first ? 0 : (size_t)(k - 1) * ABS;
        T *Sk = Sbd + (size_t)k * 3 * SS, *Pk = Pbd + (size_t)k * 3 * SS;
        __syncthreads();
        // ---- 1. blocks of knots k-2 .. k into LDS ----
        if (a.mode == 0) {
            const T *G_val = (const T *)a.G_val + sys * bs.nnzG, *C_val = (const T *)a.C_val + sys * bs.nnzC;
            const int r0 = (k >= 2 ? k - 2 : 0) * n, nrG = k * n + (last ? S : n) - r0;
            const int c0 = (k >= 2 ? k - 1 : 1) * S, nrC = first ? 0 : (k + 1) * S - c0;
            for (int i = tid; i <= nrG; i += NT) sPtrG[i] = a.G_row[r0 + i];
            if (nrC) for (int i = tid; i <= nrC; i += NT) sPtrC[i] = a.C_row[c0 + i];
            for (int i = tid; i < 3 * SS + 3 * CC + 2 * ABS; i += NT) lds[i] = (T)0;   // sQ, sR, sAB are contiguous
            for (int i = tid; i < (NSLOT + 31) / 32; i += NT) smask[i] = 0u;
            if (tid == 0) s_coll = 0;
            __syncthreads();
            const int eG0 = sPtrG[0], nG = sPtrG[nrG] - eG0;
            const int eC0 = nrC ? sPtrC[0] : 0, nC = nrC ? sPtrC[nrC] - eC0 : 0;
            // destination slot of entry t in lds[0 .. NSLOT) (sQ | sR | sAB), -1 if the entry is dropped; v = the value stored
            auto slot_of = [&](int t, int colv, T valv, T &v) -> int {
                if (t < nG) {                                              // csr_to_custom_G, gato_schur.cuh:674-704
                    const int row = r0 + row_of_entry(sPtrG, nrG, eG0 + t);
                    const int knot = row / n, isr = row - knot * n, qi = knot - (k - 2);
                    const int isc = colv % n;
                    v = valv + (colv == row ? rho : (T)0);
                    if (isc < S) return isr < S ? L::Q + qi * SS + isc * S + isr : -1;
                    return isr >= S ? L::R + qi * CC + (isc - S) * C + (isr - S) : -1;
                }
                const int row = c0 + row_of_entry(sPtrC, nrC, eC0 + (t - nG));   // csr_to_custom_C, :707-743
                const int br = row / S - 1, i = row - (br + 1) * S;
                v = valv;
                return colv / n <= br ? L::AB + (br - (k - 2)) * ABS + (colv % n) * S + i : -1;
            };
            // U entries per thread and round: all 2U loads are in flight before the first bisection result is needed
            constexpr int U = 4;
            for (int t0 = tid; t0 < nG + nC; t0 += NT * U) {
                int col[U];
                T val[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int t = t0 + u * NT;
                    if (t < nG) { col[u] = a.G_col[eG0 + t]; val[u] = G_val[eG0 + t]; }
                    else if (t < nG + nC) { col[u] = a.C_col[eC0 + (t - nG)]; val[u] = C_val[eC0 + (t - nG)]; }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int t = t0 + u * NT;
                    if (t < nG + nC) {
                        T v;
                        const int sl = slot_of(t, col[u], val[u], v);
                        if (sl >= 0) {
                            lds[sl] = v;
                            const unsigned bit = 1u << (sl & 31);
                            if (atomicOr(&smask[sl >> 5], bit) & bit) s_coll = 1;
                        }
                    }
                }
            }
            __syncthreads();
            if (s_coll) {                                                  // a row holds a column twice: the last entry in storage order wins
                for (int i = tid; i < NSLOT; i += NT) sown[i] = -1;
                __syncthreads();
                for (int pass = 0; pass < 2; ++pass) {
                    for (int t = tid; t < nG + nC; t += NT) {
                        const int colv = t < nG ? a.G_col[eG0 + t] : a.C_col[eC0 + (t - nG)];
                        const T valv = t < nG ? G_val[eG0 + t] : C_val[eC0 + (t - nG)];
                        T v;
                        const int sl = slot_of(t, colv, valv, v);
                        if (sl < 0) continue;
                        if (pass == 0) atomicMax(&sown[sl], t);
                        else if (sown[sl] == t) lds[sl] = v;
                    }
                    __syncthreads();
                }
            }
        } else {
            const T *Gin = a.mode == 2 ? (const T *)a.G_val + sys * bs.g : Gd;
            const T r2 = a.mode == 2 ? rho : (T)0;
#pragma unroll
            for (int qi = 0; qi < 3; ++qi) {
                const int knot = k - 2 + qi;
                if (knot < 0) continue;
                const size_t go = (size_t)knot * (SS + CC);
                for (int i = tid; i < SS; i += NT) sQ[qi * SS + i] = Gin[go + i] + ((i % (S + 1) == 0) ? r2 : (T)0);
                if (knot < K - 1)
                    for (int i = tid; i < CC; i += NT) sR[qi * CC + i] = Gin[go + SS + i] + ((i % (C + 1) == 0) ? r2 : (T)0);
                if (qi < 2 && knot < K - 1) copy_nt<NT>(sAB + qi * ABS, Cd + (size_t)knot * ABS, ABS, tid);
            }
        }
        if (first) { for (int i = tid; i < S; i += NT) sq[i] = g[i]; }
        else {
            copy_nt<NT>(sq, g + (size_t)(k - 1) * n, n, tid);                // q_{k-1}, r_{k-1}
            for (int i = tid; i < S; i += NT) sq[n + i] = g[(size_t)k * n + i];   // q_k
        }
        __syncthreads();
        stamp();                                                             // 1: gathered
        if (a.mode != 1) {                                                   // knot k's dense blocks, written once
            for (int i = tid; i < SS; i += NT) Gd[gk + i] = sQk[i];
            if (!last) for (int i = tid; i < CC; i += NT) Gd[gk + SS + i] = sRk[i];
            if (a.mode == 0 && !first) for (int i = tid; i < ABS; i += NT) Cd[cm + i] = sA1[i];
        }
        if (first) for (int i = tid; i < SS; i += NT) { Pk[SS + i] = -sQk[i]; imP(0, 1, i % S, i / S, -sQk[i]); }  // Pinv[0].main = -Q_0   :75-81
        if (k == 1) for (int i = tid; i < SS; i += NT) sTh0[i] = -sQm[i];    // ... which is this knot's left neighbour
        __syncthreads();
        // ---- 2. six inversions side by side, in place ----
        if (wave == 0) invert_to<T, S>(sQk, sQk, lane, (T)1);
        else if (wave == 1) { if (!first) invert_to<T, S>(sQm, sQm, lane, (T)1); }
        else if (wave == 2) { if (!first) invert_to<T, C>(sRm, sRm, lane, (T)1); }
        else if (wave == 3) { if (full0) invert_to<T, S>(sQmm, sQmm, lane, (T)1); }
        else if (wave == 4) { if (full0) invert_to<T, C>(sRmm, sRmm, lane, (T)1); }
        else if (wave == 5) { if (!last) invert_to<T, C>(sRk, sRk, lane, (T)1); }
        __syncthreads();
        stamp();                                                             // 2: inverted
        for (int i = tid; i < SS; i += NT) Ginv[gk + i] = sQk[i];
        if (!last) for (int i = tid; i < CC; i += NT) Ginv[gk + SS + i] = sRk[i];
        if (first) {                                                         // :26-147
            if (wave == 0) mv<T, S, S>(sv, sQk, sq, lane);
            for (int i = tid; i < SS; i += NT) {
                Sk[i] = (T)0;                                                // S[0].left: unused (:157-165)
                Sk[SS + i] = -sQk[i];                                        // :120-126
                Pk[i] = (T)0;
                if (last) { Pk[2 * SS + i] = (T)0; Sk[2 * SS + i] = (T)0; }
                imS(0, 0, i % S, i / S, (T)0); imS(0, 1, i % S, i / S, -sQk[i]); imP(0, 0, i % S, i / S, (T)0);
                if (last) { imP(0, 2, i % S, i / S, (T)0); imS(0, 2, i % S, i / S, (T)0); }
            }
            __syncthreads();
            for (int i = tid; i < S; i += NT) gamma[i] = c[i] - sv[i];       // :131-146, + c_0 (D4)
            continue;
        }
        // ---- 3. Schur blocks of knot k, theta of knot k-1 ----
        // phi = A Q^-1 (:277-285), BR = B R^-1 (:293-301)
        mfma_tiles_on_waves<T, S, S, NW>(wave, lane, 0,
            [&](int mt, int nt, typename Mfma<T>::acc_t acc) { return mfma_tile<T, S, S, S, false>(sA1, sQm, mt, nt, lane, acc); },
            [&](int r, int cc, T v) {
                sPhi1[cc * S + r] = v;
                Sk[cc * S + r] = -v;                                         // S[k].left = -phi      :388-394
                Sk[2 * SS - 3 * SS + r * S + cc] = -v;                       // S[k-1].right = -phi^T :443-455
                imS(k, 0, r, cc, -v); imS(k - 1, 2, cc, r, -v);
            });
        mfma_tiles_on_waves<T, S, C, NW>(wave, lane, TILES,
            [&](int mt, int nt, typename Mfma<T>::acc_t acc) { return mfma_tile<T, S, C, C, false>(sB1, sRm, mt, nt, lane, acc); },
            [&](int r, int cc, T v) { sBR1[cc * S + r] = v; });
        if (full0) {
            mfma_tiles_on_waves<T, S, S, NW>(wave, lane, 2 * TILES,
                [&](int mt, int nt, typename Mfma<T>::acc_t acc) { return mfma_tile<T, S, S, S, false>(sA0, sQmm, mt, nt, lane, acc); },
                [&](int r, int cc, T v) { sPhi0[cc * S + r] = v; });
            mfma_tiles_on_waves<T, S, C, NW>(wave, lane, 3 * TILES,
                [&](int mt, int nt, typename Mfma<T>::acc_t acc) { return mfma_tile<T, S, C, C, false>(sB0, sRmm, mt, nt, lane, acc); },
                [&](int r, int cc, T v) { sBR0[cc * S + r] = v; });
        }
        if (wave == NW - 1) mv<T, S, S>(sv, sQk, sq + n, lane);              // Q_k^-1 q_k           :306-310
        __syncthreads();
        if (wave == NW - 1) mv<T, S, S>(sv + S, sPhi1, sq, lane);            // phi q_{k-1}          :316-320
        if (wave == NW - 2) mv<T, S, C>(sv + 2 * S, sBR1, sq + S, lane);     // BR r_{k-1}           :324-328
        // theta = phi A^T + Q^-1 + BR B^T (:342-384)
        mfma_tiles_on_waves<T, S, S, NW>(wave, lane, 0,
            [&](int mt, int nt, typename Mfma<T>::acc_t acc) {
                acc = mfma_tile<T, S, S, S, true>(sPhi1, sA1, mt, nt, lane, acc);
                return mfma_tile<T, S, C, S, true>(sBR1, sB1, mt, nt, lane, acc);
            },
            [&](int r, int cc, T v) {
                const T th = v + sQk[cc * S + r];
                sTh1[cc * S + r] = th;
                Sk[SS + cc * S + r] = -th;                                   // S[k].main   :398-404
                imS(k, 1, r, cc, -th);
            });
        if (full0)
            mfma_tiles_on_waves<T, S, S, NW>(wave, lane, TILES,
                [&](int mt, int nt, typename Mfma<T>::acc_t acc) {
                    acc = mfma_tile<T, S, S, S, true>(sPhi0, sA0, mt, nt, lane, acc);
                    return mfma_tile<T, S, C, S, true>(sBR0, sB0, mt, nt, lane, acc);
                },
                [&](int r, int cc, T v) { sTh0[cc * S + r] = v + sQm[cc * S + r]; });
        __syncthreads();
        if (wave == 0) invert_to<T, S>(sTh1, sTh1, lane, (T)-1);             // Pinv[k].main = -theta^-1  :407-422
        else if (wave == 1) { if (full0) invert_to<T, S>(sTh0, sTh0, lane, (T)-1); }
        else {
            for (int i = tid - 2 * WAVE; i < S; i += NT - 2 * WAVE) {
                T gt = sv[i] - c[(size_t)k * S + i];                         // :311-313
                gt += sv[2 * S + i] + sv[S + i];                             // :336-338
                gamma[(size_t)k * S + i] = -gt;                              // :435-438
            }
            if (last)
                for (int i = tid - 2 * WAVE; i < SS; i += NT - 2 * WAVE) {
                    Pk[2 * SS + i] = (T)0;
                    Sk[2 * SS + i] = (T)0;                                   // last right: unused (:166-174)
                    imP(k, 2, i % S, i / S, (T)0); imS(k, 2, i % S, i / S, (T)0);
                }
        }
        __syncthreads();
        stamp();                                                             // 3: Schur blocks done
        for (int i = tid; i < SS; i += NT) { Pk[SS + i] = sTh1[i]; imP(k, 1, i % S, i / S, sTh1[i]); }
        // ---- 4. symmetric stair between knots k-1 and k (gato_schur.cuh:497-649; S[k].left = -phi) ----
        mfma_tiles_on_waves<T, S, S, NW>(wave, lane, 0,                      // Pinv[k].main * phi
            [&](int mt, int nt, typename Mfma<T>::acc_t acc) { return mfma_tile<T, S, S, S, false>(sTh1, sPhi1, mt, nt, lane, acc); },
            [&](int r, int cc, T v) { sT[cc * S + r] = v; });
        mfma_tiles_on_waves<T, S, S, NW>(wave, lane, TILES,                  // Pinv[k-1].main * phi^T
            [&](int mt, int nt, typename Mfma<T>::acc_t acc) { return mfma_tile<T, S, S, S, true>(sTh0, sPhi1, mt, nt, lane, acc); },
            [&](int r, int cc, T v) { sT[SS + cc * S + r] = v; });
        __syncthreads();
        mfma_tiles_on_waves<T, S, S, NW>(wave, lane, 0,                      // Pinv[k].left          :578-611
            [&](int mt, int nt, typename Mfma<T>::acc_t acc) { return mfma_tile<T, S, S, S, false>(sT, sTh0, mt, nt, lane, acc); },
            [&](int r, int cc, T v) { Pk[cc * S + r] = v; imP(k, 0, r, cc, v); });
        mfma_tiles_on_waves<T, S, S, NW>(wave, lane, TILES,                  // Pinv[k-1].right       :614-648 (D1)
            [&](int mt, int nt, typename Mfma<T>::acc_t acc) { return mfma_tile<T, S, S, S, false>(sT + SS, sTh1, mt, nt, lane, acc); },
            [&](int r, int cc, T v) { Pk[2 * SS - 3 * SS + cc * S + r] = v; imP(k - 1, 2, r, cc, v); });
        stamp();                                                             // 4: stair written
    }
}

// (Rounds 3-4 also had a CHUNKED fused launch here - one workgroup per chunk of consecutive knots, everything between the stages in
// LDS, option asm_mode = 3.  It moved less than half the bytes of the stage path and took the SAME time on every shape measured
// (512 x 14/7/50 f64: 270 us against 272, 14/7/4096 f32 42 / 41, 32/16/1024 f32 209 / 78): neither is bound by bytes but by how
// many knots a SIMD has in flight.  Removed in round 5; DESIGN_LOG.md 3.3 keeps the numbers.)

// ---- A9: dz back-substitution (gato_schur.cuh:758-867) -------------------------------------------
template <typename T, int S, int C>
__global__ __launch_bounds__(WAVE) void dz_kernel(const T *__restrict__ Ginv, const T *__restrict__ Cd,
                                                  const T *__restrict__ g, const T *__restrict__ lambda,
                                                  int K, T *__restrict__ dz, BatchStride bs)
{
    constexpr int n = S + C, SS = S * S, CC = C * C, SC = S * C;
    int sysi, kidx;
    xcd_knot_map(sysi, kidx);
    Ginv += sysi * bs.g; Cd += sysi * bs.c; g += sysi * bs.n; lambda += sysi * bs.sk;
    dz += sysi * bs.n;
    __shared__ T sQi[SS], sA[SS], sRi[CC > 0 ? CC : 1], sB[SC > 0 ? SC : 1], sl[2 * S], st[S + C], sg[S + C];
    const int lane = threadIdx.x;
    for (int k = bs.k_lo + kidx; k < bs.k_hi; k += gridDim.x) {
        const bool last = k == K - 1;
        wave_sync();
        copy_in(sQi, Ginv + (size_t)k * (SS + CC), SS, lane);
        copy_in(sl, lambda + (size_t)k * S, last ? S : 2 * S, lane);
        copy_in(sg, g + (size_t)k * n, last ? S : n, lane);
        if (!last) {
            copy_in(sRi, Ginv + (size_t)k * (SS + CC) + SS, CC, lane);
            copy_in(sA, Cd + (size_t)k * (SS + SC), SS, lane);
            copy_in(sB, Cd + (size_t)k * (SS + SC) + SS, SC, lane);
        }
        wave_sync();
        if (!last) {
            mTv<T, S, S>(st, sA, sl + S, lane);                              // A_k^T lambda_{k+1}   :833-838
            mTv<T, S, C>(st + S, sB, sl + S, lane);                          // B_k^T lambda_{k+1}   :784-789
            wave_sync();
            for (int i = lane; i < S; i += WAVE) st[i] = sg[i] - (sl[i] + st[i]);        // :841-852
            for (int i = lane; i < C; i += WAVE) st[S + i] = sg[S + i] - st[S + i];      // :792-796
        } else {
            for (int i = lane; i < S; i += WAVE) st[i] = sg[i] - sl[i];      // last state row (D2)
        }
        wave_sync();
        for (int r = lane; r < S; r += WAVE) {                               // Q_k^-1 (...)         :856-865
            T res = (T)0;
#pragma unroll
            for (int cc = 0; cc < S; ++cc) res = gato::fmaT(sQi[r + cc * S], st[cc], res);
            dz[(size_t)k * n + r] = res;
        }
        if (!last) {
            for (int r = lane; r < C; r += WAVE) {                           // R_k^-1 (...)         :799-808
                T res = (T)0;
#pragma unroll
                for (int cc = 0; cc < C; ++cc) res = gato::fmaT(sRi[r + cc * C], st[S + cc], res);
                dz[(size_t)k * n + S + r] = res;
            }
        }
    }
}

}  // namespace

template <typename T, int S, int C>
int launch_convert(const Dims &d, const int *G_row, const int *G_col, const T *G_val, const int *C_row,
                   const int *C_col, const T *C_val, T rho, T *Gd, T *Cd, T *Ginv, hipStream_t st)
{
    // one wave per knot for the small shapes: the launch is latency bound (two dependent global round trips per knot) and
    // registers cap the waves per CU, so one-wave workgroups put twice the knots in flight (batch of 25 600 knots in f64:
    // 118 -> 105 us); the 32 x 32 shapes have 1 568 entries per knot and keep two waves
    constexpr int NT = S <= 16 ? 64 : 128;
    hipLaunchKernelGGL((gather_kernel<T, S, C, NT>), dim3(d.hi() - d.lo() < (1 << 20) ? d.hi() - d.lo() : (1 << 20), d.B), dim3(NT), 0, st, G_row, G_col,
                       G_val, C_row, C_col, C_val, d.K, rho, Gd, Cd, Ginv, batch_stride(d));
    GATO_HIP_CHECK(hipGetLastError());
    return GATO_OK;
}

template <typename T, int S, int C>
int launch_add_rho(const Dims &d, const T *G_in, T rho, T *Gd, hipStream_t st)
{
    const size_t total = d.g_dense() * d.B;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL((add_rho_kernel<T, S, C>), dim3(blocks), dim3(256), 0, st, G_in, rho, Gd, d.g_dense(), d.B);
    GATO_HIP_CHECK(hipGetLastError());
    return GATO_OK;
}

static inline int knot_grid(int K) { return K < 8192 ? K : 8192; }

template <typename T, int S, int C>
int launch_form_schur(const Dims &d, const T *Gd, const T *Cd, const T *g, const T *c, T *Sbd, T *Pbd,
                      T *gamma, T *Ginv, bool have_inverses, hipStream_t st)
{
    BatchStride bs = batch_stride(d);
    bs.stair_follows = d.stair_follows;
    if (!have_inverses) {
        const int nblk = (2 * (d.hi() - d.lo()) + 3) / 4;
        hipLaunchKernelGGL((invert_G_kernel<T, S, C>), dim3(nblk < 8192 ? nblk : 8192, d.B), dim3(256), 0, st, Gd, Ginv, d.K,
                           batch_stride(d));
        GATO_HIP_CHECK(hipGetLastError());
    }
    hipLaunchKernelGGL((schur_kernel<T, S, C>), dim3(knot_grid(d.hi() - d.lo()), d.B), dim3(WAVE), 0, st, Gd, Ginv, Cd, g, c, d.K,
                       Sbd, Pbd, gamma, bs);
    GATO_HIP_CHECK(hipGetLastError());
    return GATO_OK;
}

template <typename T, int S, int C>
int launch_assemble(const Dims &d, const AsmArgs &a, hipStream_t st)
{
    constexpr int NT = 512;
    typedef AsmLds<T, S, C> L;
    if (L::BYTES > 48 * 1024) {                      // beyond the default dynamic-LDS limit: opt in once per device
        static bool attr_set[64] = {};
        int dev = 0;
        GATO_HIP_CHECK(hipGetDevice(&dev));
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            GATO_HIP_CHECK(hipFuncSetAttribute((const void *)assemble_kernel<T, S, C, NT>,
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::BYTES));
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
    }
    const int gx = d.K < (1 << 20) ? d.K : (1 << 20);
    hipLaunchKernelGGL((assemble_kernel<T, S, C, NT>), dim3(gx, d.B), dim3(NT), L::BYTES, st, a, d.K, batch_stride(d));
    GATO_HIP_CHECK(hipGetLastError());
    return GATO_OK;
}

template <typename T, int S, int C>
int launch_form_ss(const Dims &d, const T *Sbd, T *Pbd, hipStream_t st)
{
    hipLaunchKernelGGL((ss_kernel<T, S>), dim3(knot_grid(d.hi() - d.lo()), d.B), dim3(WAVE), 0, st, Sbd, Pbd, d.K, batch_stride(d));
    GATO_HIP_CHECK(hipGetLastError());
    return GATO_OK;
}

template <typename T, int S, int C>
int launch_point_jacobi(const Dims &d, const T *Sbd, T *Pbd, hipStream_t st)
{
    const size_t total = d.bd();
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL((point_jacobi_kernel<T, S>), dim3(blocks, d.B), dim3(256), 0, st, Sbd, Pbd, d.K, batch_stride(d));
    GATO_HIP_CHECK(hipGetLastError());
    return GATO_OK;
}

template <typename T, int S, int C>
int launch_compute_dz(const Dims &d, const T *Ginv, const T *Cd, const T *g, const T *lambda, T *dz,
                      hipStream_t st)
{
    hipLaunchKernelGGL((dz_kernel<T, S, C>), dim3(knot_grid(d.hi() - d.lo()), d.B), dim3(WAVE), 0, st, Ginv, Cd, g, lambda, d.K, dz,
                       batch_stride(d));
    GATO_HIP_CHECK(hipGetLastError());
    return GATO_OK;
}

#define X(S_, C_)                                                                                              \
    template int launch_convert<float, S_, C_>(const Dims &, const int *, const int *, const float *, const int *, \
                                               const int *, const float *, float, float *, float *, float *, hipStream_t); \
    template int launch_convert<double, S_, C_>(const Dims &, const int *, const int *, const double *,           \
                                                const int *, const int *, const double *, double, double *,      \
                                                double *, double *, hipStream_t);                                \
    template int launch_add_rho<float, S_, C_>(const Dims &, const float *, float, float *, hipStream_t);             \
    template int launch_add_rho<double, S_, C_>(const Dims &, const double *, double, double *, hipStream_t);          \
    template int launch_form_schur<float, S_, C_>(const Dims &, const float *, const float *, const float *,     \
                                                  const float *, float *, float *, float *, float *, bool, hipStream_t); \
    template int launch_form_schur<double, S_, C_>(const Dims &, const double *, const double *, const double *, \
                                                   const double *, double *, double *, double *, double *, bool, \
                                                   hipStream_t);                                                 \
    template int launch_assemble<float, S_, C_>(const Dims &, const AsmArgs &, hipStream_t);                      \
    template int launch_assemble<double, S_, C_>(const Dims &, const AsmArgs &, hipStream_t);                     \
    template int launch_form_ss<float, S_, C_>(const Dims &, const float *, float *, hipStream_t);               \
    template int launch_point_jacobi<float, S_, C_>(const Dims &, const float *, float *, hipStream_t);          \
    template int launch_point_jacobi<double, S_, C_>(const Dims &, const double *, double *, hipStream_t);       \
    template int launch_form_ss<double, S_, C_>(const Dims &, const double *, double *, hipStream_t);            \
    template int launch_compute_dz<float, S_, C_>(const Dims &, const float *, const float *, const float *,     \
                                                  const float *, float *, hipStream_t);                          \
    template int launch_compute_dz<double, S_, C_>(const Dims &, const double *, const double *, const double *, \
                                                   const double *, double *, hipStream_t);
GATO_SHAPES(X)
#undef X

}  // namespace gato
