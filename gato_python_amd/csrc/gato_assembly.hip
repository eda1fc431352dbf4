// Assembly stages of the gato hot path for gfx950: CSR -> dense scatter (A1), Schur complement +
// block-Jacobi blocks + gamma (A2), symmetric-stair off-diagonals (A3), dz back-substitution (A9).
//
// Replaces gato_convert_kkt_format / csr_to_custom_G / csr_to_custom_C (src/gato_schur.cuh:674-756),
// gato_form_schur_jacobi[_inner] (:13-494), gato_form_ss[_inner] (:497-670), gato_compute_dz[_inner]
// (:758-879) and the block helpers they call (src/gato_utils.cuh: invertMatrix :468-586,
// mat_mat_prod :609-659, mat_vec_prod :595-606, gato_ATx :664-679, store/load_block_bd :44-119).
//
// Design: one wavefront (64 lanes) per knot, operands staged in LDS, every Q_k / R_k inverted exactly
// once into a separate inverse buffer (the reference inverts each Q_k twice and overwrites G_dense in
// place while neighbouring blocks still read it - SURVEY.md D3).  Gauss-Jordan without pivoting in the
// reference's elimination order, so fp32 results track the CUDA path.  Boundary fixes D1, D2, D4.
#include "gato_common.h"

namespace gato {
namespace {

constexpr int WAVE = 64;

// ---- A1 ------------------------------------------------------------------------------------
// The reference walks each CSR row with one thread (gato_schur.cuh:674-743): a chain of dependent loads per
// entry.  Here a thread owns ONE (row, slot) pair - slot = position inside the row, rows longer than SLOTS are
// walked with stride SLOTS - so a solve's scatter is two dependent loads deep (indptr -> col/val -> store).
// Same arithmetic and same destination per entry as the reference; outputs pre-zeroed.
template <typename T, int S, int C>
__global__ void convert_kernel(const int *__restrict__ G_row, const int *__restrict__ G_col,
                               const T *__restrict__ G_val, const int *__restrict__ C_row,
                               const int *__restrict__ C_col, const T *__restrict__ C_val, int K, T rho,
                               T *__restrict__ Gd, T *__restrict__ Cd, BatchStride bs)
{
    constexpr int n = S + C, SS = S * S, CC = C * C, SC = S * C;
    constexpr int SLOTS = 32;
    G_val += blockIdx.y * bs.nnzG; C_val += blockIdx.y * bs.nnzC;      // batch: shared structure, own values
    Gd += blockIdx.y * bs.g; Cd += blockIdx.y * bs.c;                                   // >= S + C + 1 for the compiled shapes' C rows
    const int N = n * K - C;
    const int SK = S * K;
    const long long total = (long long)(N + SK) * SLOTS;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int rr = (int)(t / SLOTS), slot = (int)(t % SLOTS);
        if (rr < N) {                                           // csr_to_custom_G, gato_schur.cuh:674-704
            const int row = rr;
            const int in_set_row = row % n;
            const size_t set_offset = (size_t)(row / n) * (SS + CC);
            const int end = G_row[row + 1];
            for (int it = G_row[row] + slot; it < end; it += SLOTS) {
                const int col = G_col[it];
                const int in_set_col = col % n;
                const T v = G_val[it] + (col == row ? rho : (T)0);
                if (in_set_col < S) Gd[set_offset + in_set_col * S + in_set_row] = v;
                else Gd[set_offset + SS + (in_set_col - S) * C + (in_set_row - S)] = v;
            }
        } else {                                                // csr_to_custom_C, :707-743
            const int row = rr - N;
            if (row < S) continue;
            const int block_row = row / S - 1;
            const int end = C_row[row + 1];
            for (int it = C_row[row] + slot; it < end; it += SLOTS) {
                const int col = C_col[it];
                if (col / n > block_row) continue;
                Cd[(size_t)block_row * (SS + SC) + (col % n) * S + row % S] = C_val[it];
            }
        }
    }
}

// ---- block helpers on LDS operands, executed by one wavefront ---------------------------------
__device__ __forceinline__ void wave_sync() { __syncthreads(); }   // blockDim == 64: one wave

template <typename T>
__device__ __forceinline__ void copy_in(T *dst, const T *__restrict__ src, int n, int lane)
{
    for (int i = lane; i < n; i += WAVE) dst[i] = src[i];
}

// Gauss-Jordan inverse, no pivoting (invertMatrix, gato_utils.cuh:468-586): A (n x n col-major in
// LDS) is destroyed, Ainv receives A^-1.  tmp: 3n elements.
template <typename T, int n>
__device__ void gj_inverse(T *A, T *Ainv, T *tmp, int lane)
{
    for (int i = lane; i < n * n; i += WAVE) Ainv[i] = (T)((i % n) == (i / n));
    wave_sync();
    T *colv = tmp, *rowA = tmp + n, *rowI = tmp + 2 * n;
    for (int p = 0; p < n; ++p) {
        // one reciprocal per pivot (the single-matrix overload's pvInv, gato_utils.cuh:476); the pivot row is
        // saved already scaled, the pivot column already multiplied by 1/pv
        const T pvinv = (T)1 / A[p + p * n];
        for (int i = lane; i < n; i += WAVE) {
            colv[i] = A[i + p * n] * pvinv;
            rowA[i] = A[p + i * n];
            rowI[i] = Ainv[p + i * n];
        }
        wave_sync();
        for (int e = lane; e < n * n; e += WAVE) {
            const int r = e % n, c = e / n;
            if (r == p) {
                A[e] = rowA[c] * pvinv;
                Ainv[e] = rowI[c] * pvinv;
            } else {
                const T f = colv[r];
                A[e] -= f * rowA[c];
                Ainv[e] -= f * rowI[c];
            }
        }
        wave_sync();
    }
}

// out(m x n) = A(m x k) B(k x n) (mat_mat_prod, gato_utils.cuh:609-633); TB: B given as (n x k), use B^T.
template <typename T, int m, int k, int n, bool TB>
__device__ __forceinline__ void mm(T *out, const T *A, const T *B, int lane)
{
    for (int e = lane; e < m * n; e += WAVE) {
        const int r = e % m, c = e / m;
        T res = (T)0;
#pragma unroll
        for (int t = 0; t < k; ++t) res += A[t * m + r] * (TB ? B[t * n + c] : B[c * k + t]);
        out[e] = res;
    }
}
// out(m) = A(m x n) x (mat_vec_prod, :595-606)
template <typename T, int m, int n>
__device__ __forceinline__ void mv(T *out, const T *A, const T *x, int lane)
{
    for (int r = lane; r < m; r += WAVE) {
        T res = (T)0;
#pragma unroll
        for (int c = 0; c < n; ++c) res += A[r + c * m] * x[c];
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
        for (int t = 0; t < m; ++t) res += A[i * m + t] * x[t];
        out[i] = res;
    }
}

// ---- A2a: invert every Q_k and R_k once -------------------------------------------------------
template <typename T, int S, int C>
__global__ __launch_bounds__(WAVE) void invert_G_kernel(const T *__restrict__ Gd, T *__restrict__ Ginv, int K,
                                                        BatchStride bs)
{
    constexpr int SS = S * S, CC = C * C;
    Gd += blockIdx.y * bs.g; Ginv += blockIdx.y * bs.g;
    __shared__ T A[SS], Ai[SS], tmp[3 * S];
    const int lane = threadIdx.x;
    for (int k = blockIdx.x; k < K; k += gridDim.x) {
        const size_t off = (size_t)k * (SS + CC);
        wave_sync();
        copy_in(A, Gd + off, SS, lane);
        wave_sync();
        gj_inverse<T, S>(A, Ai, tmp, lane);
        for (int i = lane; i < SS; i += WAVE) Ginv[off + i] = Ai[i];
        if (k < K - 1) {
            wave_sync();
            copy_in(A, Gd + off + SS, CC, lane);
            wave_sync();
            gj_inverse<T, C>(A, Ai, tmp, lane);
            for (int i = lane; i < CC; i += WAVE) Ginv[off + SS + i] = Ai[i];
        }
    }
}

// ---- A2b: Schur blocks, block-Jacobi main blocks, gamma (gato_schur.cuh:13-460) -----------------
template <typename T, int S, int C>
__global__ __launch_bounds__(WAVE) void schur_kernel(const T *__restrict__ Gd, const T *__restrict__ Ginv,
                                                     const T *__restrict__ Cd, const T *__restrict__ g,
                                                     const T *__restrict__ c, int K, T *__restrict__ Sbd,
                                                     T *__restrict__ Pbd, T *__restrict__ gamma, BatchStride bs)
{
    constexpr int n = S + C, SS = S * S, CC = C * C, SC = S * C;
    Gd += blockIdx.y * bs.g; Ginv += blockIdx.y * bs.g; Cd += blockIdx.y * bs.c; g += blockIdx.y * bs.n;
    c += blockIdx.y * bs.sk; Sbd += blockIdx.y * bs.bd; Pbd += blockIdx.y * bs.bd; gamma += blockIdx.y * bs.sk;
    __shared__ T sA[SS], sB[SC], sQim[SS], sQik[SS], sRim[CC], sPhi[SS], sBR[SC], sTh[SS], sTmp[SS];
    __shared__ T sq[2 * S + C], sv[3 * S], stmp[3 * S];
    const int lane = threadIdx.x;
    for (int k = blockIdx.x; k < K; k += gridDim.x) {
        T *Sk = Sbd + (size_t)k * 3 * SS;
        T *Pk = Pbd + (size_t)k * 3 * SS;
        wave_sync();
        if (k == 0) {                                                        // :26-147
            copy_in(sQik, Ginv, SS, lane);
            copy_in(sq, g, S, lane);
            wave_sync();
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
        copy_in(sA, Cd + cm, SS, lane);                                      // :189-196
        copy_in(sB, Cd + cm + SS, SC, lane);
        copy_in(sQim, Ginv + gm, SS, lane);
        copy_in(sRim, Ginv + gm + SS, CC, lane);
        copy_in(sQik, Ginv + gk, SS, lane);
        copy_in(sq, g + (size_t)(k - 1) * n, n, lane);                       // q_{k-1}, r_{k-1}
        copy_in(sq + n, g + (size_t)k * n, S, lane);                         // q_k
        wave_sync();
        mm<T, S, S, S, false>(sPhi, sA, sQim, lane);                         // phi = A Q_{k-1}^-1   :277-285
        mm<T, S, C, C, false>(sBR, sB, sRim, lane);                          // BR = B R_{k-1}^-1    :293-301
        mv<T, S, S>(sv, sQik, sq + n, lane);                                 // Q_k^-1 q_k           :306-310
        wave_sync();
        mv<T, S, S>(sv + S, sPhi, sq, lane);                                 // phi q_{k-1}          :316-320
        mv<T, S, C>(sv + 2 * S, sBR, sq + S, lane);                          // BR r_{k-1}           :324-328
        mm<T, S, S, S, true>(sTh, sPhi, sA, lane);                           // phi A^T              :342-351
        mm<T, S, C, S, true>(sTmp, sBR, sB, lane);                           // BR B^T               :368-377
        wave_sync();
        for (int i = lane; i < S; i += WAVE) {
            T gt = sv[i] - c[(size_t)k * S + i];                             // :311-313
            gt += sv[2 * S + i] + sv[S + i];                                 // :336-338
            gamma[(size_t)k * S + i] = -gt;                                  // :435-438
        }
        for (int i = lane; i < SS; i += WAVE) {
            T th = sTh[i] + sQik[i];                                         // :362-364
            th += sTmp[i];                                                   // :382-384
            sTh[i] = th;
            Sk[i] = -sPhi[i];                                                // S[k].left   :388-394
            Sk[SS + i] = -th;                                                // S[k].main   :398-404
            const int r = i % S, cc = i / S;
            Sk[2 * SS + i - 3 * SS] = -sPhi[cc + r * S];                     // S[k-1].right = -phi^T  :443-455
            Pk[i] = (T)0;                                                    // stair blocks come from form_ss
            Pk[2 * SS + i] = (T)0;
            if (k == K - 1) Sk[2 * SS + i] = (T)0;                           // last right: unused (:166-174)
        }
        wave_sync();
        gj_inverse<T, S>(sTh, sTmp, stmp, lane);                             // theta^-1    :407-414
        for (int i = lane; i < SS; i += WAVE) Pk[SS + i] = -sTmp[i];         // Pinv[k].main :415-422
    }
}

// ---- A3: symmetric stair (gato_schur.cuh:497-649) ------------------------------------------------
template <typename T, int S>
__global__ __launch_bounds__(WAVE) void ss_kernel(const T *__restrict__ Sbd, T *__restrict__ Pbd, int K, BatchStride bs)
{
    constexpr int SS = S * S;
    Sbd += blockIdx.y * bs.bd; Pbd += blockIdx.y * bs.bd;
    __shared__ T sPm[SS], sX[SS], sPn[SS], sT[SS], sO[SS];
    const int lane = threadIdx.x;
    for (int k = blockIdx.x; k < K; k += gridDim.x) {
        T *Pk = Pbd + (size_t)k * 3 * SS;
        wave_sync();
        copy_in(sPm, Pk + SS, SS, lane);
        if (k > 0) {                                                         // :578-611
            copy_in(sX, Sbd + (size_t)k * 3 * SS, SS, lane);                 // S[k].left
            copy_in(sPn, Pbd + (size_t)(k - 1) * 3 * SS + SS, SS, lane);     // Pinv[k-1].main
            wave_sync();
            mm<T, S, S, S, false>(sT, sPm, sX, lane);
            wave_sync();
            mm<T, S, S, S, false>(sO, sT, sPn, lane);
            wave_sync();
            for (int i = lane; i < SS; i += WAVE) Pk[i] = -sO[i];
        }
        if (k < K - 1) {                                                     // :614-648 (k < K-1 only: D1)
            wave_sync();
            const T *Sl1 = Sbd + (size_t)(k + 1) * 3 * SS;                   // S[k+1].left, read transposed
            for (int i = lane; i < SS; i += WAVE) sX[(i % S) * S + i / S] = Sl1[i];
            copy_in(sPn, Pbd + (size_t)(k + 1) * 3 * SS + SS, SS, lane);     // Pinv[k+1].main
            wave_sync();
            mm<T, S, S, S, false>(sT, sPm, sX, lane);
            wave_sync();
            mm<T, S, S, S, false>(sO, sT, sPn, lane);
            wave_sync();
            for (int i = lane; i < SS; i += WAVE) Pk[2 * SS + i] = -sO[i];
        }
    }
}

// ---- A9: dz back-substitution (gato_schur.cuh:758-867) -------------------------------------------
template <typename T, int S, int C>
__global__ __launch_bounds__(WAVE) void dz_kernel(const T *__restrict__ Ginv, const T *__restrict__ Cd,
                                                  const T *__restrict__ g, const T *__restrict__ lambda,
                                                  int K, T *__restrict__ dz, BatchStride bs)
{
    constexpr int n = S + C, SS = S * S, CC = C * C, SC = S * C;
    Ginv += blockIdx.y * bs.g; Cd += blockIdx.y * bs.c; g += blockIdx.y * bs.n; lambda += blockIdx.y * bs.sk;
    dz += blockIdx.y * bs.n;
    __shared__ T sQi[SS], sA[SS], sRi[CC > 0 ? CC : 1], sB[SC > 0 ? SC : 1], sl[2 * S], st[S + C], sg[S + C];
    const int lane = threadIdx.x;
    for (int k = blockIdx.x; k < K; k += gridDim.x) {
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
            for (int cc = 0; cc < S; ++cc) res += sQi[r + cc * S] * st[cc];
            dz[(size_t)k * n + r] = res;
        }
        if (!last) {
            for (int r = lane; r < C; r += WAVE) {                           // R_k^-1 (...)         :799-808
                T res = (T)0;
#pragma unroll
                for (int cc = 0; cc < C; ++cc) res += sRi[r + cc * C] * st[S + cc];
                dz[(size_t)k * n + S + r] = res;
            }
        }
    }
}

}  // namespace

template <typename T, int S, int C>
int launch_convert(const Dims &d, const int *G_row, const int *G_col, const T *G_val, const int *C_row,
                   const int *C_col, const T *C_val, T rho, T *Gd, T *Cd, hipStream_t st)
{
    // one memset when the two outputs sit back to back in the solver's arena (they do in gato_linsys_device)
    const char *g_end = (const char *)(Gd + d.g_dense() * d.B);
    const char *c_end = (const char *)(Cd + d.c_dense() * d.B);
    if ((const char *)Cd >= g_end && (const char *)Cd - g_end <= 4096 && d.c_dense()) {
        GATO_HIP_CHECK(hipMemsetAsync(Gd, 0, (size_t)(c_end - (const char *)Gd), st));
    } else {
        GATO_HIP_CHECK(hipMemsetAsync(Gd, 0, d.g_dense() * d.B * sizeof(T), st));
        if (d.c_dense()) GATO_HIP_CHECK(hipMemsetAsync(Cd, 0, d.c_dense() * d.B * sizeof(T), st));
    }
    const long long work = ((long long)d.N() + (long long)d.sk()) * 32;
    const int threads = 256;
    const long long want = (work + threads - 1) / threads;
    const int blocks = (int)(want < 4096 ? want : 4096);
    hipLaunchKernelGGL((convert_kernel<T, S, C>), dim3(blocks, d.B), dim3(threads), 0, st, G_row, G_col, G_val, C_row,
                       C_col, C_val, d.K, rho, Gd, Cd, batch_stride(d));
    GATO_HIP_CHECK(hipGetLastError());
    return GATO_OK;
}

static inline int knot_grid(int K) { return K < 8192 ? K : 8192; }

template <typename T, int S, int C>
int launch_form_schur(const Dims &d, const T *Gd, const T *Cd, const T *g, const T *c, T *Sbd, T *Pbd,
                      T *gamma, T *Ginv, hipStream_t st)
{
    hipLaunchKernelGGL((invert_G_kernel<T, S, C>), dim3(knot_grid(d.K), d.B), dim3(WAVE), 0, st, Gd, Ginv, d.K,
                       batch_stride(d));
    GATO_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL((schur_kernel<T, S, C>), dim3(knot_grid(d.K), d.B), dim3(WAVE), 0, st, Gd, Ginv, Cd, g, c, d.K,
                       Sbd, Pbd, gamma, batch_stride(d));
    GATO_HIP_CHECK(hipGetLastError());
    return GATO_OK;
}

template <typename T, int S, int C>
int launch_form_ss(const Dims &d, const T *Sbd, T *Pbd, hipStream_t st)
{
    hipLaunchKernelGGL((ss_kernel<T, S>), dim3(knot_grid(d.K), d.B), dim3(WAVE), 0, st, Sbd, Pbd, d.K, batch_stride(d));
    GATO_HIP_CHECK(hipGetLastError());
    return GATO_OK;
}

template <typename T, int S, int C>
int launch_compute_dz(const Dims &d, const T *Ginv, const T *Cd, const T *g, const T *lambda, T *dz,
                      hipStream_t st)
{
    hipLaunchKernelGGL((dz_kernel<T, S, C>), dim3(knot_grid(d.K), d.B), dim3(WAVE), 0, st, Ginv, Cd, g, lambda, d.K, dz,
                       batch_stride(d));
    GATO_HIP_CHECK(hipGetLastError());
    return GATO_OK;
}

#define X(S_, C_)                                                                                              \
    template int launch_convert<float, S_, C_>(const Dims &, const int *, const int *, const float *, const int *, \
                                               const int *, const float *, float, float *, float *, hipStream_t); \
    template int launch_convert<double, S_, C_>(const Dims &, const int *, const int *, const double *,           \
                                                const int *, const int *, const double *, double, double *,      \
                                                double *, hipStream_t);                                          \
    template int launch_form_schur<float, S_, C_>(const Dims &, const float *, const float *, const float *,     \
                                                  const float *, float *, float *, float *, float *, hipStream_t); \
    template int launch_form_schur<double, S_, C_>(const Dims &, const double *, const double *, const double *, \
                                                   const double *, double *, double *, double *, double *,       \
                                                   hipStream_t);                                                 \
    template int launch_form_ss<float, S_, C_>(const Dims &, const float *, float *, hipStream_t);               \
    template int launch_form_ss<double, S_, C_>(const Dims &, const double *, double *, hipStream_t);            \
    template int launch_compute_dz<float, S_, C_>(const Dims &, const float *, const float *, const float *,     \
                                                  const float *, float *, hipStream_t);                          \
    template int launch_compute_dz<double, S_, C_>(const Dims &, const double *, const double *, const double *, \
                                                   const double *, double *, hipStream_t);
GATO_SHAPES(X)
#undef X

}  // namespace gato
