// Streaming PCG for gfx950: two launches per iteration, S and Pinv re-read from HBM every iteration.
//
// Replaces parallelPCG_fixed / parallelPCG_inner_fixed (src/gato_pcg.cuh:17-268), the grid-stride variant
// the reference falls back to when KNOT_POINTS exceeds the co-resident block count (gato_pcg.cuh:505-553),
// with its last-block-row defect fixed (the K4 boundary rule, gato_utils.cuh:157-174, is
// applied everywhere).  Used when K is too large for the register-resident kernel, and as the kernel the
// HBM roofline is measured on.
//
// MI355X design:
//  * a workgroup streams a tile of KT consecutive block rows [left|main|right] - one contiguous byte range
//    of the bd layout - from HBM with 16-byte-per-lane coalesced loads into LDS; lane = one row of one
//    knot then reads its 3S row entries from LDS (consecutive lanes -> consecutive addresses);
//  * the AXPY of the phase before is fused into the operand load: the window [x_{k-1};x_k;x_{k+1}] is
//    rebuilt on the fly (p = r~ + beta p_old, or r = r_old - alpha upsilon), so an iteration is exactly two
//    launches - the reference's five grid.sync() per iteration (gato_pcg.cuh:114-246) become two kernel
//    boundaries - and every vector is read/written once or twice per iteration;
//  * dots: lane product -> wave64 butterfly -> LDS -> one partial per workgroup in a slot array; the NEXT
//    launch sums the slots in fixed order (deterministic, no float atomics, no zeroing phase);
//  * the exit test runs on the device: once |eta'| < exit_tol a `done` word turns the remaining launches
//    into no-ops, so the host enqueues max_iters iterations without ever synchronising.
#include "gato_common.h"

namespace gato {
namespace {

template <typename T> struct Vec16;
template <> struct Vec16<float> { typedef float __attribute__((ext_vector_type(4))) type; static constexpr int W = 4; };
template <> struct Vec16<double> { typedef double __attribute__((ext_vector_type(2))) type; static constexpr int W = 2; };

template <typename T, int S>
struct StreamCfg {
    static constexpr int ROW = 3 * S * S;                                   // elements per block row
    static constexpr int KT_LDS = (48 * 1024) / (ROW * (int)sizeof(T));      // tile <= 48 KiB of LDS
    static constexpr int KT_THR = 256 / S;
    static constexpr int KT = KT_LDS < KT_THR ? (KT_LDS < 1 ? 1 : KT_LDS) : KT_THR;
    static constexpr int THREADS = (KT * S + 63) / 64 * 64;
    static constexpr int NWAVES = THREADS / 64;
};

template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Sum of n slot values in a fixed order, identical in every workgroup.  red: NWAVES elements of LDS.
template <typename T, int THREADS>
__device__ T slot_sum(const T *__restrict__ slots, int n, int stride, T *red)
{
    T acc = (T)0;
    for (int i = threadIdx.x; i < n; i += THREADS) acc += slots[(size_t)i * stride];
    acc = wave_sum(acc);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    T tot = (T)0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) tot += red[w];
    return tot;
}

// PHASE 0: init (r = gamma, lambda = 0, r~ = Pinv r, slot <- r.r~)                 gato_pcg.cuh:52-100
// PHASE 1: A   (p = r~ + beta p_old, upsilon = S p, slot <- p.upsilon)              gato_pcg.cuh:110-139,:209-215
// PHASE 2: B   (lambda += alpha p, r = r_old - alpha upsilon, r~ = Pinv r, slot <- r.r~)   :146-198
// The kernel works on a SHARD of block rows: K = local knot count, vectors are local; the blocks of the
// neighbouring shards (ghosts) come in through gh_* pointers when the shard is not at the global boundary
// (single-GPU: first_global = last_global = 1, no ghosts).
template <typename T, int S, int PHASE>
__global__ __launch_bounds__((StreamCfg<T, S>::THREADS)) void stream_step_kernel(StreamStep a)
{
    typedef StreamCfg<T, S> Cfg;
    typedef typename Vec16<T>::type V;
    constexpr int VW = Vec16<T>::W;
    constexpr int ROW = Cfg::ROW, KT = Cfg::KT, THREADS = Cfg::THREADS;
    __shared__ __attribute__((aligned(16))) T tile[KT * ROW];
    __shared__ T xw[(KT + 2) * S];
    __shared__ T red[Cfg::NWAVES];

    const int tid = threadIdx.x;
    const int K = a.K;
    if (PHASE != 0 && *a.done) return;

    // ---- coefficient of the fused AXPY ----------------------------------------------------------
    T coef = (T)0;
    if (PHASE == 1) {
        if (a.it > 0) {
            const T eta_new = slot_sum<T, THREADS>((const T *)a.part_num, a.num_n, a.num_stride, red);   // eta'(it-1)
            if (fabs(eta_new) < (T)a.exit_tol) {                                          // gato_pcg.cuh:207
                if (blockIdx.x == 0 && tid == 0) { *a.done = 1; *a.iters = a.it - 1; }
                return;
            }
            const T eta = slot_sum<T, THREADS>((const T *)a.part_den, a.den_n, a.den_stride, red);
            coef = eta_new / eta;                                                         // beta
        }
    } else if (PHASE == 2) {
        const T eta = slot_sum<T, THREADS>((const T *)a.part_num, a.num_n, a.num_stride, red);
        const T v = slot_sum<T, THREADS>((const T *)a.part_den, a.den_n, a.den_stride, red);
        coef = eta / v;                                                                   // alpha
    } else if (blockIdx.x == 0 && tid == 0) {
        *a.done = 0;
        *a.iters = a.max_iters;
    }

    const T *__restrict__ M = (const T *)a.M;
    const T *__restrict__ a_old = (const T *)a.a_old;
    const T *__restrict__ b = (const T *)a.b;
    T *__restrict__ a_new = (T *)a.a_new;
    T *__restrict__ y = (T *)a.y;
    const bool use_old = PHASE == 2 || (PHASE == 1 && a.it > 0);

    const int j = tid / S, r = tid - j * S;
    T part = (T)0;
    const int ntiles = (K + KT - 1) / KT;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int kt = t * KT;
        const int nk = min(KT, K - kt);
        __syncthreads();   // previous tile fully consumed
        // stage the matrix tile: contiguous range of the bd layout, 16 B per lane per load
        {
            const V *__restrict__ src = reinterpret_cast<const V *>(M + (size_t)kt * ROW);
            V *dst = reinterpret_cast<V *>(tile);
            const int nvec = nk * (ROW / VW);
            for (int i = tid; i < nvec; i += THREADS) {
                V v = src[i];
                const int e = i * VW;
                const int kk = kt + e / ROW, within = e % ROW;
                // first block row has no left block, last no right block (gato_utils.cuh:157-174)
                if ((a.first_global && kk == 0 && within < S * S) ||
                    (a.last_global && kk == K - 1 && within >= 2 * S * S)) v = (V)(T)0;
                dst[i] = v;
            }
        }
        // rebuild the operand window for knots kt-1 .. kt+nk  (fused AXPY of the previous phase)
        for (int i = tid; i < (nk + 2) * S; i += THREADS) {
            const long long gi = (long long)(kt - 1) * S + i;
            T x = (T)0;
            if (gi >= 0 && gi < (long long)K * S) {
                if (PHASE == 0) x = a_old[gi];                                   // r = gamma
                else if (PHASE == 1) x = b[gi] + coef * (use_old ? a_old[gi] : (T)0);   // p = r~ + beta p
                else x = a_old[gi] - coef * b[gi];                               // r = r - alpha upsilon
                if (i >= S && i < (nk + 1) * S) a_new[gi] = x;                   // own rows only
            } else {
                // block of a neighbouring shard: advance the ghost with the neighbour's boundary block
                const bool left = gi < 0;
                const bool have = left ? !a.first_global : !a.last_global;
                if (have) {
                    const int e = left ? (int)(gi + S) : (int)(gi - (long long)K * S);
                    const T *ga = (const T *)(left ? a.gh_a_left : a.gh_a_right);
                    const T *gb = (const T *)(left ? a.gh_b_left : a.gh_b_right);
                    if (PHASE == 0) x = ga[e];
                    else if (PHASE == 1) x = gb[e] + coef * (use_old ? ga[e] : (T)0);
                    else x = ga[e] - coef * gb[e];
                    T *gn = (T *)(left ? a.gh_new_left : a.gh_new_right);
                    gn[e] = x;                                                   // one writer: first / last tile
                }
            }
            xw[i] = x;
        }
        __syncthreads();
        if (j < nk) {
            const T *mrow = tile + j * ROW + r;
            const T *xv = xw + j * S;
            T acc = (T)0;
#pragma unroll
            for (int c = 0; c < 3 * S; ++c) acc = gato::fmaT(mrow[c * S], xv[c], acc);
            const size_t gi = (size_t)(kt + j) * S + r;
            y[gi] = acc;
            part += xv[S + r] * acc;
            if (PHASE == 2) {
                T *lam = (T *)a.lam;
                lam[gi] += coef * ((const T *)a.p_cur)[gi];                      // gato_pcg.cuh:150-153
            } else if (PHASE == 0) {
                ((T *)a.lam)[gi] = (T)0;
            }
        }
    }
    // one slot per workgroup
    part = wave_sum(part);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = part;
    __syncthreads();
    if (tid == 0) {
        T tot = (T)0;
        for (int w = 0; w < Cfg::NWAVES; ++w) tot += red[w];
        ((T *)a.part_out)[blockIdx.x] = tot;
    }
}

// After the last iteration: evaluate the exit test of iteration max_iters-1.
template <typename T, int THREADS>
__global__ __launch_bounds__(THREADS) void stream_finish_kernel(const T *part, int n, int stride, T exit_tol, int last_it,
                                                                int *done, int *iters, double *final_eta)
{
    __shared__ T red[THREADS / 64];
    if (*done) return;
    const T eta_new = slot_sum<T, THREADS>(part, n, stride, red);
    if (threadIdx.x == 0) {
        if (final_eta) *final_eta = (double)eta_new;
        if (last_it >= 0 && fabs(eta_new) < exit_tol) { *done = 1; *iters = last_it; }
    }
}

// Multi-GPU hand-off record of one shard: [sum of the workgroup slots | first S-block of y | last S-block of y]
template <typename T, int THREADS>
__global__ __launch_bounds__(THREADS) void stream_pack_kernel(const T *slots, int nslots, const T *y, int K, int S,
                                                              T *send)
{
    __shared__ T red[THREADS / 64];
    const T tot = slot_sum<T, THREADS>(slots, nslots, 1, red);
    if (threadIdx.x == 0) send[0] = tot;
    for (int i = threadIdx.x; i < S; i += THREADS) {
        send[1 + i] = y[i];
        send[1 + S + i] = y[(size_t)(K - 1) * S + i];
    }
}

}  // namespace

template <typename T, int S>
int stream_grid(int K, int max_groups)
{
    typedef StreamCfg<T, S> Cfg;
    const int ntiles = (K + Cfg::KT - 1) / Cfg::KT;
    int grid = ntiles < max_groups ? ntiles : max_groups;
    return grid > 2048 ? 2048 : grid;
}

template <typename T, int S>
int launch_stream_step(int phase, const StreamStep &a, int grid, hipStream_t st)
{
    typedef StreamCfg<T, S> Cfg;
    if (reinterpret_cast<uintptr_t>(a.M) & 15) {
        set_error("pcg_streaming: S / Pinv must be 16-byte aligned");
        return GATO_EINVAL;
    }
    if (phase == 0) hipLaunchKernelGGL((stream_step_kernel<T, S, 0>), dim3(grid), dim3(Cfg::THREADS), 0, st, a);
    else if (phase == 1) hipLaunchKernelGGL((stream_step_kernel<T, S, 1>), dim3(grid), dim3(Cfg::THREADS), 0, st, a);
    else hipLaunchKernelGGL((stream_step_kernel<T, S, 2>), dim3(grid), dim3(Cfg::THREADS), 0, st, a);
    GATO_HIP_CHECK(hipGetLastError());
    return GATO_OK;
}

template <typename T, int S>
int launch_stream_pack(const void *slots, int nslots, const void *y, int K, void *send, hipStream_t st)
{
    hipLaunchKernelGGL((stream_pack_kernel<T, 256>), dim3(1), dim3(256), 0, st, (const T *)slots, nslots, (const T *)y, K,
                       S, (T *)send);
    GATO_HIP_CHECK(hipGetLastError());
    return GATO_OK;
}

template <typename T, int S>
int launch_stream_finish(const void *part, int n, int stride, double exit_tol, int last_it, int *done, int *iters,
                         double *final_eta, hipStream_t st)
{
    hipLaunchKernelGGL((stream_finish_kernel<T, 256>), dim3(1), dim3(256), 0, st, (const T *)part, n, stride, (T)exit_tol,
                       last_it, done, iters, final_eta);
    GATO_HIP_CHECK(hipGetLastError());
    return GATO_OK;
}

template <typename T, int S>
int launch_pcg_streaming(const Dims &d, const T *Sbd, const T *Pbd, const T *gamma, T *lambda, T exit_tol,
                         int max_iters, int *iters, const PcgStreamWork &w, hipStream_t st)
{
    const int K = d.K;
    const int grid = stream_grid<T, S>(K, w.max_groups);
    // six S*K vectors, consecutive in the workspace: r and p ping-pong pairs (the fused AXPY reads the old
    // vector of neighbouring knots while the new one is written), upsilon, r~.
    T *vecs = (T *)w.vecs;
    const size_t sk = d.sk();
    T *r[2] = {vecs, vecs + sk};
    T *p[2] = {vecs + 2 * sk, vecs + 3 * sk};
    T *ups = vecs + 4 * sk, *rt = vecs + 5 * sk;
    T *PB = (T *)w.partials;                   // [3][max_groups]
    T *PA = PB + 3 * (size_t)w.max_groups;     // [max_groups]
    auto pb = [&](int bi) { return PB + (size_t)(bi % 3) * w.max_groups; };

    StreamStep a;
    memset(&a, 0, sizeof(a));
    a.K = K; a.max_iters = max_iters; a.exit_tol = (double)exit_tol; a.done = w.done; a.iters = iters;
    a.first_global = a.last_global = 1;
    a.num_n = a.den_n = grid; a.num_stride = a.den_stride = 1;
    int rc;
    // init: r0 = gamma, r~ = Pinv r0, eta slots -> pb(0)
    a.M = Pbd; a.a_old = gamma; a.b = nullptr; a.a_new = r[0]; a.y = rt; a.lam = lambda; a.part_out = pb(0); a.it = 0;
    if ((rc = launch_stream_step<T, S>(0, a, grid, st))) return rc;
    for (int it = 0; it < max_iters; ++it) {
        const int ri = it & 1, pi = it & 1;
        // A: p_new = r~ + beta p_old ; upsilon = S p_new ; v slots -> PA
        a.M = Sbd; a.a_old = p[pi ^ 1]; a.b = rt; a.a_new = p[pi]; a.y = ups; a.lam = nullptr; a.p_cur = nullptr;
        a.part_num = pb(it); a.part_den = pb(it + 2); a.part_out = PA; a.it = it;   // eta'(it-1)=pb(it), eta(it-1)=pb(it-1)
        if ((rc = launch_stream_step<T, S>(1, a, grid, st))) return rc;
        // B: r_new = r_old - alpha upsilon ; lambda += alpha p ; r~ = Pinv r_new ; eta' slots -> pb(it+1)
        a.M = Pbd; a.a_old = r[ri]; a.b = ups; a.a_new = r[ri ^ 1]; a.y = rt; a.lam = lambda; a.p_cur = p[pi];
        a.part_num = pb(it); a.part_den = PA; a.part_out = pb(it + 1);
        if ((rc = launch_stream_step<T, S>(2, a, grid, st))) return rc;
    }
    return launch_stream_finish<T, S>(pb(max_iters), grid, 1, (double)exit_tol, max_iters - 1, w.done, iters,
                                      (double *)w.scalars, st);
}

#define X(S_, C_)                                                                                                  \
    template int launch_pcg_streaming<float, S_>(const Dims &, const float *, const float *, const float *, float *, \
                                                 float, int, int *, const PcgStreamWork &, hipStream_t);            \
    template int launch_pcg_streaming<double, S_>(const Dims &, const double *, const double *, const double *,     \
                                                  double *, double, int, int *, const PcgStreamWork &, hipStream_t); \
    template int stream_grid<float, S_>(int, int);                                                                  \
    template int stream_grid<double, S_>(int, int);                                                                 \
    template int launch_stream_step<float, S_>(int, const StreamStep &, int, hipStream_t);                          \
    template int launch_stream_step<double, S_>(int, const StreamStep &, int, hipStream_t);                         \
    template int launch_stream_pack<float, S_>(const void *, int, const void *, int, void *, hipStream_t);          \
    template int launch_stream_pack<double, S_>(const void *, int, const void *, int, void *, hipStream_t);         \
    template int launch_stream_finish<float, S_>(const void *, int, int, double, int, int *, int *, double *, hipStream_t); \
    template int launch_stream_finish<double, S_>(const void *, int, int, double, int, int *, int *, double *, hipStream_t);
GATO_SHAPES(X)
#undef X

}  // namespace gato
