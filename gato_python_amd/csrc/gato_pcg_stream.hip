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

#ifndef GATO_STREAM_TILE_BYTES
#define GATO_STREAM_TILE_BYTES (38 * 1024)   /* measured best on MI355X (tools/stream_bench.py): 2 workgroups per CU */
#endif
#ifndef GATO_STREAM_AUX
#define GATO_STREAM_AUX 2   /* cache policy of the LDS-DMA tile loads: 0 default, 2 = nt (each block row is read once per launch: +7 %) */
#endif
#ifndef GATO_STREAM_MAX_ROWS
#define GATO_STREAM_MAX_ROWS 256
#endif
constexpr int cmin(int a, int b) { return a < b ? a : b; }
constexpr int cmax(int a, int b) { return a > b ? a : b; }

// Tile = KT consecutive block rows (one contiguous byte range of the bd layout), about 38 KB, double buffered in
// LDS: 2 workgroups per CU keep ~76 KB of HBM loads in flight per CU.
template <typename T, int S>
struct StreamCfg {
    static constexpr int VW = Vec16<T>::W;
    static constexpr int ROW = 3 * S * S;                                   // elements per block row
    static constexpr int KT = cmax(1, cmin(GATO_STREAM_TILE_BYTES / (ROW * (int)sizeof(T)), GATO_STREAM_MAX_ROWS / S));
    static constexpr int THREADS = (KT * S + 63) / 64 * 64;
    static constexpr int NWAVES = THREADS / 64;
    static constexpr int SPX = (S + VW - 1) / VW * VW;                      // padded knot stride of the operand window
    static constexpr int XW = (KT + 2) * SPX;
    static constexpr int NVP = ((KT + 2) * S + THREADS - 1) / THREADS;      // window elements prefetched per thread
    static constexpr int TILE_BYTES = KT * ROW * (int)sizeof(T);
    static constexpr int LDS_BYTES = 2 * TILE_BYTES + 2 * XW * (int)sizeof(T) + 64;
};

template <typename T>
__device__ __forceinline__ T wave_sum(T v) { return wave_sum_dpp(v); }

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
    constexpr int VW = Cfg::VW, ROW = Cfg::ROW, KT = Cfg::KT, THREADS = Cfg::THREADS, SPX = Cfg::SPX, NVP = Cfg::NVP;
    __shared__ __attribute__((aligned(16))) T tile[2][KT * ROW];   // LDS-DMA destination, double buffered
    __shared__ __attribute__((aligned(16))) T xw[2][Cfg::XW];
    __shared__ T red[Cfg::NWAVES];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
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
        if (a.eta_hist && blockIdx.x == 0 && tid == 0) a.eta_hist[a.it] = (double)eta;    // eta after iteration it-1 (init: 0)
    } else if (blockIdx.x == 0 && tid == 0) {
        *a.done = 0;
        *a.iters = a.max_iters;
    }

    const char *__restrict__ Mb = (const char *)a.M;
    const T *__restrict__ a_old = (const T *)a.a_old;
    const T *__restrict__ b = (const T *)a.b;
    T *__restrict__ a_new = (T *)a.a_new;
    T *__restrict__ y = (T *)a.y;
    T *__restrict__ lam = (T *)a.lam;
    const T *__restrict__ p_cur = (const T *)a.p_cur;
    const bool use_old = PHASE == 2 || (PHASE == 1 && a.it > 0);
    const long long nrows = (long long)K * S;

    const int j = tid / S, r = tid - j * S;
    const int ntiles = (K + KT - 1) / KT;

    // asynchronous tile load: 1 KiB per wave-instruction straight into LDS (global_load_lds_dwordx4)
    auto issue_tile = [&](int t, int buf) {
        const int kt = t * KT;
        const int nk = min(KT, K - kt);
        const int bytes = nk * ROW * (int)sizeof(T);
        const char *src = Mb + (size_t)kt * ROW * sizeof(T);
        char *dst = (char *)&tile[buf][0];
        for (int q = wave; q * 1024 < bytes; q += Cfg::NWAVES) {
            const int off = q * 1024 + lane * 16;
            if (off < bytes)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + off),
                                                 (__attribute__((address_space(3))) void *)(dst + q * 1024), 16, 0, GATO_STREAM_AUX);
        }
    };
    // operand-window elements of tile t this thread is responsible for: values of a_old / b (own rows or ghosts)
    T pa[NVP], pb[NVP], pp = (T)0, pl = (T)0;
    auto prefetch_vec = [&](int t) {
        const int kt = t * KT;
        const int nk = min(KT, K - kt);
#pragma unroll
        for (int m = 0; m < NVP; ++m) {
            const int i = tid + m * THREADS;
            pa[m] = (T)0; pb[m] = (T)0;
            if (i < (nk + 2) * S) {
                const long long gi = (long long)(kt - 1) * S + i;
                if (gi >= 0 && gi < nrows) {
                    if (PHASE == 0 || use_old) pa[m] = a_old[gi];
                    if (PHASE != 0 || b) pb[m] = b[gi];
                } else {
                    const bool left = gi < 0;
                    if (left ? !a.first_global : !a.last_global) {
                        const int e = left ? (int)(gi + S) : (int)(gi - nrows);
                        const T *ga = (const T *)(left ? a.gh_a_left : a.gh_a_right);
                        const T *gb = (const T *)(left ? a.gh_b_left : a.gh_b_right);
                        if (PHASE == 0 || use_old) pa[m] = ga[e];
                        if (PHASE != 0 || gb) pb[m] = gb[e];
                    }
                }
            }
        }
        if (PHASE == 2 && j < nk) {
            const size_t gi = (size_t)(kt + j) * S + r;
            pp = p_cur[gi];
            pl = lam[gi];
        }

    };

    T part = (T)0;
    int t = blockIdx.x;
    if (t < ntiles) { issue_tile(t, 0); prefetch_vec(t); }
    for (int n = 0; t < ntiles; t += gridDim.x, ++n) {
        const int buf = n & 1;
        const int kt = t * KT;
        const int nk = min(KT, K - kt);
        // the tile and the vector values were requested one iteration ago
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // rebuild the operand window for knots kt-1 .. kt+nk  (fused AXPY of the previous phase)
        T *xb = xw[buf];
#pragma unroll
        for (int m = 0; m < NVP; ++m) {
            const int i = tid + m * THREADS;
            if (i < (nk + 2) * S) {
                T x;
                if (PHASE == 0) x = pa[m] - pb[m];                               // r = gamma (- S lambda0 when warm)
                else if (PHASE == 1) x = pb[m] + coef * pa[m];                   // p = r~ + beta p   (pa = 0 at it 0)
                else x = pa[m] - coef * pb[m];                                   // r = r - alpha upsilon
                const long long gi = (long long)(kt - 1) * S + i;
                if (gi >= 0 && gi < nrows) {
                    if (i >= S && i < (nk + 1) * S) a_new[gi] = x;               // own rows only
                } else {
                    const bool left = gi < 0;
                    if (left ? !a.first_global : !a.last_global) {               // ghost of a neighbouring shard
                        T *gn = (T *)(left ? a.gh_new_left : a.gh_new_right);
                        gn[left ? (int)(gi + S) : (int)(gi - nrows)] = x;        // one writer: first / last tile
                    } else x = (T)0;
                }
                xb[(i / S) * SPX + (i % S)] = x;
            }
        }
        const T my_p = pp, my_l = pl;
        __syncthreads();
        // first block row has no left block, last no right block (gato_utils.cuh:157-174): those blocks may hold
        // anything in the caller's buffer, so they are cleared in LDS after the DMA landed
        const bool clr_first = a.first_global && kt == 0, clr_last = a.last_global && kt + nk == K;
        if (clr_first || clr_last) {
            if (clr_first) for (int i = tid; i < S * S; i += THREADS) tile[buf][i] = (T)0;
            if (clr_last) for (int i = tid; i < S * S; i += THREADS) tile[buf][(nk - 1) * ROW + 2 * S * S + i] = (T)0;
            __syncthreads();
        }
        // request the next tile while this one is consumed
        const int tn = t + gridDim.x;
        if (tn < ntiles) { issue_tile(tn, buf ^ 1); prefetch_vec(tn); }
        if (j < nk) {
            const T *mrow = &tile[buf][j * ROW + r];
            const T *xv = xb + j * SPX;
            T acc = (T)0;
#pragma unroll
            for (int bk = 0; bk < 3; ++bk) {
#pragma unroll
                for (int i = 0; i < SPX / VW; ++i) {
                    const V v = *reinterpret_cast<const V *>(xv + bk * SPX + i * VW);
#pragma unroll
                    for (int e = 0; e < VW; ++e)
                        if (i * VW + e < S) acc = gato::fmaT(mrow[(bk * S + i * VW + e) * S], v[e], acc);
                }
            }
            const size_t gi = (size_t)(kt + j) * S + r;
            y[gi] = acc;
            part += xv[SPX + r] * acc;
            if (PHASE == 2) lam[gi] = my_l + coef * my_p;                        // gato_pcg.cuh:150-153
            else if (PHASE == 0) lam[gi] = p_cur ? p_cur[gi] : (T)0;               // lambda = lambda0 or 0
        }
    }
    // one slot per workgroup
    part = wave_sum(part);
    __syncthreads();
    if (lane == 0) red[wave] = part;
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
                                                                int *done, int *iters, double *final_eta, double *eta_hist)
{
    __shared__ T red[THREADS / 64];
    if (*done) return;
    const T eta_new = slot_sum<T, THREADS>(part, n, stride, red);
    if (threadIdx.x == 0) {
        if (final_eta) *final_eta = (double)eta_new;
        if (eta_hist && last_it >= 0) eta_hist[last_it + 1] = (double)eta_new;
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
    // persistent grid: as many workgroups as fit the chip at once (LDS-limited), each walks tiles grid-stride
    int per_cu = (160 * 1024) / Cfg::LDS_BYTES;
    per_cu = per_cu > 8 ? 8 : (per_cu < 1 ? 1 : per_cu);
    if (per_cu * Cfg::THREADS > 2048) per_cu = 2048 / Cfg::THREADS;
    int grid = 256 * per_cu;
    if (grid > ntiles) grid = ntiles;
    if (grid > max_groups) grid = max_groups;
    return grid;
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
                         double *final_eta, double *eta_hist, hipStream_t st)
{
    hipLaunchKernelGGL((stream_finish_kernel<T, 256>), dim3(1), dim3(256), 0, st, (const T *)part, n, stride, (T)exit_tol,
                       last_it, done, iters, final_eta, eta_hist);
    GATO_HIP_CHECK(hipGetLastError());
    return GATO_OK;
}

template <typename T, int S>
int launch_pcg_streaming(const Dims &d, const T *Sbd, const T *Pbd, const T *gamma, T *lambda, T exit_tol,
                         int max_iters, int *iters, const PcgStreamWork &w, hipStream_t st)
{
    const int K = d.K;
    const bool warm = w.warm_start != 0;
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
    a.eta_hist = w.eta_hist;
    a.num_n = a.den_n = grid; a.num_stride = a.den_stride = 1;
    int rc;
    if (warm) {
        // true warm start: upsilon = S lambda0 with the init kernel used as a plain block-tridiagonal product
        // (its lambda/a_new outputs go to scratch), then r0 = gamma - upsilon, lambda = lambda0
        a.M = Sbd; a.a_old = lambda; a.b = nullptr; a.a_new = r[1]; a.y = ups; a.lam = p[1]; a.p_cur = nullptr;
        a.part_out = PA; a.it = 0;
        if ((rc = launch_stream_step<T, S>(0, a, grid, st))) return rc;
        GATO_HIP_CHECK(hipMemcpyAsync(p[0], lambda, sk * sizeof(T), hipMemcpyDeviceToDevice, st));   // lambda0 snapshot
    }
    // init: r0 = gamma (- S lambda0), r~ = Pinv r0, eta slots -> pb(0)
    a.M = Pbd; a.a_old = gamma; a.b = warm ? ups : nullptr; a.a_new = r[0]; a.y = rt; a.lam = lambda;
    a.p_cur = warm ? p[0] : nullptr; a.part_out = pb(0); a.it = 0;
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
                                      (double *)w.scalars, w.eta_hist, st);
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
    template int launch_stream_finish<float, S_>(const void *, int, int, double, int, int *, int *, double *, double *, hipStream_t); \
    template int launch_stream_finish<double, S_>(const void *, int, int, double, int, int *, int *, double *, double *, hipStream_t);
GATO_SHAPES(X)
#undef X

}  // namespace gato
