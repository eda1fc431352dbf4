// Single-reduction PCG (Chronopoulos-Gear recurrences) for gfx950 - OPT-IN variant of the resident kernel
// (solver option pcg_variant = 1).  Not the reference's recurrence: mathematically the same Krylov iterates, but
// upsilon = S p is carried by the recurrence s = w + beta s instead of a product, and both dots of an iteration are
// taken at ONE point, so an iteration needs ONE inter-workgroup hand-off instead of two:
//
//     u = Pinv r ;  w = S u ;  gamma' = r.u ;  delta = w.u                 -> one all-gather {gamma', delta, w blocks}
//     exit test on |gamma'| (the reference's eta' = r.Pinv r, gato_pcg.cuh:404) ;
//     beta = gamma'/gamma ;  alpha = gamma' / (delta - beta gamma'/alpha)
//     p = u + beta p ; s = w + beta s ; lambda += alpha p ; r -= alpha s
//
// On MI355X a hand-off between workgroups costs ~1.7-2 us and is what bounds every multi-workgroup
// shape, so halving them is worth ~1.6x there.  The price is rounding that differs from the reference recurrence
// (same answer to solver tolerance, iteration count may move by one), which is why it is not the default.
//
// Communication avoiding layout: a workgroup owns knots [k0,k1) but ALSO computes u on the two neighbouring knots
// k0-1 and k1 (it keeps their Pinv rows in the registers of "ghost lanes"), so w = S u needs no exchange of u; that
// takes r two knots deep on each side, which every workgroup advances locally (ghost_s = ghost_w + beta ghost_s,
// ghost_r -= alpha ghost_s) from the neighbours' first/last TWO blocks of w - the only vector data in the hand-off.
#include "gato_pcg_device.h"

namespace gato {
namespace {

template <typename T, int S, int MAXT>
__global__ __launch_bounds__(MAXT) void pcg_cg1_kernel(PcgLaunch a)
{
    typedef Granule<T> Gr;
    constexpr int GPV = Gr::GPV;
    constexpr int VW = VecOf<T>::W;
    constexpr int SP = pad_to(S, VW);
    constexpr int MAXK = (MAXT + S - 1) / S;          // knots covered by lanes (own + 2 ghost-lane knots)
    constexpr int PM = 256 / 64;
    __shared__ __attribute__((aligned(16))) T xr[(MAXK + 4) * SP];   // r on knots k0-2 .. k1+1   (slot = k - (k0-2))
    __shared__ __attribute__((aligned(16))) T xu[(MAXK + 2) * SP];   // u on knots k0-1 .. k1     (slot = k - (k0-1))
    __shared__ __attribute__((aligned(32))) T wpart[2][2][4 * ((MAXT + 63) / 64)];
    __shared__ T gw[4][32];                                          // received w blocks: L2, L1, R1, R2
    __shared__ T bc[2][2];
    __shared__ int s_abort;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const bool batched = a.batch > 1;
    // xcd_pack: see pcg_resident_kernel (placement hint: the working groups share X XCDs)
    const int X = a.xcd_pack;
    const int xres = X > 0 ? (int)((blockIdx.x - (unsigned)a.xcd_sel) & 7) : 0;      // a.xcd_sel: which XCD(s) of the eight host the working blocks
    if (X > 0 && xres >= X) return;
    const int per_x = X > 0 ? (int)(gridDim.x >> 3) : 0;
    const int wg = batched ? 0 : (X > 0 ? xres * per_x + (int)(blockIdx.x >> 3) : (int)blockIdx.x);
    const int W = batched ? 1 : (X > 0 ? a.groups : (int)gridDim.x);
    if (X > 0 && wg >= W) return;
    const size_t sys = batched ? blockIdx.x : 0;
    const int K = a.K;
    const int k0 = wg * a.knots_per_wg;
    const int nk = min(a.knots_per_wg, K - k0);
    const int k1 = k0 + nk;
    const int jl = tid / S, r_ = tid - jl * S;        // lane knot slot (0 = knot k0-1), row
    const int k = k0 - 1 + jl;
    const bool lane_on = jl < nk + 2 && k >= 0 && k < K;     // own or ghost-lane knot
    const bool own = lane_on && k >= k0 && k < k1;

    const T *__restrict__ dS = static_cast<const T *>(a.S_bd) + sys * 3 * S * S * K;
    const T *__restrict__ dP = static_cast<const T *>(a.P_bd) + sys * 3 * S * S * K;
    const T *__restrict__ dG = static_cast<const T *>(a.gamma) + sys * S * K;
    T *__restrict__ dL = static_cast<T *>(a.lambda) + sys * S * K;

    T sm[3 * S], pm[3 * S];
    {
        const size_t base = (size_t)(lane_on ? k : 0) * 3 * S * S + r_;
#pragma unroll
        for (int c = 0; c < 3 * S; ++c) {
            const bool ok = lane_on && !(k == 0 && c < S) && !(k == K - 1 && c >= 2 * S);   // gato_utils.cuh:157-174
            sm[c] = (ok && own) ? dS[base + (size_t)c * S] : (T)0;
            pm[c] = ok ? dP[base + (size_t)c * S] : (T)0;
        }
    }
    const int slotG = pcg_slot_granules_cg1(S, (int)sizeof(T));
    gu64 *slots = (gu64 *)a.slots;
    gi32 *g_status = (gi32 *)a.status;
    if (tid == 0) s_abort = 0;
    // r = gamma on knots k0-2 .. k1+1 (zeros outside the system)
    for (int i = tid; i < (nk + 4) * S; i += blockDim.x) {
        const int kk = k0 - 2 + i / S;
        xr[(i / S) * SP + i % S] = (kk >= 0 && kk < K) ? dG[(size_t)kk * S + i % S] : (T)0;
    }
    for (int i = tid; i < (MAXK + 2) * SP; i += blockDim.x) xu[i] = (T)0;
    __syncthreads();

    // ghost threads: thread t < 4S owns element t%S of ghost block t/S (0: k0-2, 1: k0-1, 2: k1, 3: k1+1)
    const int gb = tid / S, ge = tid - gb * S;
    const bool ghost_thr = tid < 4 * S;
    const int gk = gb < 2 ? k0 - 2 + gb : k1 + (gb - 2);
    const bool ghost_on = ghost_thr && gk >= 0 && gk < K;
    const int gslot = gb < 2 ? gb : nk + gb;                       // slot in xr
    T g_r = ghost_on ? xr[gslot * SP + ge] : (T)0, g_s = (T)0;

    T lam = (T)0, r = own ? xr[(jl + 1) * SP + r_] : (T)0;
    T p = (T)0, s = (T)0, u = (T)0, w = (T)0;
    unsigned epoch = a.epoch0;
    bool aborted = false;
    const unsigned long long t_limit = a.timeout_ticks;

    // one hand-off: two dots + the neighbours' two boundary blocks of w on each side
    auto exchange = [&](T d0, T d1, T &t0, T &t1) {
        ++epoch;
        partials_store(wpart[epoch & 1][0], wave, lane, d0);
        partials_store(wpart[epoch & 1][1], wave, lane, d1);
        gu64 *mine = slots + ((size_t)(epoch & 1) * W + wg) * slotG;
        if (W > 1 && own) {
            const int j = k - k0;                                  // own knot index
            if (j < 2) Gr::store(mine + 16 + (j * S + r_) * GPV, epoch, w);                  // first two blocks
            if (j >= nk - 2) Gr::store(mine + 16 + ((2 + j - (nk - 2)) * S + r_) * GPV, epoch, w);   // last two
        }
        __syncthreads();
        if (W == 1) {
            t0 = partials_total(wpart[epoch & 1][0], nwaves, lane);
            t1 = partials_total(wpart[epoch & 1][1], nwaves, lane);
            return;
        }
        if (wave == 0) {
            T s0 = partials_total(wpart[epoch & 1][0], nwaves, lane);
            T s1 = partials_total(wpart[epoch & 1][1], nwaves, lane);
            if (lane == 0) { Gr::store(mine, epoch, s0); Gr::store(mine + GPV, epoch, s1); }
            gu64 *pbase = slots + (size_t)(epoch & 1) * W * slotG;
            // halo: lanes [0,32) fetch from the left neighbour (its last two blocks), [32,64) from the right one
            // (its first two); each lane covers elements e, e+32 of the 2S-element pair of blocks.
            const bool left = lane < 32;
            const bool have_nb = left ? (k0 > 0) : (k1 < K);
            const int nb = left ? wg - 1 : wg + 1;
            const int he = lane & 31;
            gu64 *hp[2];
            bool hw[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = he + 32 * q;
                hw[q] = have_nb && e < 2 * S;
                hp[q] = hw[q] ? pbase + (size_t)nb * slotG + 16 + ((left ? 2 * S : 0) + e) * GPV : mine;
            }
            gu64 *pptr[PM];
#pragma unroll
            for (int m = 0; m < PM; ++m) pptr[m] = pbase + (size_t)min(lane + 64 * m, W - 1) * slotG;
            const int pm_count = (W + 63) >> 6;
            unsigned long long raw[PM][2 * GPV], hraw[2][GPV];
            const unsigned long long tstart = __builtin_amdgcn_s_memrealtime();
            bool fail = false;
            // sleep before the first sweep of a cross-XCD hand-off (round 1: s_sleep 10, -6..-17 % here): with 512-thread workgroups
            // the W-dependent rule of gato_pcg_resident.hip (l_sleep; 14/7/4096 f32 2.87 -> 2.72 us); with 256-thread workgroups
            // (32/16, fp64) that rule overshoots (32/16/1024 f32 3.06 -> 3.25): they keep the old value
            if (W > 32) {
                if (blockDim.x >= 512) {
                    const int sl = 10 + W / 22;
                    for (int i = 0; i < sl; ++i) __builtin_amdgcn_s_sleep(1);
                } else __builtin_amdgcn_s_sleep(10);
            }
            for (unsigned spin = 0;; ++spin) {
#pragma unroll
                for (int m = 0; m < PM; ++m)
                    if (m < pm_count) {
#pragma unroll
                        for (int g = 0; g < 2 * GPV; ++g)
                            raw[m][g] = __hip_atomic_load(pptr[m] + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int g = 0; g < GPV; ++g)
                        hraw[q][g] = __hip_atomic_load(hp[q] + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                bool ok = true;
#pragma unroll
                for (int m = 0; m < PM; ++m)
                    if (m < pm_count) {
#pragma unroll
                        for (int g = 0; g < 2 * GPV; ++g) ok &= (unsigned)(raw[m][g] >> 32) == epoch;
                    }
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int g = 0; g < GPV; ++g) ok &= (unsigned)(hraw[q][g] >> 32) == epoch;
                if (__all(ok)) break;
                if ((spin & 255u) == 255u) {
                    const bool late = __builtin_amdgcn_s_memrealtime() - tstart > t_limit;
                    const bool other = __hip_atomic_load(g_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.launch_id;
                    if (late || other) { fail = true; break; }
                }
            }
            if (fail && lane == 0) {
                __hip_atomic_store(g_status, a.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_abort = 1;
            }
            T a0 = (T)0, a1 = (T)0;
#pragma unroll
            for (int m = 0; m < PM; ++m)
                if (m < pm_count && lane + 64 * m < W) {
                    unsigned long long x0[GPV], x1[GPV];
#pragma unroll
                    for (int g = 0; g < GPV; ++g) { x0[g] = raw[m][g]; x1[g] = raw[m][GPV + g]; }
                    a0 += Gr::decode(x0);
                    a1 += Gr::decode(x1);
                }
            s0 = wave_sum(a0);
            s1 = wave_sum(a1);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = he + 32 * q;
                if (e < 2 * S) gw[(left ? 0 : 2) + e / S][e % S] = hw[q] ? Gr::decode(hraw[q]) : (T)0;
            }
            if (lane == 0) { bc[epoch & 1][0] = s0; bc[epoch & 1][1] = s1; }
        }
        __syncthreads();
        t0 = bc[epoch & 1][0];
        t1 = bc[epoch & 1][1];
        aborted = s_abort != 0;
    };

    // u = Pinv r (own + ghost-lane knots), w = S u (own), gamma = r.u, delta = w.u
    auto products = [&](T &gam, T &del) {
        u = lane_on ? row_times_window<T, S, SP>(pm, &xr[jl * SP]) : (T)0;      // window slots jl, jl+1, jl+2 = knots k-1..k+1
        if (lane_on) xu[jl * SP + r_] = u;
        __syncthreads();
        w = own ? row_times_window<T, S, SP>(sm, &xu[(jl - 1) * SP]) : (T)0;   // xu slot = k - (k0-1) = jl
        exchange(own ? r * u : (T)0, own ? w * u : (T)0, gam, del);
    };

    T gamma_ = (T)0, delta = (T)0, alpha = (T)0, beta = (T)0, gamma_new = (T)0;
    products(gamma_, delta);
    const bool rec = a.eta_hist && wg == 0 && tid == 0 && sys == 0;
    if (rec) a.eta_hist[0] = (double)gamma_;
    int iters = a.max_iters;
    const T tol = (T)a.exit_tol;
    if (!aborted) {
        alpha = gamma_ / delta;
        for (int it = 0; it < a.max_iters; ++it) {
            // p = u + beta p ; s = w + beta s ; lambda += alpha p ; r -= alpha s   (ghosts: s, r from received w)
            p = u + beta * p;
            s = w + beta * s;
            lam += alpha * p;
            r -= alpha * s;
            if (own) xr[(jl + 1) * SP + r_] = r;
            if (W > 1 && ghost_thr) {
                g_s = gw[gb][ge] + beta * g_s;
                g_r -= alpha * g_s;
                if (ghost_on) xr[gslot * SP + ge] = g_r;
            }
            __syncthreads();
            products(gamma_new, delta);
            if (aborted) break;
            if (rec) a.eta_hist[it + 1] = (double)gamma_new;
            if (fabs(gamma_new) < tol) { iters = it; break; }               // gato_pcg.cuh:404-411
            beta = quotient(gamma_new, gamma_);
            alpha = quotient(gamma_new, delta - quotient(beta * gamma_new, alpha));
            gamma_ = gamma_new;
        }
    }
    if (own) dL[(size_t)k * S + r_] = lam;
    if (wg == 0 && tid == 0) {
        a.iters[sys] = aborted ? -1 : iters;
        if (a.final_eta && sys == 0) *a.final_eta = (double)gamma_new;
    }
}

// ---- Pipelined PCG (Ghysels-Vanroose), OPT-IN variant 2 (solver option pcg_variant = 2) ---------------------------------
// hide the hand-off instead of shortening it.  The single-reduction kernel above still WAITS for its one
// all-to-all per iteration (95 % of the wave-cycles of a 114-workgroup launch are waits).  Here the two dots of an iteration
// are published BEFORE its two block-tridiagonal products and collected after them:
//
//     r, u = Pinv r, w = S u                                      (set-up, one blocking exchange of w's boundary blocks)
//     loop:  gamma = r.u, delta = w.u          -> published (granules), nobody waits
//            m = Pinv w ;  n = S m             <- the all-to-all travels while these run
//            n's boundary blocks               -> published (the neighbours need them for their ghost blocks)
//            collect gamma, delta, the neighbours' blocks of n
//            exit test on |gamma| (= the reference's eta' = r.Pinv r, gato_pcg.cuh:404)
//            beta = gamma / gamma_old ;  alpha = gamma / (delta - beta gamma / alpha_old)
//            z = n + beta z ; q = m + beta q ; s = w + beta s ; p = u + beta p
//            lambda += alpha p ; r -= alpha s ; u -= alpha q ; w -= alpha z
//
// Same Krylov iterates as the reference recurrence in exact arithmetic; in floating point u, w (and s, q, z) are carried by
// recurrences instead of products, so rounding differs more than in the single-reduction variant and the attainable accuracy
// is lower (the usual price of pipelined CG): opt-in only, parity is claimed for the default recurrence.  Layout as above:
// a workgroup also computes m on its two neighbouring knots (ghost lanes keep those Pinv rows), which takes w two knots deep
// on each side; those four ghost blocks of w are advanced locally (ghost_z = ghost_n + beta ghost_z, ghost_w -= alpha ghost_z)
// from the neighbours' first / last two blocks of n - the only vector data in the hand-off.  One neighbour-to-neighbour
// latency stays on the critical path (the blocks of n), the W x W sweep does not.
// MEASURED (round 4): 14/7/4096 f32 2.78 us per iteration against 2.70 single-reduction and 3.48 default; 14/7/512 2.53 /
// 2.36 / 2.20; 32/16/1024 3.99 / 3.26 / 3.58; 14/7/4096 f64 4.16 / 5.19 / 4.13 - NO gain over the single-reduction variant:
// on this chip a sweep of W granule lines and a sweep of the two neighbours' lines cost the same (store -> visible 0.5 us +
// sc1 load 0.8 us across XCDs: latency, not bandwidth), and the neighbours' blocks of n cannot be published before the
// products that form them, so exactly one exchange latency stays exposed either way.  Hiding that one too needs ghost zones
// that grow by two knots per iteration (s-step methods), not a reordering.  Kept as an option: correct, deterministic,
// tested against its own restatement (oracle.pcg_pipelined).
#ifndef GATO_PIPE_SLEEP
#define GATO_PIPE_SLEEP 12
#endif
template <typename T, int S, int MAXT>
__global__ __launch_bounds__(MAXT) void pcg_pipe_kernel(PcgLaunch a)
{
    typedef Granule<T> Gr;
    constexpr int GPV = Gr::GPV;
    constexpr int VW = VecOf<T>::W;
    constexpr int SP = pad_to(S, VW);
    constexpr int MAXK = (MAXT + S - 1) / S;          // knots covered by lanes (own + 2 ghost-lane knots)
    constexpr int PM = 256 / 64;
    __shared__ __attribute__((aligned(16))) T xw[(MAXK + 4) * SP];   // w (r during set-up) on knots k0-2 .. k1+1   (slot = k - (k0-2))
    __shared__ __attribute__((aligned(16))) T xm[(MAXK + 2) * SP];   // m (u during set-up) on knots k0-1 .. k1     (slot = k - (k0-1))
    __shared__ __attribute__((aligned(32))) T wpart[2][2][4 * ((MAXT + 63) / 64)];
    __shared__ T gw[4][32];                                          // received blocks: L2, L1, R1, R2
    __shared__ T bc[2][2];
    __shared__ int s_abort;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const bool batched = a.batch > 1;
    const int X = a.xcd_pack;
    const int xres = X > 0 ? (int)((blockIdx.x - (unsigned)a.xcd_sel) & 7) : 0;
    if (X > 0 && xres >= X) return;
    const int per_x = X > 0 ? (int)(gridDim.x >> 3) : 0;
    const int wg = batched ? 0 : (X > 0 ? xres * per_x + (int)(blockIdx.x >> 3) : (int)blockIdx.x);
    const int W = batched ? 1 : (X > 0 ? a.groups : (int)gridDim.x);
    if (X > 0 && wg >= W) return;
    const size_t sys = batched ? blockIdx.x : 0;
    const int K = a.K;
    const int k0 = wg * a.knots_per_wg;
    const int nk = min(a.knots_per_wg, K - k0);
    const int k1 = k0 + nk;
    const int jl = tid / S, r_ = tid - jl * S;        // lane knot slot (0 = knot k0-1), row
    const int k = k0 - 1 + jl;
    const bool lane_on = jl < nk + 2 && k >= 0 && k < K;     // own or ghost-lane knot
    const bool own = lane_on && k >= k0 && k < k1;

    const T *__restrict__ dS = static_cast<const T *>(a.S_bd) + sys * 3 * S * S * K;
    const T *__restrict__ dP = static_cast<const T *>(a.P_bd) + sys * 3 * S * S * K;
    const T *__restrict__ dG = static_cast<const T *>(a.gamma) + sys * S * K;
    T *__restrict__ dL = static_cast<T *>(a.lambda) + sys * S * K;

    T sm[3 * S], pm[3 * S];
    {
        const size_t base = (size_t)(lane_on ? k : 0) * 3 * S * S + r_;
#pragma unroll
        for (int c = 0; c < 3 * S; ++c) {
            const bool ok = lane_on && !(k == 0 && c < S) && !(k == K - 1 && c >= 2 * S);   // gato_utils.cuh:157-174
            sm[c] = (ok && own) ? dS[base + (size_t)c * S] : (T)0;
            pm[c] = ok ? dP[base + (size_t)c * S] : (T)0;
        }
    }
    const int slotG = pcg_slot_granules_cg1(S, (int)sizeof(T));
    gu64 *slots = (gu64 *)a.slots;
    gi32 *g_status = (gi32 *)a.status;
    if (tid == 0) s_abort = 0;
    for (int i = tid; i < (nk + 4) * S; i += blockDim.x) {          // r = gamma on knots k0-2 .. k1+1 (zeros outside the system)
        const int kk = k0 - 2 + i / S;
        xw[(i / S) * SP + i % S] = (kk >= 0 && kk < K) ? dG[(size_t)kk * S + i % S] : (T)0;
    }
    for (int i = tid; i < (MAXK + 2) * SP; i += blockDim.x) xm[i] = (T)0;
    __syncthreads();

    // ghost threads: thread t < 4S owns element t%S of ghost block t/S (0: k0-2, 1: k0-1, 2: k1, 3: k1+1)
    const int gb = tid / S, ge = tid - gb * S;
    const bool ghost_thr = tid < 4 * S;
    const int gk = gb < 2 ? k0 - 2 + gb : k1 + (gb - 2);
    const bool ghost_on = ghost_thr && gk >= 0 && gk < K;
    const int gslot = gb < 2 ? gb : nk + gb;                       // slot in xw
    T g_w = (T)0, g_z = (T)0;

    unsigned epoch = a.epoch0;
    bool aborted = false;
    const unsigned long long t_limit = a.timeout_ticks;
    gu64 *mine = nullptr;

    // the workgroup's two partial dots of this epoch -> its slot (wave 0, after the barrier that follows partials_store)
    auto publish_dots = [&]() {
        if (W > 1 && wave == 0) {
            const T s0 = partials_total(wpart[epoch & 1][0], nwaves, lane);
            const T s1 = partials_total(wpart[epoch & 1][1], nwaves, lane);
            if (lane == 0) { Gr::store(mine, epoch, s0); Gr::store(mine + GPV, epoch, s1); }
        }
    };
    // the first two and the last two own blocks of a vector -> the slot (the neighbours' ghost blocks)
    auto publish_blocks = [&](T v) {
        if (W > 1 && own) {
            const int j = k - k0;
            if (j < 2) Gr::store(mine + 16 + (j * S + r_) * GPV, epoch, v);
            if (j >= nk - 2) Gr::store(mine + 16 + ((2 + j - (nk - 2)) * S + r_) * GPV, epoch, v);
        }
    };
    // wait for every workgroup's dots (if with_dots) and the neighbours' blocks of this epoch; totals -> t0, t1, blocks -> gw
    auto collect = [&](bool with_dots, T &t0, T &t1) {
        if (W == 1) {
            if (with_dots) {
                t0 = partials_total(wpart[epoch & 1][0], nwaves, lane);
                t1 = partials_total(wpart[epoch & 1][1], nwaves, lane);
            }
            return;
        }
        if (wave == 0) {
            gu64 *pbase = slots + (size_t)(epoch & 1) * W * slotG;
            const bool left = lane < 32;
            const bool have_nb = left ? (k0 > 0) : (k1 < K);
            const int nb = left ? wg - 1 : wg + 1;
            const int he = lane & 31;
            gu64 *hp[2];
            bool hw[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = he + 32 * q;
                hw[q] = have_nb && e < 2 * S;
                hp[q] = hw[q] ? pbase + (size_t)nb * slotG + 16 + ((left ? 2 * S : 0) + e) * GPV : nullptr;
            }
            gu64 *pptr[PM];
#pragma unroll
            for (int m = 0; m < PM; ++m) pptr[m] = pbase + (size_t)min(lane + 64 * m, W - 1) * slotG;
            const int pm_count = with_dots ? (W + 63) >> 6 : 0;
            unsigned long long raw[PM][2 * GPV], hraw[2][GPV];
            const unsigned long long tstart = __builtin_amdgcn_s_memrealtime();
            bool fail = false;
            // the neighbours' blocks were stored a moment ago: a first sweep that comes too early costs a whole extra round trip
            // across XCDs (the rule of the other kernels; option ablate bits 8..15 override the units for tuning)
            if (W > 32 && with_dots) {
                // measured (tools: tune_pcg.run with ablate = (units + 1) << 8): 14/7/4096 f32 0: 2.89, 8: 2.82, 12: 2.78, 16: 2.86 us per
                // iteration; 256-thread workgroups (fp64, S = 32) 0: 4.16 / 4.09, 8: 4.36 / 3.99, 12: 4.46 / 4.01
                const int sl = (a.ablate >> 8) & 255 ? ((a.ablate >> 8) & 255) - 1 : (blockDim.x >= 512 ? GATO_PIPE_SLEEP : GATO_PIPE_SLEEP / 3);
                for (int i = 0; i < sl; ++i) __builtin_amdgcn_s_sleep(1);
            }
            for (unsigned spin = 0;; ++spin) {
#pragma unroll
                for (int m = 0; m < PM; ++m)
                    if (m < pm_count) {
#pragma unroll
                        for (int g = 0; g < 2 * GPV; ++g)
                            raw[m][g] = __hip_atomic_load(pptr[m] + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int g = 0; g < GPV; ++g)
                        hraw[q][g] = hw[q] ? __hip_atomic_load(hp[q] + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ((unsigned long long)epoch << 32);
                bool ok = true;
#pragma unroll
                for (int m = 0; m < PM; ++m)
                    if (m < pm_count) {
#pragma unroll
                        for (int g = 0; g < 2 * GPV; ++g) ok &= (unsigned)(raw[m][g] >> 32) == epoch;
                    }
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int g = 0; g < GPV; ++g) ok &= (unsigned)(hraw[q][g] >> 32) == epoch;
                if (__all(ok)) break;
                if ((spin & 255u) == 255u) {
                    const bool late = __builtin_amdgcn_s_memrealtime() - tstart > t_limit;
                    const bool other = __hip_atomic_load(g_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.launch_id;
                    if (late || other) { fail = true; break; }
                }
            }
            if (fail && lane == 0) {
                __hip_atomic_store(g_status, a.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_abort = 1;
            }
            if (with_dots) {
                T a0 = (T)0, a1 = (T)0;
#pragma unroll
                for (int m = 0; m < PM; ++m)
                    if (m < pm_count && lane + 64 * m < W) {
                        unsigned long long x0[GPV], x1[GPV];
#pragma unroll
                        for (int g = 0; g < GPV; ++g) { x0[g] = raw[m][g]; x1[g] = raw[m][GPV + g]; }
                        a0 += Gr::decode(x0);
                        a1 += Gr::decode(x1);
                    }
                const T s0 = wave_sum(a0), s1 = wave_sum(a1);
                if (lane == 0) { bc[epoch & 1][0] = s0; bc[epoch & 1][1] = s1; }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = he + 32 * q;
                if (e < 2 * S) gw[(left ? 0 : 2) + e / S][e % S] = hw[q] ? Gr::decode(hraw[q]) : (T)0;
            }
        }
        __syncthreads();
        if (with_dots) { t0 = bc[epoch & 1][0]; t1 = bc[epoch & 1][1]; }
        aborted = s_abort != 0;
    };

    // set-up: u = Pinv r (own + ghost-lane knots), w = S u (own), then the neighbours' boundary blocks of w (blocking)
    T lam = (T)0, r = own ? xw[(jl + 1) * SP + r_] : (T)0;
    T u = lane_on ? row_times_window<T, S, SP>(pm, &xw[jl * SP]) : (T)0;      // window slots jl, jl+1, jl+2 = knots k-1..k+1
    if (lane_on) xm[jl * SP + r_] = u;
    __syncthreads();
    T w = own ? row_times_window<T, S, SP>(sm, &xm[(jl - 1) * SP]) : (T)0;    // xm slot = k - (k0-1) = jl
    if (!own) u = (T)0;
    ++epoch;
    mine = slots + ((size_t)(epoch & 1) * W + wg) * slotG;
    publish_blocks(w);
    __syncthreads();                                                            // every lane has read xw (r) and xm (u)
    {
        T d0, d1;
        collect(false, d0, d1);
    }
    if (ghost_thr) g_w = W > 1 ? gw[gb][ge] : (T)0;
    if (own) xw[(jl + 1) * SP + r_] = w;
    if (ghost_thr && gslot < nk + 4) xw[gslot * SP + ge] = ghost_on ? g_w : (T)0;
    partials_store(wpart[(epoch + 1) & 1][0], wave, lane, own ? r * u : (T)0);           // the first iteration's dots
    partials_store(wpart[(epoch + 1) & 1][1], wave, lane, own ? w * u : (T)0);
    __syncthreads();

    T p = (T)0, s = (T)0, q = (T)0, z = (T)0;
    T gamma_ = (T)0, gamma_old = (T)0, delta = (T)0, alpha = (T)0, beta = (T)0;
    const bool rec = a.eta_hist && wg == 0 && tid == 0 && sys == 0;
    int iters = a.max_iters;
    const T tol = (T)a.exit_tol;
    for (int it = 0; !aborted && it <= a.max_iters; ++it) {
        ++epoch;
        mine = slots + ((size_t)(epoch & 1) * W + wg) * slotG;
        publish_dots();                    // wave 0, from the partials stored before the last barrier: the dots leave FIRST, nobody waits
        const T m = lane_on ? row_times_window<T, S, SP>(pm, &xw[jl * SP]) : (T)0;       // m = Pinv w (own + ghost-lane knots)
        if (lane_on) xm[jl * SP + r_] = m;
        __syncthreads();
        const T n = own ? row_times_window<T, S, SP>(sm, &xm[(jl - 1) * SP]) : (T)0;     // n = S m
        publish_blocks(n);
        collect(true, gamma_, delta);
        if (aborted) break;
        if (rec) a.eta_hist[it] = (double)gamma_;
        if (it > 0 && fabs(gamma_) < tol) { iters = it - 1; break; }                      // gato_pcg.cuh:404-411 (eta' after update it - 1)
        if (it == a.max_iters) break;
        if (it == 0) { beta = (T)0; alpha = quotient(gamma_, delta); }
        else {
            beta = quotient(gamma_, gamma_old);
            alpha = quotient(gamma_, delta - quotient(beta * gamma_, alpha));
        }
        gamma_old = gamma_;
        z = n + beta * z;
        q = m + beta * q;
        s = w + beta * s;
        p = u + beta * p;
        lam += alpha * p;
        r -= alpha * s;
        u -= alpha * q;
        w -= alpha * z;
        if (own) xw[(jl + 1) * SP + r_] = w;
        if (W > 1 && ghost_thr) {
            g_z = gw[gb][ge] + beta * g_z;
            g_w -= alpha * g_z;
            if (ghost_on) xw[gslot * SP + ge] = g_w;
        }
        partials_store(wpart[(epoch + 1) & 1][0], wave, lane, own ? r * u : (T)0);       // the NEXT iteration's dots, in front of its barrier
        partials_store(wpart[(epoch + 1) & 1][1], wave, lane, own ? w * u : (T)0);
        __syncthreads();
    }
    if (own) dL[(size_t)k * S + r_] = lam;
    if (wg == 0 && tid == 0) {
        a.iters[sys] = aborted ? -1 : iters;
        if (a.final_eta && sys == 0) *a.final_eta = (double)gamma_;
    }
}

template <typename T, int S> struct Cg1Threads {
    static constexpr int regs = (6 * S + 3 * S) * (int)(sizeof(T) / 4) + 48;
    static constexpr int v = regs <= 128 ? 1024 : regs <= 168 ? 768 : regs <= 256 ? 512 : 256;
};
// measured: f32/14 at 768 threads spills one VGPR and is no faster than 512 (K=50 in one workgroup 1.51 us, the same as
// the reference-recurrence pair kernel; K=4096 3.21 vs 3.06 us); f64/14 at 512 and f32/32 at 512 spill 29 / 18 registers.
// All shapes therefore use the generic rule.

}  // namespace

template <typename T, int S>
int pcg_cg1_max_threads() { return Cg1Threads<T, S>::v; }

template <typename T, int S>
int launch_pcg_cg1(const PcgLaunch &a, hipStream_t st)
{
    constexpr int MAXT = Cg1Threads<T, S>::v;
    const int lanes_needed = (a.knots_per_wg + 2) * S;
    if (a.threads > MAXT || a.threads % 64 != 0 || a.threads < 4 * S || lanes_needed > a.threads || a.groups < 1 ||
        a.groups > 256 || (long long)a.groups * a.knots_per_wg < a.K || (long long)(a.groups - 1) * a.knots_per_wg >= a.K ||
        (a.groups > 1 && (a.knots_per_wg < 2 || a.K - (a.groups - 1) * a.knots_per_wg < 2))) {
        set_error("pcg_cg1: bad launch geometry (K=%d groups=%d knots/wg=%d threads=%d max=%d)", a.K, a.groups,
                  a.knots_per_wg, a.threads, MAXT);
        return GATO_EINVAL;
    }
    if (a.batch > 1 && a.groups != 1) { set_error("pcg_cg1: a batch needs one workgroup per system"); return GATO_EINVAL; }
    if (a.ev_start) GATO_HIP_CHECK(hipEventRecord(a.ev_start, st));
    const int nblocks = a.batch > 1 ? a.batch : (a.xcd_pack > 0 ? 8 * ((a.groups + a.xcd_pack - 1) / a.xcd_pack) : a.groups);
    if (a.pipelined) hipLaunchKernelGGL((pcg_pipe_kernel<T, S, MAXT>), dim3(nblocks), dim3(a.threads), 0, st, a);
    else hipLaunchKernelGGL((pcg_cg1_kernel<T, S, MAXT>), dim3(nblocks), dim3(a.threads), 0, st, a);
    GATO_HIP_CHECK(hipGetLastError());
    if (a.ev_stop) GATO_HIP_CHECK(hipEventRecord(a.ev_stop, st));
    return GATO_OK;
}

#define X(S_, C_)                                                              \
    template int pcg_cg1_max_threads<float, S_>();                             \
    template int pcg_cg1_max_threads<double, S_>();                            \
    template int launch_pcg_cg1<float, S_>(const PcgLaunch &, hipStream_t);    \
    template int launch_pcg_cg1<double, S_>(const PcgLaunch &, hipStream_t);
GATO_SHAPES(X)
#undef X

}  // namespace gato
