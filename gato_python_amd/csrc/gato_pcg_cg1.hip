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

// MR: one rank of a cluster launch (gato_cluster_pcg with pcg_variant = 1): the knots are sharded over the GPUs of
// a node and the ONE exchange of an iteration crosses them - where an exchange costs most.  Both forms of pcg_resident_kernel<...,
// MR>: FLAT (a.flat: every workgroup of every rank has a slot in every mirror; its two partial dots go into all mirrors, its
// first / last two blocks of w into its own GPU's mirror and, at the rank's edges, the neighbouring GPU's; polls on the own mirror)
// and TWO LEVELS (level 1 = the hand-off below inside the GPU; then workgroup 0 stores the rank's two totals into every mirror, the
// rank's first / last workgroup their two edge blocks into the neighbour's mirror).  Sums in slot / rank order: identical on
// every workgroup of every rank, so every rank takes the same exit decision.  The ghost-lane knots k0 - 1 and k1 of a rank's
// edge workgroups belong to the neighbouring rank: their Pinv rows (and gamma two knots deep) are read from the full-system
// arrays, so a sharded assembly must cover one knot more on either side than for the default recurrence (gato_cluster_linsys).
template <typename T, int S, int MAXT, bool MR = false>
__global__ __launch_bounds__(MAXT) void pcg_cg1_kernel(PcgLaunch a)
{
    typedef Granule<T> Gr;
    typedef GranuleSys<T> XGr;
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
    const int k_begin = MR ? a.k_begin : 0, k_end = MR ? a.k_end : K;       // this launch's knot range (a rank's shard)
    const int R = MR ? a.nranks : 1;
    const int ex = a.split_extra;                                             // balanced split (see launch_pcg_cg1): sizes differ by one
    const int k0 = k_begin + wg * a.knots_per_wg - (ex > 0 && wg > ex ? wg - ex : 0);
    const int nk = ex > 0 ? (wg < ex ? a.knots_per_wg : a.knots_per_wg - 1) : min(a.knots_per_wg, k_end - k0);
    const int k1 = k0 + nk;
    const int jl = tid / S, r_ = tid - jl * S;        // lane knot slot (0 = knot k0-1), row
    const int k = k0 - 1 + jl;
    const bool lane_on = jl < nk + 2 && k >= 0 && k < K;     // own or ghost-lane knot
    const bool own = lane_on && k >= k0 && k < k1;
    // a neighbouring block row exists in the SYSTEM (k0 > 0, k1 < K) and belongs to a workgroup of this launch or (MR) to the
    // neighbouring rank
    const bool loc_left = MR ? wg > 0 : k0 > 0, loc_right = MR ? wg < W - 1 : k1 < K;
    const bool x_left = MR && wg == 0 && k0 > 0, x_right = MR && wg == W - 1 && k1 < K;

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
    // cross-GPU mirrors (MR): layout as in pcg_resident_kernel, two ghost blocks per side
    const int xslotG = pcg_xslot_granules(S, (int)sizeof(T));
    const int xghL = 16 * GATO_MAX_RANKS, xghR = xghL + pcg_xghost_granules(S, (int)sizeof(T));
    unsigned xepoch = MR ? a.xepoch0 : 0u;
    gu64 *xp_prev = nullptr, *xp_next = nullptr;
    __shared__ unsigned long long s_xpeer[MR ? GATO_MAX_RANKS : 1];
    if constexpr (MR) {
        if (a.rank > 0) xp_prev = (gu64 *)a.xpeer[a.rank - 1];
        if (a.rank < R - 1) xp_next = (gu64 *)a.xpeer[a.rank + 1];
        if (wave == 0 && lane < R) s_xpeer[lane] = (unsigned long long)a.xpeer[lane];
    }
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
    // ---- MR, flat form: one level across the node (see the header of this kernel and pcg_resident_kernel's allreduce_flat)
    auto exchange_flat = [&](T d0, T d1, T &t0, T &t1) {
        if constexpr (MR) {
            ++epoch; ++xepoch;
            partials_store(wpart[xepoch & 1][0], wave, lane, d0);
            partials_store(wpart[xepoch & 1][1], wave, lane, d1);
            const int WT = a.flat_groups, gwi = a.flat_base + wg;
            const size_t so = a.flat_off + ((size_t)(xepoch & 1) * WT + gwi) * slotG;      // this workgroup's slot in a mirror
            gu64 *fl = (gu64 *)a.xslots;
            if (own) {
                const int j = k - k0;
                if (j < 2) {
                    XGr::store(fl + so + 16 + (j * S + r_) * GPV, xepoch, w);
                    if (x_left) XGr::store(xp_prev + so + 16 + (j * S + r_) * GPV, xepoch, w);
                }
                if (j >= nk - 2) {
                    XGr::store(fl + so + 16 + ((2 + j - (nk - 2)) * S + r_) * GPV, xepoch, w);
                    if (x_right) XGr::store(xp_next + so + 16 + ((2 + j - (nk - 2)) * S + r_) * GPV, xepoch, w);
                }
            }
            __syncthreads();
            if (wave == 0) {
                T s0 = partials_total(wpart[xepoch & 1][0], nwaves, lane);
                T s1 = partials_total(wpart[xepoch & 1][1], nwaves, lane);
                if (lane < R) {                                                    // the two partial dots go into EVERY mirror
                    XGr::store((gu64 *)s_xpeer[lane] + so, xepoch, s0);
                    XGr::store((gu64 *)s_xpeer[lane] + so + GPV, xepoch, s1);
                }
                gu64 *pbase = fl + a.flat_off + (size_t)(xepoch & 1) * WT * slotG;
                const bool left = lane < 32;
                const bool have_nb = left ? (k0 > 0) : (k1 < K);
                const int nb = left ? gwi - 1 : gwi + 1;
                const int he = lane & 31;
                gu64 *hp[2];
                bool hw[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int e = he + 32 * q;
                    hw[q] = have_nb && e < 2 * S;
                    hp[q] = hw[q] ? pbase + (size_t)nb * slotG + 16 + ((left ? 2 * S : 0) + e) * GPV : pbase + (size_t)gwi * slotG;
                }
                gu64 *pptr[PM];
#pragma unroll
                for (int m = 0; m < PM; ++m) pptr[m] = pbase + (size_t)min(lane + 64 * m, WT - 1) * slotG;
                const int pm_count = (WT + 63) >> 6;
                unsigned long long raw[PM][2 * GPV], hraw[2][GPV];
                if (WT > 32) {
                    const int sl = 10 + WT / 22;
                    for (int i = 0; i < sl; ++i) __builtin_amdgcn_s_sleep(1);
                }
                const unsigned long long tstart = __builtin_amdgcn_s_memrealtime();
                bool fail = false;
                for (unsigned spin = 0;; ++spin) {
#pragma unroll
                    for (int m = 0; m < PM; ++m)
                        if (m < pm_count) {
#pragma unroll
                            for (int g = 0; g < 2 * GPV; ++g)
                                raw[m][g] = __hip_atomic_load(pptr[m] + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int g = 0; g < GPV; ++g)
                            hraw[q][g] = __hip_atomic_load(hp[q] + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    bool ok = true;
#pragma unroll
                    for (int m = 0; m < PM; ++m)
                        if (m < pm_count) {
#pragma unroll
                            for (int g = 0; g < 2 * GPV; ++g) ok &= (unsigned)(raw[m][g] >> 32) == xepoch;
                        }
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int g = 0; g < GPV; ++g) ok &= (unsigned)(hraw[q][g] >> 32) == xepoch;     // own partial line: current
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
                    if (m < pm_count && lane + 64 * m < WT) {
                        unsigned long long x0[GPV], x1[GPV];
#pragma unroll
                        for (int g = 0; g < GPV; ++g) { x0[g] = raw[m][g]; x1[g] = raw[m][GPV + g]; }
                        a0 += XGr::decode(x0);
                        a1 += XGr::decode(x1);
                    }
                s0 = wave_sum(a0);
                s1 = wave_sum(a1);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int e = he + 32 * q;
                    if (e < 2 * S) gw[(left ? 0 : 2) + e / S][e % S] = hw[q] ? XGr::decode(hraw[q]) : (T)0;
                }
                if (lane == 0) { bc[xepoch & 1][0] = s0; bc[xepoch & 1][1] = s1; }
            }
            __syncthreads();
            t0 = bc[xepoch & 1][0];
            t1 = bc[xepoch & 1][1];
            aborted = s_abort != 0;
        }
    };
    const bool flat = MR && a.flat != 0;
    auto exchange = [&](T d0, T d1, T &t0, T &t1) {
        if constexpr (MR) {
            if (flat) { exchange_flat(d0, d1, t0, t1); return; }
            ++xepoch;
        }
        ++epoch;
        partials_store(wpart[epoch & 1][0], wave, lane, d0);
        partials_store(wpart[epoch & 1][1], wave, lane, d1);
        gu64 *mine = slots + ((size_t)(epoch & 1) * W + wg) * slotG;
        if (W > 1 && own) {
            const int j = k - k0;                                  // own knot index
            if (j < 2) Gr::store(mine + 16 + (j * S + r_) * GPV, epoch, w);                  // first two blocks
            if (j >= nk - 2) Gr::store(mine + 16 + ((2 + j - (nk - 2)) * S + r_) * GPV, epoch, w);   // last two
        }
        if constexpr (MR) {                 // the rank's two edge blocks go straight into the neighbouring GPU's mirror
            if (own) {
                const int j = k - k0;
                const size_t xo = (size_t)(xepoch & 1) * xslotG;
                if (x_left && j < 2) XGr::store(xp_prev + xo + xghR + (j * S + r_) * GPV, xepoch, w);
                if (x_right && j >= nk - 2) XGr::store(xp_next + xo + xghL + ((j - (nk - 2)) * S + r_) * GPV, xepoch, w);
            }
        }
        __syncthreads();
        if (W == 1 && !(MR && R > 1)) {
            t0 = partials_total(wpart[epoch & 1][0], nwaves, lane);
            t1 = partials_total(wpart[epoch & 1][1], nwaves, lane);
            return;
        }
        if (wave == 0) {
            T s0 = partials_total(wpart[epoch & 1][0], nwaves, lane);
            T s1 = partials_total(wpart[epoch & 1][1], nwaves, lane);
            bool fail = false;
            const bool left = lane < 32;
            const int he = lane & 31;
            if (W == 1) {                  // (cluster launch with one workgroup on this GPU: the ghosts come from level 2 only)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int e = he + 32 * q;
                    if (e < 2 * S) gw[(left ? 0 : 2) + e / S][e % S] = (T)0;
                }
            } else {
            if (lane == 0) { Gr::store(mine, epoch, s0); Gr::store(mine + GPV, epoch, s1); }
            gu64 *pbase = slots + (size_t)(epoch & 1) * W * slotG;
            // halo: lanes [0,32) fetch from the left neighbour (its last two blocks), [32,64) from the right one
            // (its first two); each lane covers elements e, e+32 of the 2S-element pair of blocks.
            const bool have_nb = left ? loc_left : loc_right;
            const int nb = left ? wg - 1 : wg + 1;
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
            }
            if constexpr (MR) {
                if (R > 1 && !fail) {
                    // ---- level 2, across the GPUs: s0, s1 = this rank's totals (identical in all its workgroups)
                    const size_t xo = (size_t)(xepoch & 1) * xslotG;
                    if (wg == 0 && lane < R) {
                        XGr::store((gu64 *)s_xpeer[lane] + xo + a.rank * 16, xepoch, s0);
                        XGr::store((gu64 *)s_xpeer[lane] + xo + a.rank * 16 + GPV, xepoch, s1);
                    }
                    gu64 *xl = (gu64 *)a.xslots + xo;                         // polls stay on THIS GPU's memory
                    gu64 *tptr = xl + (size_t)min(lane, R - 1) * 16;
                    const bool xside = left ? x_left : x_right;
                    gu64 *xh[2];
                    bool xw_[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int e = he + 32 * q;
                        xw_[q] = xside && e < 2 * S;
                        xh[q] = xw_[q] ? xl + (left ? xghL : xghR) + e * GPV : tptr;
                    }
                    unsigned long long traw[2 * GPV], xraw[2][GPV];
                    const unsigned long long t0x = __builtin_amdgcn_s_memrealtime();
                    for (unsigned spin = 0;; ++spin) {
#pragma unroll
                        for (int g = 0; g < 2 * GPV; ++g) traw[g] = __hip_atomic_load(tptr + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int g = 0; g < GPV; ++g) xraw[q][g] = __hip_atomic_load(xh[q] + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        bool ok = true;
#pragma unroll
                        for (int g = 0; g < 2 * GPV; ++g) ok &= (unsigned)(traw[g] >> 32) == xepoch;
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int g = 0; g < GPV; ++g) ok &= (unsigned)(xraw[q][g] >> 32) == xepoch;
                        if (__all(ok)) break;
                        if ((spin & 255u) == 255u) {
                            const bool late = __builtin_amdgcn_s_memrealtime() - t0x > t_limit;
                            const bool other = __hip_atomic_load(g_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.launch_id;
                            if (late || other) { fail = true; break; }
                        }
                    }
                    unsigned long long y0[GPV], y1[GPV];
#pragma unroll
                    for (int g = 0; g < GPV; ++g) { y0[g] = traw[g]; y1[g] = traw[GPV + g]; }
                    s0 = partials_sum(lane < R ? XGr::decode(y0) : (T)0);        // rank order, the same tree on every GPU
                    s1 = partials_sum(lane < R ? XGr::decode(y1) : (T)0);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int e = he + 32 * q;
                        if (xw_[q]) gw[(left ? 0 : 2) + e / S][e % S] = XGr::decode(xraw[q]);
                    }
                }
            }
            if (fail && lane == 0) {
                __hip_atomic_store(g_status, a.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_abort = 1;
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
            if ((W > 1 || (MR && R > 1)) && ghost_thr) {
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
    if constexpr (MR) (void)cluster_lambda_ghost<T, S>(a, wg, W, dL, aborted);      // lambda_{k_end} for this rank's dz launch
    if (wg == 0 && tid == 0) {
        a.iters[sys] = aborted ? -1 : iters;
        if (a.final_eta && sys == 0) *a.final_eta = (double)gamma_new;
    }
}

// (Round 4 also had a PIPELINED recurrence here (Ghysels-Vanroose, pcg_variant = 2: the all-to-all of the dots travelling behind the
// two products).  Measured no gain over this kernel on any shape - 14/7/4096 f32 2.78 us per iteration against 2.70, 32/16/1024 3.99 /
// 3.26, 14/7/4096 f64 4.16 against the default's 4.13 - because the neighbours' boundary blocks can only be published after the
// products that form them, so one exchange latency stays exposed either way (DESIGN_LOG.md R4.5).  Removed in round 5.)
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
int launch_pcg_cg1(const PcgLaunch &a0, hipStream_t st)
{
    constexpr int MAXT = Cg1Threads<T, S>::v;
    const bool mr = a0.xslots != nullptr;                   // one rank of a cluster launch
    PcgLaunch a = a0;
    if (!mr) { a.k_begin = 0; a.k_end = a.K; a.rank = 0; a.nranks = 1; }
    const int Kl = a.k_end - a.k_begin;                     // knots this launch works on
    if (mr && (a.batch > 1 || a.xcd_pack || a.nranks < 1 || a.nranks > GATO_MAX_RANKS || a.rank < 0 || a.rank >= a.nranks ||
               a.k_begin < 0 || Kl < 1 || a.k_end > a.K || (a.rank == 0) != (a.k_begin == 0) || (a.rank == a.nranks - 1) != (a.k_end == a.K))) {
        set_error("pcg_cg1(cluster): bad shard rank=%d/%d knots [%d,%d) of %d", a.rank, a.nranks, a.k_begin, a.k_end, a.K);
        return GATO_EINVAL;
    }
    const int lanes_needed = (a.knots_per_wg + 2) * S;
    // (every workgroup hands its first / last TWO blocks of w to its neighbours: two knots each, whenever the solve has neighbours)
    const bool neighbours = a.groups > 1 || (mr && a.nranks > 1);
    // the even split (knots_per_wg each, the last workgroup the rest) unless that rest is a single knot: then the first split_extra
    // workgroups take knots_per_wg, the others one less (31 knots in 6 workgroups: 6 + 5 + 5 + 5 + 5 + 5 instead of 5 x 6 + 1)
    a.split_extra = 0;
    if (neighbours && a.groups > 1 && a.knots_per_wg >= 3 && Kl - (a.groups - 1) * a.knots_per_wg == 1)
        a.split_extra = Kl - a.groups * (a.knots_per_wg - 1);
    const bool balanced = a.split_extra > 0;
    if (a.threads > MAXT || a.threads % 64 != 0 || a.threads < 4 * S || lanes_needed > a.threads || a.groups < 1 ||
        a.groups > 256 || (long long)a.groups * a.knots_per_wg < Kl || (long long)(a.groups - 1) * a.knots_per_wg >= Kl ||
        (balanced && (a.split_extra >= a.groups || a.split_extra * a.knots_per_wg + (a.groups - a.split_extra) * (a.knots_per_wg - 1) != Kl)) ||
        (neighbours && !balanced && (a.knots_per_wg < 2 || Kl - (a.groups - 1) * a.knots_per_wg < 2))) {
        set_error("pcg_cg1: bad launch geometry (K=%d groups=%d knots/wg=%d threads=%d max=%d)", Kl, a.groups,
                  a.knots_per_wg, a.threads, MAXT);
        return GATO_EINVAL;
    }
    if (a.batch > 1 && a.groups != 1) { set_error("pcg_cg1: a batch needs one workgroup per system"); return GATO_EINVAL; }
    if (a.ev_start) GATO_HIP_CHECK(hipEventRecord(a.ev_start, st));
    const int nblocks = a.batch > 1 ? a.batch : (a.xcd_pack > 0 ? 8 * ((a.groups + a.xcd_pack - 1) / a.xcd_pack) : a.groups);
    if (mr) hipLaunchKernelGGL((pcg_cg1_kernel<T, S, MAXT, true>), dim3(nblocks), dim3(a.threads), 0, st, a);
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
