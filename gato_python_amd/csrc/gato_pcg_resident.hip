// Resident PCG for gfx950: the whole preconditioned-CG solve on the block-tridiagonal Schur
// system in ONE persistent launch, matrices register-resident.
//
// Replaces parallelPCG / parallelPCG_inner (src/gato_pcg.cuh:270-470) and its helpers
// loadBlockTriDiagonal_offDiagonal / matVecMultBlockTriDiagonal (src/gato_utils.cuh:121-185),
// dotProd / reducePlus (:253-287) and the atomicAdd + grid.sync() reductions (gato_pcg.cuh:331-393).
//
// MI355X design (not the reference's one-block-of-S-threads-per-knot):
//  * lane = one row of one knot; the lane keeps its 3S entries of S and 3S entries of Pinv in
//    VGPRs for the whole solve (84 VGPRs fp32 / 168 fp64 at S=14) - the 128 MB register file of
//    the chip holds every BASELINE shape, so the hot loop touches no HBM at all.
//  * a workgroup owns a contiguous range of knots; the 3S-wide operand window [x_{k-1};x_k;x_{k+1}]
//    is read from LDS with 16-byte broadcast reads (knot stride padded to 16 B multiples).
//  * dots: in-lane product -> wave64 butterfly -> per-wave LDS partial -> fixed-order sum.
//    Deterministic, no float atomics (the reference's atomicAdd order is unspecified).
//  * one workgroup (K=50 fp32: 11 waves on one CU): no inter-workgroup traffic at all, six
//    s_barriers per iteration.
//  * several workgroups: two hand-offs per iteration (the algorithmic minimum for PCG).  Each
//    workgroup publishes [partial dot | first S-block | last S-block] of the vector it just
//    produced as 8-byte {epoch,payload} granules (write-through agent-scope stores), wave 0 of
//    every workgroup sweeps the W partials and its two neighbours' blocks until every tag equals
//    the epoch (MI355X guide: "R2: the data IS the flag").  Ghost blocks of r and p are then
//    advanced locally (ghost_r -= alpha*ghost_upsilon, ghost_p = ghost_rtilde + beta*ghost_p), so
//    the reference's four grid.sync() per iteration become two all-gathers and no barrier.
//    Granules are double-buffered by epoch parity; every spin is bounded.
#include "gato_pcg_device.h"
#include <type_traits>

namespace gato {
namespace {

template <typename T, int S, int MAXT>
struct ResidentCfg {
    static constexpr int VW = VecOf<T>::W;
    static constexpr int SP = pad_to(S, VW);           // padded knot stride in LDS (16-B multiple)
    static constexpr int MAXK = (MAXT + S - 1) / S;    // local knots incl. the partly filled one
    static constexpr int NV = SP / VW;
    static constexpr int MAXW = 256;                   // workgroups (one per CU)
    static constexpr int PM = MAXW / 64;               // partial granule loads per lane
};

// STAMP: diagnostic build only - wave 0 of workgroup 0 accumulates s_memtime deltas per segment into
// a.stamps (never used for timing claims; it perturbs the schedule).
#define GATO_STAMP(i)                                                                       \
    if (STAMP) {                                                                            \
        if (wg == 0 && wave == 0) {                                                         \
            unsigned long long t_;                                                          \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");     \
            seg[i] += t_ - t_prev;                                                          \
            t_prev = t_;                                                                    \
        }                                                                                   \
    }

// Same product with the last NL entries of the row read from LDS (16 B per lane, lane-contiguous: conflict
// free) instead of registers.  Needs S % VW == 0 and (3S-NL) % VW == 0 so that vectors never straddle blocks.
template <typename T, int S, int SP, int NL, int MAXT>
__device__ __forceinline__ T row_times_window_lds(const T (&m)[3 * S - NL], const typename VecOf<T>::type (*tail)[MAXT],
                                                  int tid, const T *xw)
{
    typedef typename VecOf<T>::type V;
    constexpr int VW = VecOf<T>::W;
    constexpr int NREG = 3 * S - NL;
    static_assert(S % VW == 0 && NREG % VW == 0 && NL % VW == 0, "vector alignment");
    T acc = (T)0;
#pragma unroll
    for (int c = 0; c < NREG; c += VW) {
        V v = *reinterpret_cast<const V *>(xw + (c / S) * SP + (c % S));
#pragma unroll
        for (int e = 0; e < VW; ++e) acc = gato::fmaT(m[c + e], v[e], acc);
    }
#pragma unroll
    for (int c = NREG; c < 3 * S; c += VW) {
        V v = *reinterpret_cast<const V *>(xw + (c / S) * SP + (c % S));
        V mv = tail[(c - NREG) / VW][tid];
#pragma unroll
        for (int e = 0; e < VW; ++e) acc = gato::fmaT(mv[e], v[e], acc);
    }
    return acc;
}

// Same product for a row whose 3S matrix entries are NOT register resident: they are loaded (L2 / Infinity Cache /
// HBM) every time, all 3S loads in flight before the first FMA.  Rows handled this way are never in the system's
// first or last block row, so no boundary entries have to be zeroed.
template <typename T, int S, int SP>
__device__ __forceinline__ T row_from_memory(const T *__restrict__ src, const T *xw, bool no_left = false, bool no_right = false)
{
    T m[3 * S];
#pragma unroll
    for (int c = 0; c < 3 * S; ++c) m[c] = src[(size_t)c * S];
    // first / last block row of the system: the left / right block is not part of the matrix (never written) - drop it
#pragma unroll
    for (int c = 0; c < S; ++c) {
        if (no_left) m[c] = (T)0;
        if (no_right) m[2 * S + c] = (T)0;
    }
    return row_times_window<T, S, SP>(m, xw);
}

// NL > 0: single-workgroup variant whose Pinv rows do not fit the register budget: the last NL entries of
// every Pinv row live in LDS (IIWA 14/7/50 in fp64: 700 rows x 84 doubles = 470 KB > the 168 VGPRs/lane that
// 11 waves on one CU leave; 24 doubles per row = 135 KB go to LDS, the rest stays in registers).
// XR > 0: SEMI-resident variant for K beyond the register file: a workgroup owns more knots than it
// has lanes for.  The first n_res-1 knots and the LAST knot of its range keep the lane = row mapping above (so the
// boundary blocks the hand-off publishes are resident rows and nothing of the hand-off changes); the knots in between
// are "extra" rows, up to XR per lane: their r and p entries live in the LDS operand windows anyway, lambda and the
// product just formed in two more LDS arrays,
// their matrix rows are re-read from memory (mostly L2 / Infinity Cache at these sizes) in every product, one row per
// trip of a plain runtime loop (unrolling it cost registers and instruction cache and ran slower).  Still ONE persistent launch with
// two hand-offs per iteration - against two launches per iteration of the streaming kernels.
// NR (with XR > 0): NO resident rows at all - every row of the workgroup's range is an "extra" row.  Without the 6S
// matrix registers per lane the workgroup can be 2-4x larger (more loads in flight per CU): the variant for shapes whose
// resident rows leave one wave per SIMD (fp64, S = 32) and for the HBM-bound end of the range.  The boundary blocks the
// hand-off publishes are then read from the product array in LDS instead of from lane registers.
// MR: cluster launch (gato_cluster_pcg) - this kernel is ONE RANK of a solve whose knots are sharded over the GPUs of a
// node (the reference is single-device, gato_utils.cuh:831).  The hand-off gets a second level:
// after the workgroups of this GPU have gathered their partials (level 1, unchanged), workgroup 0 stores the rank's
// total into EVERY rank's mirror (peer-mapped fine-grained memory: xGMI peer stores, system scope), the rank's first /
// last workgroup store their boundary block into the left / right neighbour rank's mirror, and wave 0 of every
// workgroup polls its OWN GPU's mirror until the R totals (and, at the rank's edges, the neighbour's block) carry the
// epoch; totals are summed in rank order (identical on every rank => identical exit decision).  The grid barriers of
// the reference (gato_pcg.cuh:363,378,393,428) thus become two device-initiated all-gathers per iteration across the
// node, no host involvement, no collective library inside the loop.
// WP: launches of 2..32 workgroups of the plain variant.  EVERY WAVE publishes its own partial (one granule per wave in line 0
// of the workgroup's slot - still one writing workgroup per line) and the polling wave of every workgroup reads W x waves
// granules: the same W lines as before, coalesced.  The gather of the workgroup's total in wave 0 (LDS write, barrier B1, LDS
// read, second DPP sum) leaves the critical path of the hand-off and an iteration has four barriers instead of six.  (Not the
// "every wave polls" form that DESIGN.md 3.1 records as a dead end: one polling wave per workgroup it stays.)  The ghost
// blocks of r and p live in registers of the polling lanes (lanes 0..S-1 left, 32..32+S-1 right) - no staging array, no LDS
// read-modify-write on the way to the next product.  A compile-time variant: as a run-time switch in the one kernel the
// extra scalar paths cost every launch 3-5 % (measured, same box: 14/7/512 f32 2.93 -> 3.02 us per iteration).
// WPM = poll loads per lane of that form (0 = the gathered form): 4 serves W << ceil(log2(waves)) <= 256 granules (up to 32
// workgroups of 8 waves).  Measured and rejected: 16 loads per lane for up to 128 workgroups (14/7/4096 f32, W = 114: 4.77 us
// per iteration against 3.87 gathered - fifteen load instructions per sweep cost more than the gather they replace).
// DR: DPP-row layout (gato_pcg_device.h: row_times_dpp) - a knot owns whole 16-lane DPP rows and the products read their operand
// window from the neighbouring lanes' REGISTERS (v_fmac_*_dpp row_newbcast) instead of 16-byte LDS reads: the LDS-window
// products are bound by the LDS return path (fp64 14/7: 1.13 us of a 3.56 us iteration at 15 workgroups, 21 16-byte reads per
// lane and product), this form reads two scalars per lane and product.  Same summation order per row: identical bits.
// Plain and cluster launches (NL = 0, XR = 0); lanes S..15 of a row idle (S = 14: 32 knots per 512 threads instead of 36).
template <typename T, int S, int MAXT, int NL = 0, int DIAG = 0, int XR = 0, bool NR = false, bool MR = false, int WPM = 0, bool DR = false>
__global__ __launch_bounds__(MAXT) void pcg_resident_kernel(PcgLaunch a)
{
    static_assert(!DR || (NL == 0 && XR == 0 && !NR && DIAG != 1 && DppRows<S>::ok), "DPP-row layout: plain and cluster variants");
    constexpr int LPK = DR ? DppRows<S>::lanes : S;                            // lanes per knot
    constexpr bool WP = WPM > 0;            // per-wave published partials
    constexpr bool RG = WPM != 0;           // ghost blocks in the polling lanes' registers (WPM = -1: that alone, gathered partials)
    static_assert(!RG || (NL == 0 && XR == 0 && !NR && DIAG != 1), "wave-published partials / register ghosts: plain and cluster variants");
    // DIAG: 0 = production; 1 = cycle stamps + the timing-only switches of a.ablate; 2 = the switches alone (what
    // bench.py's latency floor times: the stamps cost registers, and this instantiation has none to spare)
    constexpr bool STAMP = DIAG == 1, ABL = DIAG != 0;
    typedef ResidentCfg<T, S, MAXT> Cfg;
    typedef Granule<T> Gr;
    typedef GranuleXcd<T> LGr;
    typedef GranuleSys<T> XGr;
    constexpr int SP = Cfg::SP;
    constexpr int GPV = Gr::GPV;
    // hand-off layout invariant (DESIGN.md 3.1 dead end 2: granules of two writers in one 128-B line get lost across
    // XCDs): a workgroup's slot is a whole number of 128-B lines (16 granules), the partial has line 0 to itself
    static_assert(MAXT % 64 == 0 && 2 * S * GPV <= 16 * ((2 * S * GPV + 15) / 16), "slot layout");
    static_assert(!MR || (NL == 0 && DIAG == 0), "cluster launches use the plain and the semi-resident variants");
    static_assert(DIAG == 0 || !MR, "diagnostic builds are single-GPU");

    constexpr int MAXKX = NR ? Cfg::MAXK * XR : Cfg::MAXK * (1 + XR);          // local knots incl. the extra ones
    static_assert(NL == 0 || XR == 0, "the LDS-tail variant is single-workgroup only");
    static_assert(!NR || (XR > 0 && 2 * S <= 64), "NR: every row is an extra row; wave 0 publishes both boundary blocks");
    __shared__ __attribute__((aligned(16))) T xs[2][(MAXKX + 2) * SP];        // [0] = p window, [1] = r window
    __shared__ __attribute__((aligned(32))) T wpart[2][4 * ((MAXT + 63) / 64)];   // per-wave, per-row partial dots, double-buffered by epoch parity
    typedef typename VecOf<T>::type V;
    constexpr int NREG = 3 * S - NL;
    __shared__ __attribute__((aligned(16))) V ptail[NL > 0 ? NL / VecOf<T>::W : 1][NL > 0 ? MAXT : 1];
    __shared__ T gh[2][32];          // ghost blocks of the vector just gathered: [0] left, [1] right
    __shared__ T bc[2];              // broadcast scalars
    __shared__ int s_abort;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int nwaves = blockDim.x >> 6;
    // batch > 1: one workgroup per independent system (blockIdx.x = system), no inter-workgroup traffic
    const bool batched = a.batch > 1;
    // xcd_pack = X in 1..7: blocks are dealt round-robin over the 8 XCDs, so with an oversubscribed grid of 8*per blocks
    // of which only those with blockIdx % 8 < X work, the W working groups sit on X XCDs, neighbouring knot ranges on
    // the same one (hand-offs inside an XCD are 15-20 % faster).  A speed hint only: nothing below depends on where a
    // block really runs.
    const int X = a.xcd_pack;
    const int xres = X > 0 ? (int)((blockIdx.x - (unsigned)a.xcd_sel) & 7) : 0;      // a.xcd_sel: which XCD(s) of the eight host the working blocks
    if (X > 0 && xres >= X) return;
    const int per_x = X > 0 ? (int)(gridDim.x >> 3) : 0;
    const int wg = batched ? 0 : (X > 0 ? xres * per_x + (int)(blockIdx.x >> 3) : (int)blockIdx.x);
    const int W = (NL > 0 || batched) ? 1 : (X > 0 ? a.groups : (int)gridDim.x);
    if (X > 0 && wg >= W) return;
    const size_t sys = batched ? blockIdx.x : 0;
    const int K = a.K;
    // this launch's knot range: the whole system, or this rank's shard of it (MR)
    const int k_begin = MR ? a.k_begin : 0, k_end = MR ? a.k_end : K;
    const int R = MR ? a.nranks : 1;
    const int k0 = k_begin + wg * a.knots_per_wg;
    const int nk = min(a.knots_per_wg, k_end - k0);
    const int jl = tid / LPK;              // the lane's slot
    const int r_ = tid - jl * LPK;         // row inside the knot (DR: rows S..LPK-1 do not exist, those lanes idle)
    const int n_res = NR ? 0 : (XR > 0 ? min(nk, (int)blockDim.x / S) : nk);    // knots with lanes of their own
    const int n_ext = nk - n_res;                                    // knots handled as extra rows (XR > 0 only)
    const int j = (XR > 0 && !NR && n_ext > 0 && jl == n_res - 1) ? nk - 1 : jl;   // local knot: the last slot holds the LAST knot
    const bool active = jl < n_res && (!DR || r_ < S);
    const int xk = NR ? 0 : n_res - 1;                               // first local knot of the extra rows
    // extra rows of this lane: rows q = tid + e * blockDim.x (e < ne) of the knots [xk, xk + n_ext)
    const int n_ext_rows = n_ext * S;
    const int ne = XR > 0 ? (n_ext_rows + (int)blockDim.x - 1) / (int)blockDim.x : 0;      // workgroup-uniform trip count
    __shared__ T xst[2][XR > 0 ? XR * MAXT : 1];                                            // [lambda | product][row]; r, p: the windows
    const int k = k0 + j;
    const bool has_left = k0 > 0;          // a neighbouring block row exists in the SYSTEM ...
    const bool has_right = k0 + nk < K;
    const bool loc_left = MR ? wg > 0 : has_left;            // ... and it belongs to a workgroup of this launch,
    const bool loc_right = MR ? wg < W - 1 : has_right;
    const bool x_left = MR && wg == 0 && has_left;           // or to the neighbouring rank (another GPU)
    const bool x_right = MR && wg == W - 1 && has_right;
    const bool multi = W > 1 || (MR && R > 1);               // ghost blocks exist and travel through the hand-off

    const T *__restrict__ dS = static_cast<const T *>(a.S_bd) + sys * 3 * S * S * K;
    const T *__restrict__ dP = static_cast<const T *>(a.P_bd) + sys * 3 * S * S * K;
    const T *__restrict__ dG = static_cast<const T *>(a.gamma) + sys * S * K;
    T *__restrict__ dL = static_cast<T *>(a.lambda) + sys * S * K;

    // ---- load this lane's rows of S and Pinv into registers (once per solve) ----------------
    // bd layout: block-row k = [left|main|right], each S*S column-major -> element (r, c) of the
    // S x 3S strip sits at c*S + r (gato_utils.cuh:53-54,97-98).  First/last block rows have no
    // left/right block (gato_utils.cuh:157-174): those entries are forced to zero here.
    T sm[NR ? 1 : 3 * S], pm[NR ? 1 : NREG];
    if constexpr (!NR) {
        const size_t base = (size_t)(active ? k : 0) * 3 * S * S + r_;
#pragma unroll
        for (int c = 0; c < 3 * S; ++c) {
            const bool ok = active && !(k == 0 && c < S) && !(k == K - 1 && c >= 2 * S);
            sm[c] = ok ? dS[base + (size_t)c * S] : (T)0;
            const T pv_ = ok ? dP[base + (size_t)c * S] : (T)0;
            if (c < NREG) pm[c < NREG ? c : 0] = pv_;
            else ptail[(c - NREG) / VecOf<T>::W][tid][(c - NREG) % VecOf<T>::W] = pv_;   // own lane only: no barrier
        }
        // (issuing all 6S loads first and selecting afterwards - what pays in the one-workgroup kernels below - measured no
        //  better here: 14/7/1024 f32 2.34 -> 2.31 but 14/7/4096 f32 3.44 -> 3.53, 32/16/256 2.53 -> 2.62 us per iteration)
    }

    // ---- hand-off area ----------------------------------------------------------------------
    const int slotG = pcg_slot_granules(S, (int)sizeof(T));
    gu64 *slots = (gu64 *)a.slots;
    gi32 *g_status = (gi32 *)a.status;
    const unsigned long long t_limit = a.timeout_ticks;
    // cross-GPU mirror (MR): granules per parity, ghost block offsets
    const int xslotG = pcg_xslot_granules(S, (int)sizeof(T));
    const int xghL = 16 * GATO_MAX_RANKS, xghR = xghL + pcg_xghost_granules(S, (int)sizeof(T));
    unsigned xepoch = MR ? a.xepoch0 : 0u;
    // mirrors of the peers, read once from the device table: the neighbouring ranks' (edge blocks) and, in lane r of
    // wave 0 of workgroup 0, rank r's (the rank total goes to every rank)
    gu64 *xp_prev = nullptr, *xp_next = nullptr;
    __shared__ unsigned long long s_xpeer[MR ? GATO_MAX_RANKS : 1];      // the peers' mirrors: lane r of wave 0 fetches rank r's at each hand-off
    if constexpr (MR) {
        if (a.rank > 0) xp_prev = (gu64 *)a.xpeer[a.rank - 1];
        if (a.rank < R - 1) xp_next = (gu64 *)a.xpeer[a.rank + 1];
        if (wave == 0 && lane < R) s_xpeer[lane] = (unsigned long long)a.xpeer[lane];   // read back by the same lanes only
    }

    if (tid == 0) s_abort = 0;     // the status word is never cleared here: the host matches launch ids (gato_pcg_status)
    // test hook, diagnostic build only (options stamp_pcg + ablate = 16): the last workgroup never shows up, as if it
    // had not been scheduled - the others must give up after the time-out and report it
    if (ABL && (a.ablate & 16) && W > 1 && wg == W - 1) return;
    for (int i = tid; i < 2 * (MAXKX + 2) * SP; i += blockDim.x) (&xs[0][0])[i] = (T)0;
    __syncthreads();

    // r = gamma, lambda = 0 (gato_pcg.cuh:300-304); ghost r read straight from gamma.
    T lam = (T)0;
    T r = active ? dG[(size_t)k * S + r_] : (T)0;
    T p = (T)0, ups, rt;
#pragma unroll 1
    for (int e = 0; e < ne; ++e) {
        const int q = tid + e * (int)blockDim.x;
        if (q < n_ext_rows) {
            const int jx = xk + q / S, rx = q % S;
            const T g_ = dG[(size_t)(k0 + jx) * S + rx];
            xst[0][q] = (T)0; xst[1][q] = (T)0;
            xs[1][(jx + 1) * SP + rx] = g_;
        }
    }
    // product = M x on the extra rows (x = window w: 0 = p, 1 = r); returns this lane's share of x . (M x)
    auto extra_rows = [&](const T *__restrict__ M, int w) -> T {
        T dot = (T)0;
        if constexpr (NR && S % 2 == 0) {
            // No resident rows: a lane takes TWO adjacent rows of a knot per trip.  Element (r, c) of a block row sits at
            // c*S + r, so the pair (r, r+1) of a column is ONE 8-byte (f32) / 16-byte (f64) load: half the load
            // instructions and half the L1 sector accesses per byte (the 4-byte-per-lane form keeps the L1 at 83 % of
            // its 64 B/clk - DESIGN.md 3.1), and both rows share the operand-window reads (packed FMAs in f32).
            typedef T T2 __attribute__((ext_vector_type(2)));
            typedef typename VecOf<T>::type V;
            constexpr int VW = VecOf<T>::W, H = S / 2;
            const int n_pairs = n_ext * H;
            const int ne2 = (n_pairs + (int)blockDim.x - 1) / (int)blockDim.x;
#pragma unroll 1
            for (int e = 0; e < ne2; ++e) {
                const int qp = tid + e * (int)blockDim.x;
                const bool on = qp < n_pairs;
                const int qq = on ? qp : 0;
                const int jx = xk + qq / H, r0 = 2 * (qq % H);
                const T *__restrict__ src = M + (size_t)(k0 + jx) * 3 * S * S + r0;
                T2 m[3 * S];
#pragma unroll
                for (int c = 0; c < 3 * S; ++c) m[c] = *reinterpret_cast<const T2 *>(src + (size_t)c * S);
                const bool nl = k0 + jx == 0, nr = k0 + jx == K - 1;        // first / last block row of the system
#pragma unroll
                for (int c = 0; c < S; ++c) {
                    if (nl) m[c] = T2{(T)0, (T)0};
                    if (nr) m[2 * S + c] = T2{(T)0, (T)0};
                }
                const T *xw = &xs[w][jx * SP];
                T2 acc = {(T)0, (T)0};
#pragma unroll
                for (int b = 0; b < 3; ++b) {
#pragma unroll
                    for (int i = 0; i < SP / VW; ++i) {
                        const V v = *reinterpret_cast<const V *>(xw + b * SP + i * VW);
#pragma unroll
                        for (int e2 = 0; e2 < VW; ++e2)
                            if (i * VW + e2 < S) acc = __builtin_elementwise_fma(m[b * S + i * VW + e2], T2{v[e2], v[e2]}, acc);
                    }
                }
                if (on) {
                    const int q0 = (jx - xk) * S + r0;
                    xst[1][q0] = acc[0];
                    xst[1][q0 + 1] = acc[1];
                    dot = gato::fmaT(xw[SP + r0], acc[0], dot);
                    dot = gato::fmaT(xw[SP + r0 + 1], acc[1], dot);
                }
            }
            return dot;
        }
#pragma unroll 1
        for (int e = 0; e < ne; ++e) {
            const int q = tid + e * (int)blockDim.x;
            const bool on = q < n_ext_rows;
            const int qq = on ? q : 0;                                      // lanes without a row here read row 0
            const int jx = xk + qq / S, rx = qq % S;
            const T y = row_from_memory<T, S, SP>(M + (size_t)(k0 + jx) * 3 * S * S + rx, &xs[w][jx * SP],
                                                  NR && k0 + jx == 0, NR && k0 + jx == K - 1);
            if (on) {
                xst[1][q] = y;
                dot = gato::fmaT(xs[w][(jx + 1) * SP + rx], y, dot);
            }
        }
        return dot;
    };
    if (active) xs[1][(j + 1) * SP + r_] = r;
    if (tid < S) {
        if (has_left) xs[1][tid] = dG[(size_t)(k0 - 1) * S + tid];
    } else if (tid < 2 * S) {
        if (has_right) xs[1][(nk + 1) * SP + (tid - S)] = dG[(size_t)(k0 + nk) * S + (tid - S)];
    }
    __syncthreads();

    T g_r_init = (T)0;
    if constexpr (RG) {                     // ghost r starts as the neighbours' gamma blocks (just written to the window)
        if (wave == 0 && (lane < S || (lane >= 32 && lane < 32 + S))) g_r_init = xs[1][lane < 32 ? lane : (nk + 1) * SP + (lane - 32)];
    }
    unsigned long long seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = 0, t_begin = 0, rt_begin = 0;
    if (STAMP) {
        t_begin = t_prev = __builtin_amdgcn_s_memtime();
        rt_begin = __builtin_amdgcn_s_memrealtime();
    }
    unsigned epoch = a.epoch0;
    T eta = (T)0, eta_new = (T)0;
    int iters = a.max_iters;
    const T tol = (T)a.exit_tol;
    bool aborted = false;

    // One reduction + halo exchange.  `val` = the vector just produced (upsilon or r~), `prod` the
    // lane's dot contribution.  On return: total in every thread; gh[][] = neighbours' boundary
    // blocks of `val` (zeros where there is no neighbour).
    const int abl = ABL ? a.ablate : 0;   // diagnostic timing-only switches, compiled out of the production build
    // One-XCD launches (xcd_pack): once the workgroups have verified - below, with agent-scope granules - that they really all
    // sit on ONE XCD, the hand-off granules are stored with WORKGROUP scope: they stay in that XCD's L2 (its CUs' sc1 loads see
    // them there) instead of being written through to memory.  A placement that is not what the launch hoped for (the
    // dispatcher is free to place blocks anywhere) just keeps the agent-scope stores.
    bool fast_st = false;
    auto gstore = [&](gu64 *g, unsigned ep, T v) {
        if (fast_st) LGr::store(g, ep, v);
        else Gr::store(g, ep, v);
    };
    // WP: the ghost blocks of r and p and the boundary entry just gathered, in the polling lanes' registers
    T hv_reg = (T)0, g_r = g_r_init, g_p = (T)0;
    const bool g_lane = RG && wave == 0 && (lane < S || (lane >= 32 && lane < 32 + S));
    const int gslot = lane < 32 ? lane : (nk + 1) * SP + (lane - 32);      // the lane's ghost entry in an operand window
    auto allreduce_and_halo = [&](T val, T prod, T &total) {
        ++epoch;
        if constexpr (MR) ++xepoch;
        if (abl & 4) { total = (T)1 + prod * (T)1e-30; return; }
        if (aborted) { total = (T)0; return; }             // the placement round below already timed out: no second wait
        T *wp = wpart[epoch & 1];
        gu64 *mine = slots + ((size_t)(epoch & 1) * W + wg) * slotG;
        if constexpr (WP) {
            if (W > 1) {
                const T ws = wave_sum(prod);
                if (lane == 0) gstore(mine + wave * GPV, epoch, ws);
            } else partials_store(wp, wave, lane, prod);      // one workgroup per rank (cluster): its total comes from LDS
        } else partials_store(wp, wave, lane, prod);
        if (!NR && W > 1 && active) {
            if (j == 0) gstore(mine + 16 + r_ * GPV, epoch, val);
            if (j == nk - 1) gstore(mine + 16 + (S + r_) * GPV, epoch, val);
        }
        if constexpr (MR && !NR) {          // the rank's edge blocks go straight into the neighbouring GPU's mirror
            if (active) {
                if (x_left && j == 0) XGr::store(xp_prev + (size_t)(xepoch & 1) * xslotG + xghR + r_ * GPV, xepoch, val);
                if (x_right && j == nk - 1) XGr::store(xp_next + (size_t)(xepoch & 1) * xslotG + xghL + r_ * GPV, xepoch, val);
            }
        }
        if (!WP || W == 1) __syncthreads();                                    // B1
        if constexpr (NR) {                 // boundary blocks of the vector just formed: from the product array (complete after B1)
            if (W > 1) {
                if (tid < S) gstore(mine + 16 + tid * GPV, epoch, xst[1][tid]);
                else if (tid < 2 * S) gstore(mine + 16 + tid * GPV, epoch, xst[1][(nk - 1) * S + (tid - S)]);
            }
            if constexpr (MR) {
                if (x_left && tid < S) XGr::store(xp_prev + (size_t)(xepoch & 1) * xslotG + xghR + tid * GPV, xepoch, xst[1][tid]);
                if (x_right && tid >= S && tid < 2 * S)
                    XGr::store(xp_next + (size_t)(xepoch & 1) * xslotG + xghL + (tid - S) * GPV, xepoch, xst[1][(nk - 1) * S + (tid - S)]);
            }
        }
        if (W == 1 && !(MR && R > 1)) {
            // one workgroup: every wave sums the per-wave partials itself (fixed order), no second barrier
            // (one LDS read per lane + a DPP sum: a serial loop over the partials would pay one LDS
            //  round trip per wave)
            total = partials_total<T, (MAXT <= 512 ? 8 : 16)>(wp, nwaves, lane);
            return;
        }
        if (wave == 0) {
            T tot = (T)0;
            if (!WP || W == 1) tot = partials_total<T, (MAXT <= 512 ? 8 : 16)>(wp, nwaves, lane);
            bool fail = false;
            if (W > 1) {
                if constexpr (!WP) {
                    if (lane == 0) gstore(mine, epoch, tot);
                }
                // sweep: partials of all workgroups + neighbours' halo blocks.  Every lane issues ALL its loads
                // back to back from clamped (always valid) addresses and waits once: predicated loads would each
                // get their own s_waitcnt, i.e. one L2 round trip after the other.
                gu64 *pbase = slots + (size_t)(epoch & 1) * W * slotG;
                // The per-lane addresses are re-derived from the lane id at every hand-off, behind an empty asm the compiler
                // cannot see through: as loop invariants it computes them once for both parities, runs out of registers
                // and re-loads them from scratch at the head of every hand-off (a memory round trip on the critical path).
                int ln = lane;
                asm volatile("" : "+v"(ln));
                const bool want_l = loc_left && ln < S;
                const bool want_r = loc_right && ln >= 32 && ln < 32 + S;
                gu64 *hptr = want_l ? pbase + (size_t)(wg - 1) * slotG + 16 + (S + ln) * GPV
                           : want_r ? pbase + (size_t)(wg + 1) * slotG + 16 + (ln - 32) * GPV
                                    : mine;
                // gathered form: entry e = workgroup e's total.  WP: entry e = wave (e & mask) of workgroup (e >> wsh); entries of
                // waves that do not exist read the workgroup's wave 0 and count as zero
                constexpr int PMX = WP ? WPM : Cfg::PM;
                gu64 *pptr[PMX];
                int pm_count = (W + 63) >> 6;                 // wave-uniform
                int wsh = 0;
                if constexpr (WP) {
                    wsh = nwaves <= 1 ? 0 : 32 - __builtin_clz((unsigned)(nwaves - 1));
                    pm_count = ((W << wsh) + 63) >> 6;
                }
#pragma unroll
                for (int m = 0; m < PMX; ++m) {
                    if constexpr (WP) {
                        const int e = ln + 64 * m, wi = e >> wsh, wv = e & ((1 << wsh) - 1);
                        pptr[m] = pbase + (size_t)min(wi, W - 1) * slotG + (wv < nwaves ? wv : 0) * GPV;
                    } else pptr[m] = pbase + (size_t)min(ln + 64 * m, W - 1) * slotG;
                }
                unsigned long long raw[PMX][GPV], hraw[GPV];
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                // Cross-XCD launches: the first poll can never hit (the publishers' stores need a fabric round
                // trip), and W*W early loads only queue in front of those stores.  ~0.35 us of sleep before the
                // first poll measured -5..-10 % per iteration for W > 32 and +6 % for one-XCD launches.
                if (W > 64) __builtin_amdgcn_s_sleep(12);
                else if (W > 32) __builtin_amdgcn_s_sleep(10);      // round 3 sweep: 32/16/1024 (W = 64) 10: 4.22 / 12: 4.32 us, 14/7/2048 (W = 57) 3.51 / 3.61
                for (unsigned spin = 0;; ++spin) {
#pragma unroll
                    for (int m = 0; m < PMX; ++m) {
                        if (m < pm_count) {
#pragma unroll
                            for (int g = 0; g < GPV; ++g)
                                raw[m][g] = __hip_atomic_load(pptr[m] + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
#pragma unroll
                    for (int g = 0; g < GPV; ++g)
                        hraw[g] = __hip_atomic_load(hptr + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bool ok = true;
#pragma unroll
                    for (int m = 0; m < PMX; ++m) {
                        if (m < pm_count) {
#pragma unroll
                            for (int g = 0; g < GPV; ++g) ok &= (unsigned)(raw[m][g] >> 32) == epoch;
                        }
                    }
#pragma unroll
                    for (int g = 0; g < GPV; ++g) ok &= (unsigned)(hraw[g] >> 32) == epoch;   // own slot: always current
                    if (__all(ok)) break;
                    if ((spin & 255u) == 255u) {
                        const bool late = __builtin_amdgcn_s_memrealtime() - t0 > t_limit;
                        const bool other = __hip_atomic_load(g_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.launch_id;
                        if (late || other) { fail = true; break; }
                    }
                }
                T pv[PMX];
#pragma unroll
                for (int m = 0; m < PMX; ++m) {
                    bool on = m < pm_count && lane + 64 * m < W;
                    if constexpr (WP) {
                        const int e = lane + 64 * m;
                        on = m < pm_count && (e >> wsh) < W && (e & ((1 << wsh) - 1)) < nwaves;
                    }
                    pv[m] = on ? Gr::decode(raw[m]) : (T)0;
                }
                const T hv = Gr::decode(hraw);
                T acc = (T)0;
#pragma unroll
                for (int m = 0; m < PMX; ++m) acc += pv[m];
                tot = wave_sum(acc);
                if constexpr (RG) hv_reg = (want_l || want_r) ? hv : (T)0;
                else {
                    if (lane < S) gh[0][lane] = want_l ? hv : (T)0;
                    if (lane >= 32 && lane < 32 + S) gh[1][lane - 32] = want_r ? hv : (T)0;
                }
            } else {                       // one workgroup on this GPU (cluster launch): the ghosts come from level 2 only
                if constexpr (RG) hv_reg = (T)0;
                else {
                    if (lane < S) gh[0][lane] = (T)0;
                    if (lane >= 32 && lane < 32 + S) gh[1][lane - 32] = (T)0;
                }
            }
            if constexpr (MR) {
                if (R > 1 && !fail) {
                    // ---- level 2: across the GPUs of the node.  tot = this rank's total (identical in all its workgroups)
                    const size_t xo = (size_t)(xepoch & 1) * xslotG;
                    if (wg == 0 && lane < R) XGr::store((gu64 *)s_xpeer[lane] + xo + a.rank * 16, xepoch, tot);
                    gu64 *xl = (gu64 *)a.xslots + xo;                     // polls stay on THIS GPU's memory
                    int l2 = lane;
                    asm volatile("" : "+v"(l2));
                    const bool xw_l = x_left && l2 < S;
                    const bool xw_r = x_right && l2 >= 32 && l2 < 32 + S;
                    gu64 *tptr = xl + (size_t)min(l2, R - 1) * 16;
                    gu64 *xhp = xw_l ? xl + xghL + l2 * GPV : xw_r ? xl + xghR + (l2 - 32) * GPV : tptr;
                    unsigned long long traw[GPV], xraw[GPV];
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    for (unsigned spin = 0;; ++spin) {
#pragma unroll
                        for (int g = 0; g < GPV; ++g) {
                            traw[g] = __hip_atomic_load(tptr + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            xraw[g] = __hip_atomic_load(xhp + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                        bool ok = true;
#pragma unroll
                        for (int g = 0; g < GPV; ++g) ok &= (unsigned)(traw[g] >> 32) == xepoch && (unsigned)(xraw[g] >> 32) == xepoch;
                        if (__all(ok)) break;
                        if ((spin & 255u) == 255u) {
                            const bool late = __builtin_amdgcn_s_memrealtime() - t0 > t_limit;
                            const bool other = __hip_atomic_load(g_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.launch_id;
                            if (late || other) { fail = true; break; }
                        }
                    }
                    tot = partials_sum(lane < R ? XGr::decode(traw) : (T)0);     // rank order, the same tree on every GPU
                    const T xv = XGr::decode(xraw);
                    if constexpr (RG) {
                        if (xw_l || xw_r) hv_reg = xv;
                    } else {
                        if (xw_l) gh[0][lane] = xv;
                        if (xw_r) gh[1][lane - 32] = xv;
                    }
                }
            }
            if (fail) {
                if (lane == 0) {
                    __hip_atomic_store(g_status, a.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_abort = 1;
                }
            }
            if (lane == 0) bc[epoch & 1] = tot;
        }
        __syncthreads();                                                       // B2
        total = bc[epoch & 1];
        aborted = s_abort != 0;
    };

    // Flat cluster exchange (MR, a.flat): ONE level across the node.  Every workgroup of every rank has a slot in every
    // mirror (global workgroup index gw = a.flat_base + wg of a.flat_groups); it stores its partial into ALL mirrors (lane r
    // -> rank r) and its boundary blocks into its own GPU's mirror and, at the rank's edges, the neighbour's; wave 0 polls
    // the partials of all workgroups and its two neighbours' blocks in ITS OWN GPU's mirror.  Same sum order on every
    // workgroup of every rank.  Against the two-level form this saves the wait for the rank's own gather before anything
    // crosses the fabric (a hand-off costs one fabric store + one poll instead of level 1 + that).
    auto allreduce_flat = [&](T val, T prod, T &total) {
        if constexpr (MR) {
            ++epoch; ++xepoch;
            T *wp = wpart[xepoch & 1];
            partials_store(wp, wave, lane, prod);
            const int WT = a.flat_groups, gw = a.flat_base + wg;
            const size_t so = a.flat_off + ((size_t)(xepoch & 1) * WT + gw) * slotG;       // this workgroup's slot in a mirror
            gu64 *fl = (gu64 *)a.xslots;
            if (!NR && active) {
                if (j == 0) {
                    XGr::store(fl + so + 16 + r_ * GPV, xepoch, val);
                    if (x_left) XGr::store(xp_prev + so + 16 + r_ * GPV, xepoch, val);
                }
                if (j == nk - 1) {
                    XGr::store(fl + so + 16 + (S + r_) * GPV, xepoch, val);
                    if (x_right) XGr::store(xp_next + so + 16 + (S + r_) * GPV, xepoch, val);
                }
            }
            __syncthreads();                                                   // B1
            if constexpr (NR) {
                if (tid < 2 * S) {
                    const T v2 = tid < S ? xst[1][tid] : xst[1][(nk - 1) * S + (tid - S)];
                    XGr::store(fl + so + 16 + tid * GPV, xepoch, v2);
                    if (tid < S && x_left) XGr::store(xp_prev + so + 16 + tid * GPV, xepoch, v2);
                    if (tid >= S && x_right) XGr::store(xp_next + so + 16 + tid * GPV, xepoch, v2);
                }
            }
            if (wave == 0) {
                T tot = partials_total<T, (MAXT <= 512 ? 8 : 16)>(wp, nwaves, lane);
                if (lane < R) XGr::store((gu64 *)s_xpeer[lane] + so, xepoch, tot);   // the partial goes into EVERY mirror
                gu64 *pbase = fl + a.flat_off + (size_t)(xepoch & 1) * WT * slotG;
                int ln = lane;                                 // re-derived at every hand-off (see allreduce_and_halo)
                asm volatile("" : "+v"(ln));
                const bool want_l = has_left && ln < S;
                const bool want_r = has_right && ln >= 32 && ln < 32 + S;
                gu64 *hptr = want_l ? pbase + (size_t)(gw - 1) * slotG + 16 + (S + ln) * GPV
                           : want_r ? pbase + (size_t)(gw + 1) * slotG + 16 + (ln - 32) * GPV
                                    : pbase + (size_t)gw * slotG;
                gu64 *pptr[Cfg::PM];
#pragma unroll
                for (int m = 0; m < Cfg::PM; ++m) pptr[m] = pbase + (size_t)min(ln + 64 * m, WT - 1) * slotG;
                const int pm_count = (WT + 63) >> 6;
                unsigned long long raw[Cfg::PM][GPV], hraw[GPV];
                // sleep before the first sweep as in the single-GPU launches (l_sleep below): the partials of WT workgroups cross
                // the XCDs' fabric at least; across GPUs the peers' stores take longer still
                if (WT > 32) {
                    const int sl = 10 + WT / 22;
                    for (int i = 0; i < sl; ++i) __builtin_amdgcn_s_sleep(1);
                }
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                bool fail = false;
                for (unsigned spin = 0;; ++spin) {
#pragma unroll
                    for (int m = 0; m < Cfg::PM; ++m) {
                        if (m < pm_count) {
#pragma unroll
                            for (int g = 0; g < GPV; ++g)
                                raw[m][g] = __hip_atomic_load(pptr[m] + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        }
                    }
#pragma unroll
                    for (int g = 0; g < GPV; ++g) hraw[g] = __hip_atomic_load(hptr + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    bool ok = true;
#pragma unroll
                    for (int m = 0; m < Cfg::PM; ++m) {
                        if (m < pm_count) {
#pragma unroll
                            for (int g = 0; g < GPV; ++g) ok &= (unsigned)(raw[m][g] >> 32) == xepoch;
                        }
                    }
#pragma unroll
                    for (int g = 0; g < GPV; ++g) ok &= (unsigned)(hraw[g] >> 32) == xepoch;
                    if (__all(ok)) break;
                    if ((spin & 255u) == 255u) {
                        const bool late = __builtin_amdgcn_s_memrealtime() - t0 > t_limit;
                        const bool other = __hip_atomic_load(g_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.launch_id;
                        if (late || other) { fail = true; break; }
                    }
                }
                T acc = (T)0;
#pragma unroll
                for (int m = 0; m < Cfg::PM; ++m)
                    acc += (m < pm_count && lane + 64 * m < WT) ? XGr::decode(raw[m]) : (T)0;
                const T hv = XGr::decode(hraw);
                tot = wave_sum(acc);
                if constexpr (RG) hv_reg = (want_l || want_r) ? hv : (T)0;
                else {
                    if (lane < S) gh[0][lane] = want_l ? hv : (T)0;
                    if (lane >= 32 && lane < 32 + S) gh[1][lane - 32] = want_r ? hv : (T)0;
                }
                if (fail && lane == 0) {
                    __hip_atomic_store(g_status, a.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_abort = 1;
                }
                if (lane == 0) bc[xepoch & 1] = tot;
            }
            __syncthreads();                                                   // B2
            total = bc[xepoch & 1];
            aborted = s_abort != 0;
        }
    };
    // ---- LEAN hand-off: the plain single-GPU launches (RG forms).  Same protocol and same arithmetic as allreduce_and_halo,
    // with everything that does not change between hand-offs computed ONCE: per-lane 32-bit BYTE offsets inside a parity block
    // of the hand-off area (stores and loads take the uniform block base in SGPRs plus that offset - no 64-bit address
    // arithmetic per hand-off; the loads of one sweep differ by a uniform stride, so they share ONE offset register), the
    // predicates of the storing / decoding lanes, and the poll instantiated per number of loads.  FAST (workgroup-scope stores,
    // one-XCD launches that verified their placement) is a compile-time argument: the iteration loop exists twice instead of
    // branching at every store.  The stamps of a diagnostic build said why: of a hand-off's ~2 300 cycles at 15 workgroups
    // the poll itself was 970, the rest address arithmetic, scalar branches and two wave sums in the polling wave.
    constexpr bool LEAN = RG && !MR;
    // KEEP: the polling lanes' load offsets stay in registers for the whole solve; kernels whose matrix rows leave few registers
    // (fp64 at S = 14, fp32 at S = 32, the 768-thread bound) re-derive them from the lane id at every hand-off instead (a
    // dozen vector instructions in the polling wave; spilling them costs a memory round trip on the critical path)
    constexpr int REGCAP = MAXT <= 256 ? 512 : MAXT <= 512 ? 256 : MAXT <= 768 ? 168 : 128;
    constexpr bool KEEP = REGCAP - 6 * S * (int)(sizeof(T) / 4) >= 120;
    const int l_wsh = WP ? (nwaves <= 1 ? 0 : 32 - __builtin_clz((unsigned)(nwaves - 1))) : 0;
    const unsigned l_slotB = (unsigned)slotG * 8u;
    const unsigned l_st_part = (unsigned)wg * l_slotB + (WP ? (unsigned)__builtin_amdgcn_readfirstlane(wave) * (unsigned)GPV * 8u : 0u);   // uniform
    const unsigned l_st_halo = (unsigned)wg * l_slotB + (16u + (unsigned)r_ * (unsigned)GPV) * 8u;     // first block; the last block S values further
    const bool l_hl = LEAN && active && j == 0, l_hr = LEAN && active && j == nk - 1;
    // poll entry e = lane + 64 m: gathered form = workgroup e; WP = wave (e & mask) of workgroup (e >> wsh).  64 entries are a
    // whole number of workgroups, so load m reads at the lane's offset of load 0 plus m uniform strides.  Entries beyond the
    // launch read this workgroup's own granule instead (the other parity's block follows this one: its lines are being written)
    // and are masked out of the epoch test and of the sum.
    struct LeanLd { int wi0, wv0; unsigned part, halo; bool want_l, want_r; };
    auto l_derive = [&](int ln) {
        LeanLd d;
        d.wi0 = ln >> l_wsh; d.wv0 = ln & ((1 << l_wsh) - 1);
        d.part = (unsigned)d.wi0 * l_slotB + (unsigned)d.wv0 * (unsigned)GPV * 8u;
        d.want_l = loc_left && ln < S; d.want_r = loc_right && ln >= 32 && ln < 32 + S;
        d.halo = d.want_l ? (unsigned)(wg - 1) * l_slotB + (16u + (unsigned)(S + ln) * (unsigned)GPV) * 8u
               : d.want_r ? (unsigned)(wg + 1) * l_slotB + (16u + (unsigned)(ln - 32) * (unsigned)GPV) * 8u
                          : (unsigned)wg * l_slotB;                        // wave 0's own partial granule: always current
        return d;
    };
    const LeanLd l_kept = l_derive(lane);
    const unsigned l_ld_step = (unsigned)(64 >> l_wsh) * l_slotB;
    const int l_pm = ((W << l_wsh) + 63) >> 6;
    typedef __attribute__((address_space(1))) char gchar;
    auto l_at = [](gu64 *base, unsigned boff) { return (gu64 *)((gchar *)base + boff); };
    // one sweep of N loads per lane until every granule watched carries the epoch; false = gave up (time-out / another
    // workgroup reported one)
    // (returns the lane's sum of the partials it read - in load order, as allreduce_and_halo - and its halo entry)
    // Sleep before the first sweep (launches across XCDs; gathered form), in units of ~74 cycles (s_sleep 1 + the loop).  A
    // sweep that comes before the last publisher's store has crossed the fabric is wasted and the next one costs a whole
    // round trip more; a sweep that comes late wastes the difference.  Swept with this hand-off (tools/sleep_sweep.py on a scratch
    // build; us per iteration): 14/7/4096 f32 (W = 114) 12: 3.64, 14: 3.56, 16: 3.40, 18: 3.46; 14/7/2048 f32 (57) 12: 3.25,
    // 14: 3.13, 16: 3.22; 14/7/4096 f64 (128) 12: 4.40, 14: 4.18, 16: 4.24; 32/16/1024 f32 (64) 10: 3.71, 12: 3.60, 14: 3.69;
    // 32/16/2048 f32 (128) 12: 4.20, 14: 4.01, 16: 4.09 - the optimum grows with the number of workgroups (their skew).  A
    // controller in the polling wave (failed first sweep -> longer, a run of successes -> shorter) was tried and lost: one
    // workgroup's probe that fails delays everybody's next hand-off, so 114 independent probes keep the whole launch inflated
    // (3.58 against 3.40 with the fixed value).
    const int l_sleep = W > 32 ? 10 + W / 22 : 0;
    auto l_poll = [&](auto nc, const LeanLd &d, gu64 *pb, T &acc_out, T &hv_out) -> bool {
        constexpr int N = decltype(nc)::value;
        const unsigned l_ld_part = d.part, l_ld_halo = d.halo;
        unsigned long long raw[N][GPV], hraw[GPV];
        bool valid[N];
#pragma unroll
        for (int m = 0; m < N; ++m) valid[m] = d.wi0 + m * (64 >> l_wsh) < W && d.wv0 < nwaves;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (unsigned spin = 0;; ++spin) {
#pragma unroll
            for (int m = 0; m < N; ++m) {
                gu64 *pm = l_at(pb, valid[m] ? l_ld_part + (unsigned)m * l_ld_step : (unsigned)wg * l_slotB);
#pragma unroll
                for (int g = 0; g < GPV; ++g) raw[m][g] = __hip_atomic_load(pm + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            gu64 *ph = l_at(pb, l_ld_halo);
#pragma unroll
            for (int g = 0; g < GPV; ++g) hraw[g] = __hip_atomic_load(ph + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool ok = true;
#pragma unroll
            for (int m = 0; m < N; ++m) {
                bool okm = true;
#pragma unroll
                for (int g = 0; g < GPV; ++g) okm &= (unsigned)(raw[m][g] >> 32) == epoch;
                ok &= okm | !valid[m];
            }
#pragma unroll
            for (int g = 0; g < GPV; ++g) ok &= (unsigned)(hraw[g] >> 32) == epoch;
            bool stop = __all(ok), good = stop;
            if (!stop && (spin & 255u) == 255u) {
                const bool late = __builtin_amdgcn_s_memrealtime() - t0 > t_limit;
                const bool other = __hip_atomic_load(g_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.launch_id;
                stop = late || other;
            }
            if (stop) {
                T acc = (T)0;
#pragma unroll
                for (int m = 0; m < N; ++m) acc += valid[m] ? Gr::decode(raw[m]) : (T)0;
                acc_out = acc;
                hv_out = Gr::decode(hraw);
                return good;
            }
        }
    };
    auto handoff_lean = [&](auto fastc, T val, T prod, T &total) {
        constexpr bool FAST = decltype(fastc)::value;
        typedef typename std::conditional<FAST, LGr, Gr>::type SG;
        ++epoch;
        if constexpr (ABL) {
            if (abl & 4) { total = (T)1 + prod * (T)1e-30; return; }
        }
        if (aborted) { total = (T)0; return; }
        gu64 *pb = slots + (size_t)(epoch & 1) * W * slotG;                       // this parity's block (uniform)
        T *wp = wpart[epoch & 1];
        if constexpr (WP) {
            const T ws = wave_sum(prod);
            if (lane == 0) SG::store(l_at(pb, l_st_part), epoch, ws);
        } else partials_store(wp, wave, lane, prod);
        if (l_hl) SG::store(l_at(pb, l_st_halo), epoch, val);
        if (l_hr) SG::store(l_at(pb, l_st_halo + (unsigned)(S * GPV * 8)), epoch, val);
        if constexpr (!WP) __syncthreads();                                        // B1
        if (wave == 0) {
            if constexpr (!WP) {
                const T mine_tot = partials_total<T, (MAXT <= 512 ? 8 : 16)>(wp, nwaves, lane);
                if (lane == 0) SG::store(l_at(pb, l_st_part), epoch, mine_tot);
            }
            if constexpr (!WP) {
                for (int i = 0; i < l_sleep; ++i) __builtin_amdgcn_s_sleep(1);
            }
            LeanLd d = l_kept;
            if constexpr (!KEEP) {                   // re-derived behind an empty asm the compiler cannot hoist out of the loop
                int ln = lane;
                asm volatile("" : "+v"(ln));
                d = l_derive(ln);
            }
            bool done;
            T acc, hv;
            if (l_pm == 1) done = l_poll(std::integral_constant<int, 1>{}, d, pb, acc, hv);
            else if (l_pm == 2) done = l_poll(std::integral_constant<int, 2>{}, d, pb, acc, hv);
            else if (l_pm == 3) done = l_poll(std::integral_constant<int, 3>{}, d, pb, acc, hv);
            else done = l_poll(std::integral_constant<int, 4>{}, d, pb, acc, hv);
            const T tot = wave_sum(acc);
            hv_reg = (d.want_l || d.want_r) ? hv : (T)0;
            if (lane == 0) {
                if (!done) {
                    __hip_atomic_store(g_status, a.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_abort = 1;
                }
                bc[epoch & 1] = tot;
            }
        }
        __syncthreads();                                                           // B2
        total = bc[epoch & 1];
        aborted = s_abort != 0;
    };
    const bool flat = MR && a.flat != 0;
    auto exchange = [&](auto fastc, T val, T prod, T &total) {
        if constexpr (LEAN) handoff_lean(fastc, val, prod, total);
        else if (flat) allreduce_flat(val, prod, total);
        else allreduce_and_halo(val, prod, total);
    };
    // outside the iteration loop: the store scope as a run-time choice
    auto exchange_rt = [&](T val, T prod, T &total) {
        if constexpr (LEAN && WP) {
            if (fast_st) exchange(std::true_type{}, val, prod, total);
            else exchange(std::false_type{}, val, prod, total);
        } else exchange(std::false_type{}, val, prod, total);
    };

    // ---- one-XCD launches: are we really on one XCD?  One extra all-to-all round (agent scope) with the XCC id as payload;
    // every workgroup reads the same W ids, so all take the same decision.  ~0.7 us once per launch.
    if ((WP || !LEAN) && !MR && X > 0 && W > 1 && W <= 64 && !(abl & 4)) {      // (the lean gathered form keeps agent scope: no round)
        __shared__ int s_same;
        ++epoch;
        if (wave == 0) {
            unsigned id;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
            id &= 0xfu;
            gu64 *pb = slots + (size_t)(epoch & 1) * W * slotG;
            if (lane == 0) __hip_atomic_store(pb + (size_t)wg * slotG, ((unsigned long long)epoch << 32) | id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            gu64 *pp = pb + (size_t)min(lane, W - 1) * slotG;
            unsigned long long raw = 0;
            bool fail = false;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (unsigned spin = 0;; ++spin) {
                raw = __hip_atomic_load(pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all((unsigned)(raw >> 32) == epoch)) break;
                if ((spin & 255u) == 255u) {
                    const bool late = __builtin_amdgcn_s_memrealtime() - t0 > t_limit;
                    const bool other = __hip_atomic_load(g_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.launch_id;
                    if (late || other) { fail = true; break; }
                }
            }
            const bool same = !fail && __all((unsigned)raw == id);
            if (lane == 0) {
                s_same = same ? 1 : 0;
                if (fail) {
                    __hip_atomic_store(g_status, a.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    s_abort = 1;
                }
            }
        }
        __syncthreads();
        fast_st = s_same != 0;
        aborted = s_abort != 0;
    }
    // ---- r~ = Pinv r ; p = r~ ; eta = r . r~   (gato_pcg.cuh:316-335) ------------------------
    // DR: the operand window of the lane's knot comes from registers - its own block sits in the lanes of its DPP row(s) (`own` =
    // this lane's entry), the neighbouring knots' entries of the lane's row index are read from the LDS window (two scalars;
    // written before the last barrier).  Lanes without a row clamp their index: in bounds, finite, times a zero matrix row.
    auto dpp_times = [&](const auto &mat, int w, T own) -> T {
        if constexpr (DR) {
            const T *xw = &xs[w][j * SP];
            T y;
            if constexpr (S <= 16) {
                const int rc = r_ < S ? r_ : S - 1;
                const T x3[3] = {xw[rc], own, xw[2 * SP + rc]};
                y = row_times_dpp<T, S>(mat, x3);
            } else {
                const int c = r_ & 15;
                const T x6[6] = {xw[c], xw[16 + c], xw[SP + c], xw[SP + 16 + c], xw[2 * SP + c], xw[2 * SP + 16 + c]};
                y = row_times_dpp<T, S>(mat, x6);
            }
            return active ? y : (T)0;
        } else return (T)0;
    };
    auto pinv_times = [&](const T *xw, T own) -> T {
        if constexpr (NR) return (T)0;
        else if constexpr (DR) return dpp_times(pm, 1, own);
        else if constexpr (NL > 0) return row_times_window_lds<T, S, SP, NL, MAXT>(pm, ptail, tid, xw);
        else return row_times_window<T, S, SP>(pm, xw);
    };
    // ---- optional true warm start (the reference accepts input_lambda but restarts from
    // zero, gato_pcg.cuh:303):  lambda = lambda0,  r = gamma - S lambda0.  The ghost blocks of r then come from the
    // neighbours through the ordinary hand-off.
    if (a.lambda0) {
        const T *__restrict__ dL0 = static_cast<const T *>(a.lambda0) + sys * S * K;
        lam = active ? dL0[(size_t)k * S + r_] : (T)0;
        if (active) xs[0][(j + 1) * SP + r_] = lam;
#pragma unroll 1
        for (int e = 0; e < ne; ++e) {
            const int q = tid + e * (int)blockDim.x;
            if (q < n_ext_rows) {
                const T l0 = dL0[(size_t)(k0 + xk) * S + q];
                xst[0][q] = l0;
                xs[0][(xk + q / S + 1) * SP + q % S] = l0;
            }
        }
        if (tid < S) {
            if (has_left) xs[0][tid] = dL0[(size_t)(k0 - 1) * S + tid];
        } else if (tid < 2 * S) {
            if (has_right) xs[0][(nk + 1) * SP + (tid - S)] = dL0[(size_t)(k0 + nk) * S + (tid - S)];
        }
        __syncthreads();
        if constexpr (DR) r -= dpp_times(sm, 0, lam);
        else if constexpr (!NR) r -= row_times_window<T, S, SP>(sm, &xs[0][j * SP]);
        if constexpr (XR > 0) (void)extra_rows(dS, 0);                       // product array <- S lambda0 on the extra rows
        __syncthreads();
        if (active) xs[1][(j + 1) * SP + r_] = r;
#pragma unroll 1
        for (int e = 0; e < ne; ++e) {
            const int q = tid + e * (int)blockDim.x;
            if (q < n_ext_rows) xs[1][(xk + q / S + 1) * SP + q % S] -= xst[1][q];
        }
        if constexpr (NR) {                 // the hand-off publishes from the product array: put r's boundary blocks there
            __syncthreads();
            if (tid < S) xst[1][tid] = xs[1][SP + tid];
            else if (tid < 2 * S) xst[1][(nk - 1) * S + (tid - S)] = xs[1][nk * SP + (tid - S)];
        }
        if (multi) {
            T dummy;
            exchange_rt(r, (T)0, dummy);
            if constexpr (RG) {
                g_r = hv_reg;
                if (g_lane) xs[1][gslot] = g_r;
            } else {
                if (tid < S) xs[1][tid] = gh[0][tid];
                else if (tid < 2 * S) xs[1][(nk + 1) * SP + (tid - S)] = gh[1][tid - S];
            }
        }
        // the p window is rebuilt from r~ below; clear what lambda0 left in its ghost slots
        if (tid < S) xs[0][tid] = (T)0;
        else if (tid < 2 * S) xs[0][(nk + 1) * SP + (tid - S)] = (T)0;
        __syncthreads();
    }
    rt = pinv_times(&xs[1][j * SP], r);
    {
        T prod0 = r * rt;
        if constexpr (XR > 0) prod0 += extra_rows(dP, 1);
        exchange_rt(rt, prod0, eta);
    }
    const bool rec_on = a.eta_hist != nullptr;                 // wave-uniform: one scalar branch when recording is off
    const bool rec = wg == 0 && tid == 0 && sys == 0;
    if (rec_on && rec) a.eta_hist[0] = (double)eta;
    if (!aborted) {
        p = rt;
        if (active) xs[0][(j + 1) * SP + r_] = p;
#pragma unroll 1
        for (int e = 0; e < ne; ++e) {
            const int q = tid + e * (int)blockDim.x;
            if (q < n_ext_rows) xs[0][(xk + q / S + 1) * SP + q % S] = xst[1][q];
        }
        if (multi) {
            if constexpr (RG) {
                g_p = hv_reg;
                if (g_lane) xs[0][gslot] = g_p;
            } else {
                if (tid < S) xs[0][tid] = gh[0][tid];
                else if (tid < 2 * S) xs[0][(nk + 1) * SP + (tid - S)] = gh[1][tid - S];
            }
        }
        __syncthreads();

        auto iterate = [&](auto fastc) {
        for (int it = 0; it < a.max_iters; ++it) {                              // gato_pcg.cuh:348
            // upsilon = S p ; v = p . upsilon                                     (:349-357)
            GATO_STAMP(5)
            if constexpr (NR) ups = (T)0;
            else if constexpr (DR) ups = (abl & 1) ? p * sm[0] : dpp_times(sm, 0, p);
            else ups = (abl & 1) ? p * sm[0] : row_times_window<T, S, SP>(sm, &xs[0][j * SP]);
            GATO_STAMP(0)
            T v;
            {
                T prod = p * ups;
                if constexpr (XR > 0) prod += extra_rows(dS, 0);
                exchange(fastc, ups, prod, v);
            }
            GATO_STAMP(1)
            if (aborted) break;
            const T alpha = quotient(eta, v);                                    // :364
            lam += alpha * p;                                                   // :373-377
            r -= alpha * ups;
            if (active) xs[1][(j + 1) * SP + r_] = r;
#pragma unroll 1
            for (int e = 0; e < ne; ++e) {
                const int q = tid + e * (int)blockDim.x;
                if (q < n_ext_rows) {
                    const int wi = (xk + q / S + 1) * SP + q % S;
                    xst[0][q] += alpha * xs[0][wi];
                    xs[1][wi] -= alpha * xst[1][q];
                }
            }
            if (multi) {   // ghost r advances with the neighbours' upsilon blocks
                if constexpr (RG) {
                    g_r -= alpha * hv_reg;
                    if (g_lane) xs[1][gslot] = g_r;
                } else {
                    if (tid < S) xs[1][tid] -= alpha * gh[0][tid];
                    else if (tid < 2 * S) xs[1][(nk + 1) * SP + (tid - S)] -= alpha * gh[1][tid - S];
                }
            }
            if (!(abl & 8)) __syncthreads();                                    // B3
            GATO_STAMP(2)
            // r~ = Pinv r ; eta' = r . r~                                        (:380-394)
            rt = (abl & 2) ? r * pm[0] : pinv_times(&xs[1][j * SP], r);
            GATO_STAMP(3)
            {
                T prod = r * rt;
                if constexpr (XR > 0) prod += extra_rows(dP, 1);
                exchange(fastc, rt, prod, eta_new);
            }
            GATO_STAMP(4)
            if (aborted) break;
            if (rec_on) {
                if (rec) a.eta_hist[it + 1] = (double)eta_new;
            }
            if (fabs(eta_new) < tol) { iters = it; break; }                     // :404-411
            const T beta = quotient(eta_new, eta);                               // :415
            p = rt + beta * p;                                                  // :416-419
            if (active) xs[0][(j + 1) * SP + r_] = p;
#pragma unroll 1
            for (int e = 0; e < ne; ++e) {
                const int q = tid + e * (int)blockDim.x;
                if (q < n_ext_rows) {
                    const int wi = (xk + q / S + 1) * SP + q % S;
                    xs[0][wi] = xst[1][q] + beta * xs[0][wi];
                }
            }
            if (multi) {
                if constexpr (RG) {
                    g_p = hv_reg + beta * g_p;
                    if (g_lane) xs[0][gslot] = g_p;
                } else {
                    if (tid < S) xs[0][tid] = gh[0][tid] + beta * xs[0][tid];
                    else if (tid < 2 * S) xs[0][(nk + 1) * SP + (tid - S)] = gh[1][tid - S] + beta * xs[0][(nk + 1) * SP + (tid - S)];
                }
            }
            eta = eta_new;                                                      // :420
            if (!(abl & 8)) __syncthreads();                                    // B6
        }
        };
        if constexpr (LEAN && WP) {                 // the loop twice: workgroup-scope stores (verified one-XCD placement) / agent scope
            if (fast_st) iterate(std::true_type{});
            else iterate(std::false_type{});
        } else iterate(std::false_type{});
    }
    if (active) dL[(size_t)k * S + r_] = lam;                                   // :433-435
#pragma unroll 1
    for (int e = 0; e < ne; ++e) {
        const int q = tid + e * (int)blockDim.x;
        if (q < n_ext_rows) dL[(size_t)(k0 + xk) * S + q] = xst[0][q];
    }
    if constexpr (MR) (void)cluster_lambda_ghost<T, S>(a, wg, W, dL, aborted);      // lambda_{k_end} for this rank's dz launch
    // ---- dz back-substitution in the same launch (one-workgroup launches: every lambda_k is here).  Same formulas and
    // accumulation order as dz_kernel (gato_assembly.hip; gato_schur.cuh:758-867, D2 fixed): bit-identical results.
    if constexpr (XR == 0 && !MR) {
        if (a.dz != nullptr && W == 1) {
            const int Cn = a.C, n = S + Cn;
            const size_t gs = (size_t)(S * S + Cn * Cn), cs = (size_t)(S * S + S * Cn), Nn = (size_t)n * K - Cn;
            const T *__restrict__ Gi = static_cast<const T *>(a.dz_Ginv) + sys * (gs * K - (size_t)Cn * Cn);
            const T *__restrict__ Cdn = static_cast<const T *>(a.dz_Cd) + sys * (cs * (K - 1));
            const T *__restrict__ gv = static_cast<const T *>(a.dz_g) + sys * Nn;
            T *__restrict__ dzo = static_cast<T *>(a.dz) + sys * Nn;
            const bool last = k == K - 1;
            __syncthreads();                                                // every wave has left the loop: the windows are free
            if (active) xs[0][(j + 1) * SP + r_] = lam;                     // lambda window
            __syncthreads();
            T tx = (T)0, tu = (T)0;
            if (active) {
                if (!last) {
                    const T *__restrict__ A = Cdn + (size_t)k * cs;
                    const T *lp = &xs[0][(j + 2) * SP];                    // lambda_{k+1}
                    T res = (T)0;
#pragma unroll
                    for (int t = 0; t < S; ++t) res = gato::fmaT(A[r_ * S + t], lp[t], res);      // A_k^T lambda_{k+1}   :833-838
                    tx = gv[(size_t)k * n + r_] - (lam + res);                                    // :841-852
                    if (r_ < Cn) {
                        const T *__restrict__ B = A + S * S;
                        T rb = (T)0;
#pragma unroll
                        for (int t = 0; t < S; ++t) rb = gato::fmaT(B[r_ * S + t], lp[t], rb);    // B_k^T lambda_{k+1}   :784-789
                        tu = gv[(size_t)k * n + S + r_] - rb;                                     // :792-796
                    }
                } else tx = gv[(size_t)k * n + r_] - lam;                                         // last state row (D2)
                xs[1][(j + 1) * SP + r_] = tx;
            }
            __syncthreads();                                                // lambda_{k+1} has been read everywhere
            if (active && !last && r_ < Cn) xs[0][(j + 1) * SP + r_] = tu;
            __syncthreads();
            if (active) {
                const T *__restrict__ Qi = Gi + (size_t)k * gs;
                const T *tv = &xs[1][(j + 1) * SP];
                T res = (T)0;
#pragma unroll
                for (int cc = 0; cc < S; ++cc) res = gato::fmaT(Qi[r_ + cc * S], tv[cc], res);   // Q_k^-1 (...)         :856-865
                dzo[(size_t)k * n + r_] = res;
                if (!last && r_ < Cn) {
                    const T *__restrict__ Ri = Qi + S * S;
                    const T *uv = &xs[0][(j + 1) * SP];
                    T ru = (T)0;
                    for (int cc = 0; cc < Cn; ++cc) ru = gato::fmaT(Ri[r_ + cc * Cn], uv[cc], ru);   // R_k^-1 (...)         :799-808
                    dzo[(size_t)k * n + S + r_] = ru;
                }
            }
        }
    }
    if (wg == 0 && tid == 0) {
        a.iters[sys] = aborted ? -1 : iters;      // in-band: a timed-out hand-off is visible without a second call
        if (a.final_eta && sys == 0) *a.final_eta = (double)eta_new;
        if (STAMP && a.stamps) {
            for (int i = 0; i < 8; ++i) a.stamps[i] = seg[i];
            a.stamps[8] = __builtin_amdgcn_s_memtime() - t_begin;
            a.stamps[9] = __builtin_amdgcn_s_memrealtime() - rt_begin;
        }
    }
}

// Helper blocks of the ONE-SYSTEM launches of the one-workgroup kernels (gridDim = 1 + 8 x helpers; block 0 solves).  Called by
// every block but block 0; `scratch`: LDS, (4 S^2 + 6 S) values per wave of the block.
template <typename T, int S>
__device__ __forceinline__ void one_system_helper(const PcgLaunch &a, T *scratch)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nt = blockDim.x, nwv = nt >> 6, K = a.K;
    if ((blockIdx.x & 7) != 0) return;
    const size_t h = (blockIdx.x >> 3) - 1, nh = (gridDim.x - 1) >> 3;
    const size_t bytes = a.imgS ? (size_t)3 * S * a.img_ld * sizeof(T) : (size_t)3 * S * S * K * sizeof(T);   // of each array the solver loads from
    const size_t lines = (bytes + 127) / 128;
    typedef int I4 __attribute__((ext_vector_type(4)));
    const I4 *s4 = static_cast<const I4 *>(a.imgS ? a.imgS : a.S_bd), *p4 = static_cast<const I4 *>(a.imgS ? a.imgP : a.P_bd);
    const size_t last16 = bytes / 16 - 1;                                     // last whole 16-byte unit of an array
    I4 acc = {0, 0, 0, 0};
    for (size_t l = h * nt + tid; l < lines; l += nh * nt) {
        const size_t u = l * 8 < last16 ? l * 8 : last16;
        acc ^= s4[u] ^ p4[u];
    }
    asm volatile("" ::"v"(acc));
    // ... and then stay for the dz back-substitution (compute_dz, gato_schur.cuh:758-867): wave w of helper h takes knot
    // h * WT + w, brings Q_k^-1, R_k^-1, A_k, B_k and g_k into LDS while the solve runs, waits (sleeping) until the solving
    // workgroup has stored launch_id in *dz_flag - lambda is complete and released - and finishes in about a microsecond.
    // That replaces a launch of its own (5.3 us + the gap in front of it) at the end of every step.  Formulas and
    // accumulation order of dz_kernel (gato_assembly.hip), row by row: the same bits.  The solving workgroup never waits for
    // a helper, so helpers that are scheduled late (or after it has finished) just find the flag set.
    if (a.dz_helpers && a.dz != nullptr) {
        const int kq = (int)h * nwv + wave;
        if (kq >= K) return;
        const int Cn = a.C, n = S + Cn, SS = S * S;
        const size_t gs = (size_t)(SS + Cn * Cn), cs = (size_t)(SS + S * Cn);
        const T *__restrict__ Qg = static_cast<const T *>(a.dz_Ginv) + (size_t)kq * gs;
        const T *__restrict__ Ag = static_cast<const T *>(a.dz_Cd) + (size_t)kq * cs;
        const T *__restrict__ gg = static_cast<const T *>(a.dz_g) + (size_t)kq * n;
        const T *__restrict__ lg = static_cast<const T *>(a.lambda) + (size_t)kq * S;
        T *__restrict__ dzo = static_cast<T *>(a.dz) + (size_t)kq * n;
        const bool last = kq == K - 1;
        T *scr = scratch + (size_t)wave * (4 * SS + 6 * S);                                   // this wave's own part of the LDS
        T *sQi = scr, *sA = sQi + SS, *sRi = sA + SS, *sB = sRi + SS, *sl = sB + SS, *st = sl + 2 * S, *sg = st + 2 * S;
        for (int i = lane; i < SS; i += 64) sQi[i] = Qg[i];
        for (int i = lane; i < (last ? S : n); i += 64) sg[i] = gg[i];
        if (!last) {
            for (int i = lane; i < Cn * Cn; i += 64) sRi[i] = Qg[SS + i];
            for (int i = lane; i < SS; i += 64) sA[i] = Ag[i];
            for (int i = lane; i < S * Cn; i += 64) sB[i] = Ag[SS + i];
        }
        gi32 *flag = (gi32 *)a.dz_flag;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        // the solve itself may legitimately take longer than the hand-off time-out (max_iters is the caller's): the helpers allow
        // it 50 us per iteration on top (25 x what an iteration takes) before they call the solving workgroup dead
        const unsigned long long patience = a.timeout_ticks + (unsigned long long)(a.max_iters > 0 ? a.max_iters : 0) * 5000ull;
        bool late = false;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != a.launch_id) {
            __builtin_amdgcn_s_sleep(4);                      // ~0.1 us between looks: 50 waves, one load each
            if (__builtin_amdgcn_s_memrealtime() - t0 > patience) { late = true; break; }
        }
        if (late) {         // cannot happen unless the solving workgroup died: report it like a hand-off time-out (the solving
            // workgroup, should it still arrive, reads the status word and marks the solve incomplete: iters = -1)
            if (lane == 0) __hip_atomic_store((gi32 *)a.status, a.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        for (int i = lane; i < (last ? S : 2 * S); i += 64) sl[i] = __hip_atomic_load(lg + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wave_lds_fence();
        if (!last) {
            for (int i = lane; i < S; i += 64) {                                  // A_k^T lambda_{k+1}   :833-838
                T res = (T)0;
#pragma unroll
                for (int t = 0; t < S; ++t) res = gato::fmaT(sA[i * S + t], sl[S + t], res);
                st[i] = res;
            }
            for (int i = lane; i < Cn; i += 64) {                                 // B_k^T lambda_{k+1}   :784-789
                T res = (T)0;
#pragma unroll
                for (int t = 0; t < S; ++t) res = gato::fmaT(sB[i * S + t], sl[S + t], res);
                st[S + i] = res;
            }
            wave_lds_fence();
            for (int i = lane; i < S; i += 64) st[i] = sg[i] - (sl[i] + st[i]);                    // :841-852
            for (int i = lane; i < Cn; i += 64) st[S + i] = sg[S + i] - st[S + i];                 // :792-796
        } else {
            for (int i = lane; i < S; i += 64) st[i] = sg[i] - sl[i];             // last state row (D2)
        }
        wave_lds_fence();
        for (int r = lane; r < S; r += 64) {                                      // Q_k^-1 (...)         :856-865
            T res = (T)0;
#pragma unroll
            for (int cc = 0; cc < S; ++cc) res = gato::fmaT(sQi[r + cc * S], st[cc], res);
            dzo[r] = res;
        }
        if (!last) {
            for (int r = lane; r < Cn; r += 64) {                                 // R_k^-1 (...)         :799-808
                T res = (T)0;
                for (int cc = 0; cc < Cn; ++cc) res = gato::fmaT(sRi[r + cc * Cn], st[S + cc], res);
                dzo[S + r] = res;
            }
        }
    }
}

// Independent accumulation chains per row of the fp32 two-rows-per-lane kernel's products.  One chain of 3S dependent packed
// FMAs (the reference's left-to-right order, GATO_PAIR_CHAINS = 1) is latency: a SIMD that hosts one wave of the launch idles
// between them and the compiler pads every one with a wait state.  2 = even columns + odd columns (the order of
// row_times_window's packed form, which every other fp32 kernel of the family runs): 14/7/50 f32 1.365 -> 1.256 us per
// iteration (same box; 3 chains 1.27, 4 chains 1.29; S even: a column's parity is its parity inside the 16-byte read).  fp32 parity is measured against the fp64 oracle beside the reference
// order's own error (tests/f32_parity.py); fp64 keeps the reference's order everywhere.
#ifndef GATO_PAIR_CHAINS
#define GATO_PAIR_CHAINS 2
#endif
#if GATO_PAIR_CHAINS != 1 && GATO_PAIR_CHAINS != 2
#error "GATO_PAIR_CHAINS: 1 (the reference's left-to-right order) or 2 (even + odd columns)"
#endif
// ---- fp32, one workgroup (or one workgroup per system of a batch), TWO rows per lane --------------------------
// The single-workgroup loop is instruction-issue bound: with two rows of the same knot per lane
// the operand-window reads are shared by both rows, the FMAs pair up as v_pk_fma_f32 and the wave count halves
// (IIWA 14/7/50: 6 waves instead of 11), and the one-CU regime extends to K*S <= 2*MAXT rows.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// WAVE-PRIVATE operand windows (round 3; the shared-window form it replaced - option shared_windows - was removed in round 5).  In a shared-window form an iteration has four
// barriers: two inside the block sums and two between a vector update and the product that reads the updated window (every
// lane must have stored its entries of r / p before any lane reads its neighbours').  Here every wave keeps its OWN copy of
// the part of the r and p windows its lanes read - its own 128 rows plus a halo of up to 2S - 1 rows on either side - and
// advances the halo rows itself: up to 4S - 2 lanes of the wave hold one halo row of r and p in a register, fetch that row's
// entry of upsilon (of r~) from a shared exchange window which the owners fill BEFORE the block sum's barrier, and apply the
// owner's own FMA (r - alpha upsilon, r~ + beta p: the same bits, tools/pw_check.py and test_private_windows_*).  A wave then
// reads only LDS words it wrote itself (LDS operations of one wave execute in order: no barrier, no wait), so the two window
// barriers and the LDS write latency in front of them are gone: two barriers per iteration.  14/7/50 fp32: 1.405 -> 1.340 us
// per iteration (with 14 spilled VGPRs).  The fp64 mixed-rows kernel gains nothing from it (its loop is bound by the LDS read
// queue, not by barriers: measured 1.945 us with two barriers and no halo update at all against 1.94) and keeps shared windows.
template <int S, int MAXT>
__global__ __launch_bounds__(MAXT) void pcg_single_f32x2_kernel(PcgLaunch a)
{
    constexpr int H = S / 2;                       // lanes per knot
    constexpr int SP = pad_to(S, 4);
    constexpr int MAXK = (MAXT + H - 1) / H;
    static_assert(S % 2 == 0, "two rows per lane need an even STATE_SIZE");
    __shared__ __attribute__((aligned(16))) float xs[2][(MAXK + 2) * SP];
    // (partials_total_all8 reads the four row sums of EIGHT waves whatever the launch has: slots of waves that do not exist - MAXT =
    //  256 for the generic shapes - exist here and hold zeros)
    constexpr int WPW = (MAXT + 63) / 64 < 8 ? 8 : (MAXT + 63) / 64;
    __shared__ __attribute__((aligned(32))) float wpart[2][4 * WPW];
    static_assert(MAXT > 512 || sizeof(wpart) / 2 >= 32 * sizeof(float), "partials_total_all8 reads 32 values per parity");
    constexpr int PK = (128 - 1 + S - 1) / S + 1 + 2;                 // knots a wave's 128 rows can span + a halo knot on either side
    constexpr int PWLEN = (MAXT / 64) * PK * SP;
    static_assert(4 * S - 2 <= 64, "private windows: one halo row per lane");
    __shared__ __attribute__((aligned(16))) float pwin[2][PWLEN];     // [0] = p, [1] = r, wave after wave

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
    const int K = a.K;
    // one system: helper blocks (L2 warm-up of S and Pinv, then dz) as in pcg_single_f64m_kernel
    __shared__ __attribute__((aligned(16))) float hscr[(MAXT / 64) * (4 * S * S + 6 * S)];
    if (a.batch <= 1 && blockIdx.x > 0) {
        one_system_helper<float, S>(a, hscr);
        return;
    }
    const size_t sys = a.batch > 1 ? blockIdx.x : 0;
    const int j = tid / H, h = tid - j * H;        // knot, row pair (2h, 2h + 1): adjacent, so every matrix column is ONE 8-byte load
    const bool active = j < K;                     // (pairs (h, h + H) cost two scattered 4-byte loads per column: 10 us per launch)
    const int r0 = 2 * h, r1 = 2 * h + 1;

    const float *__restrict__ dS = static_cast<const float *>(a.S_bd) + sys * 3 * S * S * K;
    const float *__restrict__ dP = static_cast<const float *>(a.P_bd) + sys * 3 * S * S * K;
    const float *__restrict__ dG = static_cast<const float *>(a.gamma) + sys * S * K;
    float *__restrict__ dL = static_cast<float *>(a.lambda) + sys * S * K;

    f32x2 sm[3 * S], pm[3 * S];
    if (a.imgS != nullptr) {          // transposed copies written by the assembly launch of this solve (see pcg_single_f64m_kernel)
        const float *__restrict__ iS = static_cast<const float *>(a.imgS) + 2 * tid, *__restrict__ iP = static_cast<const float *>(a.imgP) + 2 * tid;
        const size_t ld = (size_t)a.img_ld;            // rows (2 tid, 2 tid + 1) = (j S + r0, j S + r1): the lane's pair
#pragma unroll
        for (int c = 0; c < 3 * S; ++c) sm[c] = *reinterpret_cast<const f32x2 *>(iS + c * ld);
#pragma unroll
        for (int c = 0; c < 3 * S; ++c) pm[c] = *reinterpret_cast<const f32x2 *>(iP + c * ld);
    } else {
        const size_t base = (size_t)(active ? j : 0) * 3 * S * S;
#pragma unroll
        for (int c = 0; c < 3 * S; ++c) sm[c] = *reinterpret_cast<const f32x2 *>(dS + base + c * S + r0);   // even index: 8-byte aligned
#pragma unroll
        for (int c = 0; c < 3 * S; ++c) pm[c] = *reinterpret_cast<const f32x2 *>(dP + base + c * S + r0);
#pragma unroll
        for (int c = 0; c < 3 * S; ++c) {        // (the selects after ALL loads: see pcg_single_f64m_kernel; 14/7/50 f32 1.423 -> 1.408 us per iteration)
            const bool ok = active && !(j == 0 && c < S) && !(j == K - 1 && c >= 2 * S);   // gato_utils.cuh:157-174
            sm[c] = ok ? sm[c] : f32x2{0.f, 0.f};
            pm[c] = ok ? pm[c] : f32x2{0.f, 0.f};
        }
    }
    for (int i = tid; i < 2 * (MAXK + 2) * SP; i += blockDim.x) (&xs[0][0])[i] = 0.f;
    for (int i = tid; i < 2 * 4 * WPW; i += blockDim.x) (&wpart[0][0])[i] = 0.f;       // partials_total_all8 reads all eight waves' slots
    for (int i = tid; i < 2 * PWLEN; i += blockDim.x) (&pwin[0][0])[i] = 0.f;
    __syncthreads();
    // the wave's rows [R0, R1), its private windows (slot 0 = knot jf - 1) and the lane's halo row
    int hoff = 0, hpo = 0, own_po = 0, win_po = 0;
    bool hvalid = false;
    float *pw_p = nullptr, *pw_r = nullptr;
    {
        const int R0 = 128 * wave, R1 = min(R0 + 128, K * S);
        pw_p = &pwin[0][wave * PK * SP]; pw_r = &pwin[1][wave * PK * SP];
        if (R0 < R1) {
            const int jf = R0 / S, jl = (R1 - 1) / S, base_row = (jf - 1) * S;
            const int nb = R0 - base_row, na = (jl + 2) * S - R1;
            const int hrow = lane < nb ? base_row + lane : R1 + (lane - nb);
            hvalid = lane < nb + na && hrow >= 0 && hrow < K * S;
            const int hj = hvalid ? hrow / S : 0, hr = hvalid ? hrow - hj * S : 0;
            hoff = hvalid ? (hj + 1) * SP + hr : 0;
            hpo = hvalid ? (hj - jf + 1) * SP + hr : 0;
            own_po = active ? (j - jf + 1) * SP + r0 : 0;
            win_po = active ? (j - jf) * SP : 0;
        }
    }

    auto times_window = [&](const f32x2 (&m)[3 * S], const float *xw) -> f32x2 {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
#if GATO_PAIR_CHAINS == 2
        // even and odd columns in chains of their own.  (Written exactly like this: with the accumulators in an array indexed by
        // column % 2 the scheduler ran one whole chain after the other - no gain - and pinning the interleaved order with empty
        // asm statements cost as many wait states as it saved; this form compiles to alternating FMAs without any.)
        f32x2 acc = {0.f, 0.f}, acc1 = {0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 3; ++b) {
#pragma unroll
            for (int i = 0; i < SP / 4; ++i) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(xw + b * SP + i * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (i * 4 + e < S) {
                        if (e & 1) acc1 = __builtin_elementwise_fma(m[b * S + i * 4 + e], f32x2{v[e], v[e]}, acc1);
                        else acc = __builtin_elementwise_fma(m[b * S + i * 4 + e], f32x2{v[e], v[e]}, acc);
                    }
            }
        }
        return acc + acc1;
#else
        f32x2 acc = {0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 3; ++b) {
#pragma unroll
            for (int i = 0; i < SP / 4; ++i) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(xw + b * SP + i * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (i * 4 + e < S) acc = __builtin_elementwise_fma(m[b * S + i * 4 + e], f32x2{v[e], v[e]}, acc);
            }
        }
        return acc;
#endif
    };
    unsigned epoch = 0;
    // (lanes without rows hold zeros in every vector - their matrix rows are zero - and store them into the zero padding in
    //  front of knot 0 instead of sitting out behind an exec mask, as in pcg_single_f64m_kernel)
    const int put_off = active ? (j + 1) * SP + r0 : r0;
    auto put = [&](float *buf, f32x2 v) { *reinterpret_cast<f32x2 *>(buf + put_off) = v; };

    const float *wp_ = pw_p + win_po, *wr_ = pw_r + win_po;
    auto put_private = [&](float *pw, f32x2 v, float g) {
        if (active) *reinterpret_cast<f32x2 *>(pw + own_po) = v;
        if (hvalid) pw[hpo] = g;
        wave_lds_fence();
    };
    auto block_sum_x = [&](float prod, const float *xw, float &hx) -> float {
        ++epoch;
        float *wp = wpart[epoch & 1];
        partials_store(wp, wave, lane, prod);
        __syncthreads();
        hx = xw[hoff];
        if constexpr (MAXT <= 512) return partials_total_all8<float>(wp, lane);
        return partials_total<float, 16>(wp, nwaves, lane);
    };

    f32x2 lam = {0.f, 0.f};
    f32x2 r = active ? f32x2{dG[(size_t)j * S + r0], dG[(size_t)j * S + r1]} : f32x2{0.f, 0.f};
    if (a.lambda0) {                                                       // true warm start (opt-in)
        const float *__restrict__ dL0 = static_cast<const float *>(a.lambda0) + sys * S * K;
        if (active) lam = f32x2{dL0[(size_t)j * S + r0], dL0[(size_t)j * S + r1]};
        put(xs[0], lam);
        __syncthreads();
        r -= times_window(sm, &xs[0][j * SP]);
        __syncthreads();
    }
    float gr = 0.f, gp = 0.f;                                              // r and p of the lane's halo row
    put(xs[0], r);
    __syncthreads();
    gr = hvalid ? xs[0][hoff] : 0.f;
    put_private(pw_r, r, gr);
    f32x2 rt = times_window(pm, wr_);                                      // gato_pcg.cuh:316-335
    float eta, eta_new = 0.f;
    put(xs[1], rt);
    eta = block_sum_x(r[0] * rt[0] + r[1] * rt[1], xs[1], gp);
    const bool rec = a.eta_hist && tid == 0 && sys == 0;
    if (rec) a.eta_hist[0] = (double)eta;
    f32x2 p = rt, ups;
    put_private(pw_p, p, gp);
    int iters = a.max_iters;
    const float tol = (float)a.exit_tol;
    for (int it = 0; it < a.max_iters; ++it) {                             // gato_pcg.cuh:348
        ups = times_window(sm, wp_);
        float hx = 0.f;
        put(xs[0], ups);
        const float v = block_sum_x(p[0] * ups[0] + p[1] * ups[1], xs[0], hx);
        const float alpha = quotient(eta, v);
        lam += alpha * p;
        r -= alpha * ups;
        gr -= alpha * hx;
        asm volatile("" : "+v"(gr) : : "memory");          // the halo value first: no wait for hx BETWEEN the two LDS writes
        put_private(pw_r, r, gr);
        rt = times_window(pm, wr_);
        put(xs[1], rt);
        eta_new = block_sum_x(r[0] * rt[0] + r[1] * rt[1], xs[1], hx);
        if (rec) a.eta_hist[it + 1] = (double)eta_new;
        if (__builtin_amdgcn_readfirstlane((int)(fabsf(eta_new) < tol))) { iters = it; break; }                   // :404-411
        const float beta = quotient(eta_new, eta);
        p = rt + beta * p;
        eta = eta_new;
        gp = hx + beta * gp;
        asm volatile("" : "+v"(gp) : : "memory");
        put_private(pw_p, p, gp);
    }
    if (active) { dL[(size_t)j * S + r0] = lam[0]; dL[(size_t)j * S + r1] = lam[1]; }
    if (a.dz_helpers && a.dz != nullptr && a.batch <= 1) {       // lambda is complete: release it and tell the helper blocks
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) __hip_atomic_store((gi32 *)a.dz_flag, a.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- dz back-substitution in the same launch (batches: one workgroup per system), as in pcg_single_f64m_kernel: formulas and
    // accumulation order of dz_kernel (gato_assembly.hip; gato_schur.cuh:758-867, D2 fixed), row by row: bit-identical results.
    if (a.dz != nullptr && !a.dz_helpers) {
        const int Cn = a.C, n = S + Cn, k = j;
        const size_t gs = (size_t)(S * S + Cn * Cn), cs = (size_t)(S * S + S * Cn), Nn = (size_t)n * K - Cn;
        const float *__restrict__ Gi = static_cast<const float *>(a.dz_Ginv) + sys * (gs * K - (size_t)Cn * Cn);
        const float *__restrict__ Cdn = static_cast<const float *>(a.dz_Cd) + sys * (cs * (K - 1));
        const float *__restrict__ gv = static_cast<const float *>(a.dz_g) + sys * Nn;
        float *__restrict__ dzo = static_cast<float *>(a.dz) + sys * Nn;
        const bool last = k == K - 1;
        __syncthreads();                                                     // every wave has left the loop: both windows are free
        put(xs[0], lam);                                                     // lambda window
        __syncthreads();
        float tx[2] = {0.f, 0.f}, tu[2] = {0.f, 0.f};
        if (active) {
            for (int q = 0; q < 2; ++q) {
                const int rr = r0 + q;
                if (!last) {
                    const float *__restrict__ A = Cdn + (size_t)k * cs;
                    const float *lp = &xs[0][(j + 2) * SP];                  // lambda_{k+1}
                    float res = 0.f;
#pragma unroll
                    for (int t = 0; t < S; ++t) res = gato::fmaT(A[rr * S + t], lp[t], res);          // A_k^T lambda_{k+1}   :833-838
                    tx[q] = gv[(size_t)k * n + rr] - (lam[q] + res);                                  // :841-852
                    if (rr < Cn) {
                        const float *__restrict__ B = A + S * S;
                        float rb = 0.f;
#pragma unroll
                        for (int t = 0; t < S; ++t) rb = gato::fmaT(B[rr * S + t], lp[t], rb);        // B_k^T lambda_{k+1}   :784-789
                        tu[q] = gv[(size_t)k * n + S + rr] - rb;                                      // :792-796
                    }
                } else tx[q] = gv[(size_t)k * n + rr] - lam[q];                                       // last state row (D2)
                xs[1][(j + 1) * SP + rr] = tx[q];
            }
        }
        __syncthreads();                                                     // lambda_{k+1} has been read everywhere
        if (active && !last) {
            for (int q = 0; q < 2; ++q)
                if (r0 + q < Cn) xs[0][(j + 1) * SP + r0 + q] = tu[q];
        }
        __syncthreads();
        if (active) {
            const float *__restrict__ Qi = Gi + (size_t)k * gs;
            const float *tv = &xs[1][(j + 1) * SP];
            for (int q = 0; q < 2; ++q) {
                const int rr = r0 + q;
                float res = 0.f;
#pragma unroll
                for (int cc = 0; cc < S; ++cc) res = gato::fmaT(Qi[rr + cc * S], tv[cc], res);        // Q_k^-1 (...)         :856-865
                dzo[(size_t)k * n + rr] = res;
                if (!last && rr < Cn) {
                    const float *__restrict__ Ri = Qi + S * S;
                    const float *uv = &xs[0][(j + 1) * SP];
                    float ru = 0.f;
                    for (int cc = 0; cc < Cn; ++cc) ru = gato::fmaT(Ri[rr + cc * Cn], uv[cc], ru);    // R_k^-1 (...)         :799-808
                    dzo[(size_t)k * n + S + rr] = ru;
                }
            }
        }
    }
    if (tid == 0) {
        // a helper block that gave up on this launch (see one_system_helper) left dz rows unwritten: in-band, as a hand-off time-out
        const bool dz_lost = a.dz_helpers && a.dz != nullptr && a.batch <= 1 &&
                             __hip_atomic_load((gi32 *)a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.launch_id;
        a.iters[sys] = dz_lost ? -1 : iters;
        if (a.final_eta && sys == 0) *a.final_eta = (double)eta_new;
    }
}

// scheduling pattern for the straight-line block in front of it: DEPTH LDS reads, then (FA FMAs, RA reads) until the reads are
// out - FA : RA = the block's FMAs per read, so that the number of reads in flight stays at DEPTH - then the remaining FMAs
template <int DEPTH, int NREAD, int NFMA, int FA, int RA>
__device__ __forceinline__ void pin_reads_then_fmas()
{
    constexpr int D = DEPTH < NREAD ? DEPTH : NREAD, STEPS = (NREAD - D) / RA, TAILR = NREAD - D - STEPS * RA;
    __builtin_amdgcn_sched_group_barrier(0x100, D, 0);
#pragma unroll
    for (int i = 0; i < STEPS; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x002, FA, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, RA, 0);
    }
    if constexpr (TAILR > 0) {
        __builtin_amdgcn_sched_group_barrier(0x002, FA, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, TAILR, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x002, NFMA - FA * (STEPS + (TAILR > 0 ? 1 : 0)), 0);
}

// (Round 4 also had a HYBRID fp32 kernel here - four two-row waves + four DPP-row waves, one of each per SIMD, option f32_hybrid.
// Measured equal to the kernel above for two rounds (14/7/50: 1.234 against 1.250 us per iteration, 14/7/45 1.232 / 1.220): the
// fp32 iteration is bound by the serial chain around its two block sums, which every wave repeats, not by what a SIMD pulls
// through its LDS return path.  Removed in round 5.)
// threads of the two-rows-per-lane kernel: 2*3S*2 matrix registers + window + state must stay under the cap
// ---- fp64, one workgroup, MIXED rows per lane (IIWA 14/7/50 in fp64 = BASELINE configs[1]) -------------------------------
// The one-workgroup loop is bound by its LDS reads: every lane reads the 3S-entry operand window of its knot for each
// of the two products (16-byte broadcast reads), and 700 rows x 84 doubles do not fit the register file, so part of Pinv
// lives in LDS as well (pcg_resident_kernel<double, 14, 704, NL = 24>: 54 reads per lane and iteration on 11 waves = 594
// wave-reads).  Here 8 waves instead of 11: the lanes of the first W2 waves own TWO adjacent rows of a knot each - both
// rows share every window read - and the other waves own one knot per 16-lane DPP row with S and Pinv entirely in registers
// (round 2's layout had dense one-row waves there: option mixed_dense, removed in round 5 after two rounds of A/B at 1.94
// against 1.54 us per iteration).  Same recurrence, same per-row summation order as every other kernel of the family (a row's
// 3S products are added left to right); block sums as in partials_store.
// ABL: timing-only switches (bench.py's latency floor) as COMPILE-TIME constants - 3 no products, 4 no block sums, 15 loop
// skeleton; with run-time switches this loop compiles 40 % slower than the production kernel, which is no yardstick.
#ifndef GATO_L2_HELPERS
#define GATO_L2_HELPERS 8
#endif
// Round 4.  In-kernel s_memtime stamps of one iteration (tools/f64m_stamps.py, wave by wave) said: products 1020 + 1660 cycles of
// 4560, the two block sums 620 each, quotient + update + window write 260 / 330 - and that the waves sharing a SIMD with a
// two-row wave finish their products late while the two one-row waves that share the fourth SIMD are done after 690 / 770
// cycles = 42 reads x 16: a ds_read_b128 occupies its SIMD's LDS return path for 16 cycles (64 B per clock and SIMD; the 256 B
// per clock of the LDS array needs all four SIMDs reading).  What a SIMD reads per iteration is therefore the bound of the
// products: 126 reads on three SIMDs (two-row wave 21 + 63, one-row wave 21 + 21), 84 on the fourth.  Changes:
//  (1) DR: the one-row waves own 16-lane DPP rows (one knot per row, lanes S..15 idle, row_times_dpp: the operand window comes
//      from the neighbouring lanes' registers, 2 eight-byte LDS reads per product instead of 21 sixteen-byte ones) and FOUR waves
//      hold two rows per lane, one per SIMD beside one DPP wave: the same reads per SIMD and iteration on all four.  The lanes
//      the DPP rows leave idle are paid for by LDS: K2MAX knots x S / 2 two-row lanes keep the ODD columns of S and of Pinv there
//      (the even ones in registers - see m[] below: either product reads 3S/2 windows + 3S/2 pairs, one read per two packed-row
//      FMAs), less Pinv's last NPR odd columns, which stay in registers so that it fits (14/7: 34 knots = 238 lanes + 1 slot of
//      zeros, x 41 pairs x 16 B = 156,784 B + one window 5,824 + partial sums 512 = 163,120 of 163,840 B).  Same per-row
//      summation order as before (left to right): the same bits.
//  (2) the Pinv pairs in LDS through base registers + 16-bit immediate offsets (the compiler gave every column beyond the offset
//      range an address register of its own: 20 VGPRs);
//  (3) the order of LDS reads and FMAs of the two-row products is PINNED (sched_group_barrier: DEPTH reads first, then reads and
//      FMAs in the block's own ratio): the compiler's order keeps 3-5 reads in flight;
//  (4) wave-uniform branches around the two forms of a product (readfirstlane) that start with DIFFERENT empty asm statements:
//      the compiler had hoisted their common first window read and FMA in front of the branch with s_waitcnt lgkmcnt(0)
//      between them - one exposed LDS latency per product.
// Measured and not kept (same box, tools/ab_libs.py): block sums on the matrix core (two v_mfma_f64_16x16x4 with B = ones per
// wave sum + every lane adding the eight wave totals: 1.94 -> 2.04 us per iteration), the divisor-only half of beta = eta' / eta
// formed during the Pinv product (1.835 -> 1.833), Pinv pairs read before the barrier in front of the Pinv product.
#ifndef GATO_F64M_DEV_ABL
#define GATO_F64M_DEV_ABL 0   // A/B builds only (tools/devbuild.sh -DGATO_F64M_DEV_ABL=8: the loop without its two window barriers, timing only)
#endif
// VERDICT r4 #3 route (ii), A/B builds only (K = 50): the two WINDOW barriers of the default recurrence replaced by neighbour-wave
// synchronisation - a product reads window entries of its own and the adjacent waves only, so a wave bumps an LDS epoch word
// after its window entries are written (s_waitcnt lgkmcnt(0)) and polls the two neighbouring waves' words.  Measured: see
// DESIGN_LOG.md R5.3 (two dependent LDS round trips cost more than s_waitcnt + s_barrier).
#ifndef GATO_F64M_NBSYNC
#define GATO_F64M_NBSYNC 0
#endif
#ifndef GATO_F64M_D0
#define GATO_F64M_D0 8      // reads in flight: two-row lanes, S product (21 reads, 84 FMAs)
#endif
#ifndef GATO_F64M_D2
#define GATO_F64M_D2 8      // two-row lanes, Pinv product (60 or 63 reads, 84 FMAs)
#endif
// Round 5, VERDICT r4 #3 - both routes measured, neither kept (profiles/r05_headline_experiments.log, r05_f64m_stamps.log; DESIGN_LOG.md R5.3):
//  (i) the SINGLE-REDUCTION recurrence in this layout (one block sum of two values, three barriers, a second operand window paid
//      for by two more Pinv pairs in registers): 2.36 us per iteration with all vectors in registers (256 VGPRs: the products'
//      landing registers are gone and their LDS reads run two at a time), 1.75 with r and u re-read from the windows, 1.70 with
//      them live between the products only - against 1.55 for the default recurrence.  This layout has 12 registers to spare;
//      the recurrence needs 16 more.  pcg_variant = 1 at this shape therefore stays with the general kernel of gato_pcg_cg1.hip.
//  (ii) neighbour-wave synchronisation instead of the two window barriers (GATO_F64M_NBSYNC below): 1.684 against 1.555 - two
//      dependent LDS round trips (flag write -> visible -> poll) cost more than s_waitcnt + s_barrier (188 / 120 cycles in the
//      stamps, of which ~100 are the window write's own latency).
template <int S, int W2, int WT, int ABL = 0, int K2MAX = 0, int NPR = 0>
__global__ __launch_bounds__(64 * WT) void pcg_single_f64m_kernel(PcgLaunch a)
{
    typedef double T;
    typedef double V2 __attribute__((ext_vector_type(2)));
    constexpr int SP = pad_to(S, 2), NT = 64 * WT, L2 = 64 * W2;
    constexpr int KD = 4 * (WT - W2);                             // knots of the DPP waves (one per 16-lane row)
    constexpr int L2U = K2MAX * (S / 2);                          // two-row lanes that can own rows
    constexpr int LSTR = L2U + 1;                                 // lane slots per Pinv column in LDS (+ one slot of zeros for the idle lanes)
    constexpr int NC = 3 * S - NPR;                               // column pairs of a two-row lane in LDS (the other 3S + NPR in registers)
    constexpr int MAXK = K2MAX + KD;
    static_assert(S % 2 == 0 && W2 >= 1 && W2 < WT && WT <= 16, "two adjacent rows of one knot per lane in the first W2 waves");
    static_assert(DppRows<S>::ok && DppRows<S>::lanes == 16 && L2U <= L2 && L2U < NT && NPR >= 0 && NPR < 3 * S, "DPP rows: one knot per 16 lanes");
    // ONE operand window: p while the S product reads it, r while the Pinv product does.  Either is written after a barrier
    // behind the other's last read (the block sum's), so the two never meet - and a second window's 5.8 KB are what lets the
    // Pinv pairs of 238 two-row lanes fit the LDS with only NPR of them in registers
    __shared__ __attribute__((aligned(16))) T xs[(MAXK + 2) * SP];
    constexpr int WPW = WT < 8 ? 8 : WT;                                           // partials_total_all8 reads eight waves' slots (zeros beyond WT)
    __shared__ __attribute__((aligned(32))) T wpart[2][4 * WPW];
    static_assert(WT > 8 || sizeof(wpart) / 2 >= 32 * sizeof(T), "partials_total_all8 reads 32 values per buffer");
    __shared__ __attribute__((aligned(16))) V2 ptail[NC][LSTR];                    // Pinv entry NPR + c of (row a, row b) of a two-row lane
    static_assert(sizeof(xs) + sizeof(wpart) + sizeof(ptail) <= 160 * 1024, "LDS of one CU");

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool two = __builtin_amdgcn_readfirstlane(wave) < W2;                    // wave-uniform and known as such: scalar branches
    const int K = a.K;
    // One system: the launch brings HELPER blocks (gridDim = 1 + 8 x helpers).  In-kernel time stamps said the solving
    // workgroup spends 15.4 of its 201 us loading its 470 KB of S and Pinv - written a moment ago by the assembly launch on
    // other XCDs, so every line is an L2 miss, and one CU has at most 8 x 63 loads in flight.  The blocks dealt to the XCD of
    // block 0 (blockIdx % 8 == 0: blocks go round-robin over the XCDs) touch every 128-byte line of both arrays once, each
    // thread one line per array, and leave; the solving workgroup's loads then find the lines in that XCD's L2 or merge with
    // the misses in flight.  A speed hint only: wherever the blocks really run, nothing read or written depends on it.
    if (a.batch <= 1 && blockIdx.x > 0) {
        static_assert((size_t)WT * (4 * S * S + 6 * S) * sizeof(T) <= sizeof(ptail), "dz scratch of the helper waves");
        one_system_helper<T, S>(a, reinterpret_cast<T *>(&ptail[0][0]));
        return;
    }
    const size_t sys = a.batch > 1 ? blockIdx.x : 0;
    // lane -> row(s): the first K - KD knots (at most K2MAX) in the two-row lanes (rows 2 tid, 2 tid + 1), the last KD knots one
    // per 16-lane row of the other waves.
    int row0, j, r0, tp = tid;                                                      // first row, its knot, its row in the knot; lane slot in ptail
    bool active;
    const int K2 = K > KD ? K - KD : 0;
    if (two) {
        row0 = 2 * tid; j = row0 / S; r0 = row0 - j * S;
        active = j < K2;
        tp = tid < L2U ? tid : L2U;
    } else {
        const int q = (tid - L2) >> 4;
        r0 = tid & 15; j = K2 + q;
        active = j < K && r0 < S;
        row0 = j * S + r0;
        tp = 0;
    }
    constexpr int abl = ABL;

    const T *__restrict__ dS = static_cast<const T *>(a.S_bd) + sys * 3 * S * S * K;
    const T *__restrict__ dP = static_cast<const T *>(a.P_bd) + sys * 3 * S * S * K;
    const T *__restrict__ dG = static_cast<const T *>(a.gamma) + sys * S * K;
    T *__restrict__ dL = static_cast<T *>(a.lambda) + sys * S * K;

    // m: two-row lanes [S row a | S row b], one-row lanes [S row | Pinv row]; pr: the first NPR Pinv pairs of a two-row lane
    // The loads are issued in BATCHES with nothing that needs their data in between: written column by column (load S, load
    // Pinv, select, store the Pinv pair to LDS) the compiler reused one set of registers and waited for every column's loads
    // before the next (84 memory round trips in a row: 9 us of a 198 us launch even with every line in L2).
    // Two-row lanes keep 3S + NPR column PAIRS (row a, row b) in registers, the other 3S - NPR in LDS: the EVEN columns of both
    // matrices in registers (and the last NPR odd ones of Pinv), the odd ones in LDS - either product then reads 3S/2 windows +
    // 3S/2 pairs, one read per two FMAs throughout.
    T m[6 * S + 2 * NPR];
    auto in_reg = [](int which, int c) -> bool {
        return c % 2 == 0 || (which == 1 && c >= 3 * S - 2 * NPR);
    };
    auto reg_idx = [](int which, int c) -> int {             // pair index in m (entries 2 i, 2 i + 1)
        if (c % 2 == 0) return (which ? 3 * S / 2 : 0) + c / 2;
        return 3 * S + (c - (3 * S - 2 * NPR)) / 2;
    };
    auto lds_idx = [](int which, int c) -> int {             // column slot in ptail
        return (which ? 3 * S / 2 : 0) + (c - 1) / 2;
    };
    {
        const bool in_sys = active && j < K;
        const size_t base = (size_t)(in_sys ? j : 0) * 3 * S * S + (r0 < S ? r0 : 0);
        auto ok_col = [&](int c) { return active && !(j == 0 && c < S) && !(j == K - 1 && c >= 2 * S); };   // gato_utils.cuh:157-174
        auto keep_pair = [&](int which, int c, V2 v) {                             // pair of column c of S (0) / Pinv (1): registers or LDS slot
            if (in_reg(which, c)) { m[2 * reg_idx(which, c)] = v[0]; m[2 * reg_idx(which, c) + 1] = v[1]; }
            else ptail[lds_idx(which, c)][tp] = v;                                 // own slot (idle lanes: zeros, all into the one spare slot)
        };
        if (a.imgS != nullptr) {
            // the assembly launch of this solve also left S and Pinv transposed (column c of ALL rows contiguous, zeros where a
            // block or a row does not exist): the lane's entries are unit-stride across the wave, no boundary selects (a lane
            // without rows of its own reads rows of other lanes and drops them)
            const int rowc = active ? row0 : 0;
            const T *__restrict__ iS = static_cast<const T *>(a.imgS) + rowc, *__restrict__ iP = static_cast<const T *>(a.imgP) + rowc;
            const size_t ld = (size_t)a.img_ld;
            if (two) {
                V2 sv[3 * S];
#pragma unroll
                for (int c = 0; c < 3 * S; ++c) sv[c] = *reinterpret_cast<const V2 *>(iS + c * ld);      // row0 even, ld even: 16-byte aligned
#pragma unroll
                for (int c = 0; c < 3 * S; ++c) keep_pair(0, c, active ? sv[c] : V2{0, 0});
                constexpr int PB = 14;
#pragma unroll
                for (int c0 = 0; c0 < 3 * S; c0 += PB) {
                    V2 pv[PB];
#pragma unroll
                    for (int q = 0; q < PB; ++q) pv[q] = *reinterpret_cast<const V2 *>(iP + (c0 + q) * ld);
#pragma unroll
                    for (int q = 0; q < PB; ++q) keep_pair(1, c0 + q, active ? pv[q] : V2{0, 0});
                }
            } else {
#pragma unroll
                for (int c = 0; c < 3 * S; ++c) m[c] = iS[c * ld];
#pragma unroll
                for (int c = 0; c < 3 * S; ++c) m[3 * S + c] = iP[c * ld];
#pragma unroll
                for (int c = 0; c < 6 * S; ++c) m[c] = active ? m[c] : (T)0;
            }
        } else if (two) {
            V2 sv[3 * S];
#pragma unroll
            for (int c = 0; c < 3 * S; ++c) sv[c] = *reinterpret_cast<const V2 *>(dS + base + (size_t)c * S);   // rows r0, r0 + 1: adjacent, 16-byte aligned
#pragma unroll
            for (int c = 0; c < 3 * S; ++c) keep_pair(0, c, ok_col(c) ? sv[c] : V2{0, 0});
            constexpr int PB = 14;                                                            // Pinv pairs per batch (56 registers)
            static_assert((3 * S) % PB == 0, "batches of Pinv columns");
#pragma unroll
            for (int c0 = 0; c0 < 3 * S; c0 += PB) {
                V2 pv[PB];
#pragma unroll
                for (int q = 0; q < PB; ++q) pv[q] = *reinterpret_cast<const V2 *>(dP + base + (size_t)(c0 + q) * S);
#pragma unroll
                for (int q = 0; q < PB; ++q) keep_pair(1, c0 + q, ok_col(c0 + q) ? pv[q] : V2{0, 0});
            }
        } else {
#pragma unroll
            for (int c = 0; c < 3 * S; ++c) m[c] = dS[base + (size_t)c * S];
#pragma unroll
            for (int c = 0; c < 3 * S; ++c) m[3 * S + c] = dP[base + (size_t)c * S];
#pragma unroll
            for (int c = 0; c < 3 * S; ++c) {
                const bool ok = ok_col(c);
                m[c] = ok ? m[c] : (T)0;
                m[3 * S + c] = ok ? m[3 * S + c] : (T)0;
            }
        }
    }
    for (int i = tid; i < (MAXK + 2) * SP; i += NT) xs[i] = (T)0;
    for (int i = tid; i < 2 * 4 * WPW; i += NT) (&wpart[0][0])[i] = (T)0;
    __syncthreads();

    // the lane's Pinv pairs in LDS: column NPR + c at ptail[c][tp].  A few LDS base registers, the rest 16-bit immediate offsets.
    typedef __attribute__((address_space(3))) V2 LV2;
    constexpr int PCH = 65535 / (LSTR * 16) + 1 < NC ? 65535 / (LSTR * 16) + 1 : NC;     // columns per base register
    constexpr int NBASE = (NC + PCH - 1) / PCH;
    static_assert((PCH - 1) * LSTR * 16 < 65536 && NBASE <= 4, "immediate offsets of a base register's columns");
    LV2 *ptb[NBASE];
#pragma unroll
    for (int i = 0; i < NBASE; ++i) {
        ptb[i] = (LV2 *)&ptail[i * PCH][tp];
        if (i > 0) asm volatile("" : "+v"(ptb[i]));               // opaque: or the compiler re-derives it from ptb[0] with a 17-bit offset
    }
    auto pair_of = [&](int which, int c) -> V2 {
        if (in_reg(which, c)) return V2{m[2 * reg_idx(which, c)], m[2 * reg_idx(which, c) + 1]};
        return ptb[lds_idx(which, c) / PCH][(lds_idx(which, c) % PCH) * LSTR];
    };
    // y = [L M R]_row . window for the lane's row(s); which = 0: S, 1: Pinv; own: the lane's own entries of the operand (DPP rows)
    auto times_window = [&](int which, const T *xw, V2 own) -> V2 {
        T ya = (T)0, yb = (T)0;
        constexpr int NW = 3 * (SP / 2);                                         // window reads of a product
        if (two) {
            asm volatile("; two rows per lane" ::: "memory");
#pragma unroll
            for (int b = 0; b < 3; ++b) {
#pragma unroll
                for (int i = 0; i < SP / 2; ++i) {
                    const V2 v = *reinterpret_cast<const V2 *>(xw + b * SP + i * 2);
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int c = b * S + i * 2 + e;
                        if (i * 2 + e < S) {
                            const V2 t = pair_of(which, c);
                            ya = gato::fmaT(t[0], v[e], ya);
                            yb = gato::fmaT(t[1], v[e], yb);
                        }
                    }
                }
            }
            if (which == 0) pin_reads_then_fmas<GATO_F64M_D0, NW + 3 * S / 2, 6 * S, 4, 2>();
            else pin_reads_then_fmas<GATO_F64M_D2, NW + 3 * S / 2 - NPR, 6 * S, 4, 2>();
        } else {
            asm volatile("; one row per lane, DPP rows" ::: "memory");
            const int rc = r0 < S ? r0 : S - 1;                                  // idle lanes read inside the window (their rows are zero)
            const T x3[3] = {xw[rc], own[0], xw[2 * SP + rc]};                   // the lane's row index in knots j - 1, j, j + 1
            ya = row_times_dpp<T, S>(*reinterpret_cast<const T(*)[3 * S]>(m + (which ? 3 * S : 0)), x3);
        }
        return V2{ya, yb};
    };
#ifdef GATO_F64M_STAMP      // scratch builds only (tools/f64m_stamps.py): s_memtime at the phase boundaries of ONE iteration, per wave
    unsigned long long st_[16] = {};
    bool st_on = false;
#define GATO_ST(i) do { if (st_on) st_[i] = clock64(); } while (0)
#else
#define GATO_ST(i) do { } while (0)
#endif
    // `which`: the partials buffer of this call site - consecutive block sums alternate between the two (p . upsilon in [0],
    // r . r~ in [1]), so a wave may store its next partials while a slower one still reads the previous ones
    auto block_sum = [&](T prod, int si, int which) -> T {
        if (abl & 4) return (T)1 + prod * (T)1e-30;
        T *wp = wpart[which];
        partials_store(wp, wave, lane, prod);
#ifdef GATO_F64M_STAMP
        if (st_on) st_[si] = clock64();
#endif
        __syncthreads();
#ifdef GATO_F64M_STAMP
        if (st_on) st_[si + 1] = clock64();
#endif
        if constexpr (WT <= 8) return partials_total_all8<T>(wp, lane);        // (wpart is zeroed at set-up: waves beyond WT add zeros)
        return partials_total<T, (WT <= 8 ? 8 : 16)>(wp, WT, lane);
    };
    // own entries into the operand window.  Lanes without rows hold zeros in every vector (their matrix rows are zero): they
    // store them into the zero padding in front of knot 0 instead of sitting out behind an exec mask - no mask, no branch but the
    // wave-uniform one in the loop
    const int put_off = active ? (j + 1) * SP + r0 : (r0 < S ? r0 & ~1 : 0);
    auto put = [&](T *buf, V2 v) {                                               // (1.567 -> 1.537 us per iteration against the masked form)
        if (two) *reinterpret_cast<V2 *>(buf + put_off) = v;                    // r0 even, SP even: aligned
        else buf[put_off] = v[0];
    };
    const T *wp_ = &xs[j * SP], *wr_ = wp_;                                      // the lane's window: slot j = left neighbour

    V2 lam = {0, 0}, r = {0, 0};
    if (active) {
        r[0] = dG[(size_t)j * S + r0];
        if (two) r[1] = dG[(size_t)j * S + r0 + 1];
    }
    if (a.lambda0) {                                                             // true warm start (opt-in)
        const T *__restrict__ dL0 = static_cast<const T *>(a.lambda0) + sys * S * K;
        if (active) {
            lam[0] = dL0[(size_t)j * S + r0];
            if (two) lam[1] = dL0[(size_t)j * S + r0 + 1];
        }
        put(xs, lam);
        __syncthreads();
        r -= times_window(0, wp_, lam);
        __syncthreads();
    }
    const bool rec = a.eta_hist && tid == 0 && sys == 0;
    int iters = a.max_iters;
    const T tol = (T)a.exit_tol;
    T eta = (T)0, eta_new = (T)0;
#if GATO_F64M_NBSYNC
    __shared__ unsigned nb_flag[WT + 2];                       // [w + 1] = wave w's epoch; [0], [WT + 1]: always current
    if (tid < WT + 2) nb_flag[tid] = (tid == 0 || tid == WT + 1) ? 0xffffffffu : 0u;
    __syncthreads();
    unsigned nb_seq = 0;
    auto window_sync = [&]() {
        ++nb_seq;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this wave's window entries are written
        volatile unsigned *f = nb_flag;
        if (lane == 0) f[wave + 1] = nb_seq;
        for (;;) {
            const unsigned l = f[wave], r_ = f[wave + 2];
            if (l >= nb_seq && r_ >= nb_seq) break;
        }
        asm volatile("" ::: "memory");
    };
#else
    auto window_sync = [&]() { __syncthreads(); };
#endif
    put(xs, r);
    __syncthreads();
    V2 rt = times_window(1, wr_, r);                                             // gato_pcg.cuh:316-335
    eta = block_sum(r[0] * rt[0] + r[1] * rt[1], 0, 1);
    if (rec) a.eta_hist[0] = (double)eta;
    V2 p = rt, ups;
    put(xs, p);
    __syncthreads();
    for (int it = 0; it < a.max_iters; ++it) {                                   // gato_pcg.cuh:348
#ifdef GATO_F64M_STAMP
        st_on = it == a.max_iters - 2;
#endif
        GATO_ST(0);
        ups = (abl & 1) ? p * m[0] : times_window(0, wp_, p);                    // upsilon = S p         (:349-351)
        GATO_ST(1);
        const T v = block_sum(p[0] * ups[0] + p[1] * ups[1], 2, 0);              // v = p . upsilon       (:353-357)
        GATO_ST(4);
        const T alpha = quotient(eta, v);                                        // :364
        lam += alpha * p;                                                        // :373-377
        r -= alpha * ups;
        put(xs, r);
        GATO_ST(5);
        if (!(abl & 8)) window_sync();
        GATO_ST(6);
        rt = (abl & 2) ? r * m[1] : times_window(1, wr_, r);                     // r~ = Pinv r           (:380-381)
        GATO_ST(7);
        eta_new = block_sum(r[0] * rt[0] + r[1] * rt[1], 8, 1);                  // eta' = r . r~         (:382-394)
        GATO_ST(10);
        if (rec) a.eta_hist[it + 1] = (double)eta_new;
        if (__builtin_amdgcn_readfirstlane((int)(fabs(eta_new) < tol))) { iters = it; break; }   // :404-411 (eta' is the same in every lane)
        const T beta = quotient(eta_new, eta);                                   // :415
        p = rt + beta * p;                                                       // :416-419
        put(xs, p);
        eta = eta_new;                                                           // :420
        GATO_ST(11);
        if (!(abl & 8)) window_sync();
        GATO_ST(12);
    }
#ifdef GATO_F64M_STAMP
    if (a.eta_hist && lane == 0 && sys == 0)
        for (int i = 0; i < 13; ++i) a.eta_hist[1024 + wave * 16 + i] = (double)(st_[i] & 0xffffffffffull);
#endif
    if (active) {                                                                // :433-435
        dL[(size_t)j * S + r0] = lam[0];
        if (two) dL[(size_t)j * S + r0 + 1] = lam[1];
    }
    if (a.dz_helpers && a.dz != nullptr && a.batch <= 1) {       // lambda is complete: release it and tell the helper blocks
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) __hip_atomic_store((gi32 *)a.dz_flag, a.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- dz back-substitution in the same launch (batches: one workgroup per system).  Formulas and accumulation order of
    // dz_kernel (gato_assembly.hip; gato_schur.cuh:758-867, D2 fixed), row by row: bit-identical results.
    if (a.dz != nullptr && !a.dz_helpers) {
        const int Cn = a.C, n = S + Cn, k = j, nrow = two ? 2 : 1;
        const size_t gs = (size_t)(S * S + Cn * Cn), cs = (size_t)(S * S + S * Cn), Nn = (size_t)n * K - Cn;
        const T *__restrict__ Gi = static_cast<const T *>(a.dz_Ginv) + sys * (gs * K - (size_t)Cn * Cn);
        const T *__restrict__ Cdn = static_cast<const T *>(a.dz_Cd) + sys * (cs * (K - 1));
        const T *__restrict__ gv = static_cast<const T *>(a.dz_g) + sys * Nn;
        T *__restrict__ dzo = static_cast<T *>(a.dz) + sys * Nn;
        const bool last = k == K - 1;
        __syncthreads();                                                         // every wave has left the loop: the window and the Pinv pairs are free
        T *xs0 = xs, *xs1 = reinterpret_cast<T *>(&ptail[0][0]);                 // two windows for the back-substitution
        static_assert(sizeof(xs) <= sizeof(ptail), "second window of the dz epilogue");
        put(xs0, lam);                                                           // lambda window
        __syncthreads();
        T tx[2] = {0, 0}, tu[2] = {0, 0};
        if (active) {
            for (int q = 0; q < nrow; ++q) {
                const int rr = r0 + q;
                if (!last) {
                    const T *__restrict__ A = Cdn + (size_t)k * cs;
                    const T *lp = &xs0[(j + 2) * SP];                          // lambda_{k+1}
                    T res = (T)0;
#pragma unroll
                    for (int t = 0; t < S; ++t) res = gato::fmaT(A[rr * S + t], lp[t], res);          // A_k^T lambda_{k+1}   :833-838
                    tx[q] = gv[(size_t)k * n + rr] - (lam[q] + res);                                  // :841-852
                    if (rr < Cn) {
                        const T *__restrict__ B = A + S * S;
                        T rb = (T)0;
#pragma unroll
                        for (int t = 0; t < S; ++t) rb = gato::fmaT(B[rr * S + t], lp[t], rb);        // B_k^T lambda_{k+1}   :784-789
                        tu[q] = gv[(size_t)k * n + S + rr] - rb;                                      // :792-796
                    }
                } else tx[q] = gv[(size_t)k * n + rr] - lam[q];                                       // last state row (D2)
                xs1[(j + 1) * SP + rr] = tx[q];
            }
        }
        __syncthreads();                                                         // lambda_{k+1} has been read everywhere
        if (active && !last) {
            for (int q = 0; q < nrow; ++q)
                if (r0 + q < Cn) xs0[(j + 1) * SP + r0 + q] = tu[q];
        }
        __syncthreads();
        if (active) {
            const T *__restrict__ Qi = Gi + (size_t)k * gs;
            const T *tv = &xs1[(j + 1) * SP];
            for (int q = 0; q < nrow; ++q) {
                const int rr = r0 + q;
                T res = (T)0;
#pragma unroll
                for (int cc = 0; cc < S; ++cc) res = gato::fmaT(Qi[rr + cc * S], tv[cc], res);        // Q_k^-1 (...)         :856-865
                dzo[(size_t)k * n + rr] = res;
                if (!last && rr < Cn) {
                    const T *__restrict__ Ri = Qi + S * S;
                    const T *uv = &xs0[(j + 1) * SP];
                    T ru = (T)0;
                    for (int cc = 0; cc < Cn; ++cc) ru = gato::fmaT(Ri[rr + cc * Cn], uv[cc], ru);    // R_k^-1 (...)         :799-808
                    dzo[(size_t)k * n + S + rr] = ru;
                }
            }
        }
    }
    if (tid == 0) {
        // a helper block that gave up on this launch (see one_system_helper) left dz rows unwritten: in-band, as a hand-off time-out
        const bool dz_lost = a.dz_helpers && a.dz != nullptr && a.batch <= 1 &&
                             __hip_atomic_load((gi32 *)a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.launch_id;
        a.iters[sys] = dz_lost ? -1 : iters;
        if (a.final_eta && sys == 0) *a.final_eta = (double)eta_new;
    }
}

// shape of the mixed kernel per STATE_SIZE (0 = none): waves with two rows per lane, waves in all (the others own 16-lane DPP rows,
// one knot each), k2 = knots the two-row lanes can take (LDS), npr = their Pinv columns in registers
template <int S> struct MixedCfg { static constexpr int w2 = 0, wt = 0, k2 = 0, npr = 0; };
template <> struct MixedCfg<14> { [[maybe_unused]] static constexpr int w2 = 4, wt = 8, k2 = 34, npr = 1; };   // 34 + 4 x 4 = 50 knots
template <int S> constexpr int mixed_rows()
{
    if (MixedCfg<S>::wt <= 0) return 0;
    return (MixedCfg<S>::k2 + 4 * (MixedCfg<S>::wt - MixedCfg<S>::w2)) * S;
}

template <int S> struct PairThreads { static constexpr int v = 4 * S - 2 > 64 ? 0 : (12 * S + 3 * S + 48) <= 256 ? 512 : ((12 * S + 3 * S + 48) <= 512 ? 256 : 0); };   // (private windows: one halo row per lane)
template <> struct PairThreads<14> { static constexpr int v = 512; };     // measured: 248 VGPRs, no spill at the 256 cap

// Generic rule for shapes added at build time: VGPRs per lane ~ matrix rows (6S words, x2 for fp64) + the
// operand window the compiler keeps in flight (3S words) + ~40; the specialisations below are the measured ones.
template <typename T, int S> struct MaxThreads {
    static constexpr int regs = (6 * S + 3 * S) * (int)(sizeof(T) / 4) + 40;
    static constexpr int v = regs <= 128 ? 1024 : regs <= 168 ? 768 : regs <= 256 ? 512 : 256;
    static_assert(regs <= 512, "STATE_SIZE too large for the register-resident PCG");
};
// VGPR budget: 3S*2 matrix registers per lane (x2 for fp64).  launch bound -> registers per lane:
// 1024 threads -> 128, 768 -> 168, 512 -> 256, 256 -> 512 (MI355X register file: 512 per lane per SIMD).
// Chosen so that the matrix rows plus the 3S-wide operand window stay in registers without spilling.
template <> struct MaxThreads<float, 2> { static constexpr int v = 1024; };
template <> struct MaxThreads<double, 2> { static constexpr int v = 1024; };
template <> struct MaxThreads<float, 14> { static constexpr int v = 768; };
template <> struct MaxThreads<double, 14> { static constexpr int v = 512; };
template <> struct MaxThreads<float, 32> { static constexpr int v = 512; };
template <> struct MaxThreads<double, 32> { static constexpr int v = 256; };

}  // namespace

// Semi-resident variant (XR extra rows per lane): workgroup size with room for the extra rows' registers.
// Semi-resident variant: workgroup size by register need (resident rows 6S words + one streamed row 3S + ~80; two waves
// per SIMD when that fits 256 registers, else one), and extra rows per lane (their four state vectors take 64 KB of LDS).
template <typename T, int S> struct SemiThreads {
    static constexpr int need = 9 * S * (int)(sizeof(T) / 4) + 80;
    static constexpr int t = need <= 256 ? 512 : (need <= 512 ? 256 : 0);
    static constexpr int v = (t > 0 && MaxThreads<T, S>::v >= t && t >= 2 * S) ? t : 0;
};
template <typename T, int S> struct SemiRows {       // by LDS: two operand windows over all local knots + lambda and product of the extra rows
    static constexpr int t = SemiThreads<T, S>::v;
    static constexpr int maxk = (t + S - 1) / S, sp = pad_to(S, VecOf<T>::W), w = (int)sizeof(T);
    static constexpr int per_row = 2 * maxk * sp * w + 2 * t * w;
    static constexpr int fit = t > 0 ? (148 * 1024 - 2 * (maxk + 2) * sp * w) / per_row : 0;
    static constexpr int v = fit > 32 ? 32 : fit;
};

// No-resident-rows variant (NR): workgroup size by the registers one streamed row needs, rows per lane by LDS.
template <typename T, int S> struct NoresThreads {
    static constexpr int need = 4 * S * (int)(sizeof(T) / 4) + 70;
    static constexpr int v = 2 * S > 64 ? 0 : (need <= 120 ? 1024 : need <= 160 ? 768 : need <= 250 ? 512 : 256);
};
template <typename T, int S> struct NoresRows {
    static constexpr int t = NoresThreads<T, S>::v;
    static constexpr int maxk = t > 0 ? (t + S - 1) / S : 1, sp = pad_to(S, VecOf<T>::W), w = (int)sizeof(T);
    static constexpr int per_row = 2 * maxk * sp * w + 2 * t * w;
    static constexpr int fit = t > 0 ? (148 * 1024 - 4 * sp * w) / per_row : 0;
    static constexpr int v = fit > 32 ? 32 : fit;
};

// Single-workgroup variants with part of the Pinv rows in LDS: (threads, NL).
template <typename T, int S> struct SingleCu { static constexpr int threads = 0, nl = 0; };
template <> struct SingleCu<double, 14> { static constexpr int threads = 704, nl = 24; };   // IIWA 14/7/50 fp64

// The plain and the cluster launches (every row register resident): geometry check and the hand-off form.  DR: the DPP-row
// layout (its instantiations are compiled in gato_pcg_resident_dpp.hip, which includes this file).
template <typename T, int S, bool DR>
int launch_plain(const PcgLaunch &a, bool mr, int Kl, hipStream_t st)
{
    constexpr int LPK = DR ? DppRows<S>::lanes : S;
    constexpr int MAXT0 = MaxThreads<T, S>::v;
    constexpr int SINGLE_T = SingleCu<T, S>::threads;
    const bool single_lds = !DR && !mr && SINGLE_T > MAXT0 && a.groups == 1 && a.threads > MAXT0 && a.threads <= SINGLE_T;
    const int MAXT = single_lds ? SINGLE_T : MAXT0;
    if (a.batch > 1 && a.groups != 1) {
        set_error("pcg_resident: a batch needs one workgroup per system");
        return GATO_EINVAL;
    }
    if ((DR && a.stamps) || a.threads > MAXT || a.threads % 64 != 0 || a.threads < 2 * S || a.knots_per_wg * LPK > a.threads ||
        a.groups < 1 || a.groups > 256 || (long long)a.groups * a.knots_per_wg < Kl ||
        (long long)(a.groups - 1) * a.knots_per_wg >= Kl) {
        set_error("pcg_resident: bad launch geometry (K=%d groups=%d knots/wg=%d threads=%d max=%d)", a.K,
                  a.groups, a.knots_per_wg, a.threads, MAXT);
        return GATO_EINVAL;
    }
    // no re-initialisation of the hand-off area: granules carry epochs from the solver's ever-growing counter and
    // the status word is matched against this launch's id
    if (a.ev_start) GATO_HIP_CHECK(hipEventRecord(a.ev_start, st));
    const int nblocks = a.batch > 1 ? a.batch
                      : (a.xcd_pack > 0 ? 8 * ((a.groups + a.xcd_pack - 1) / a.xcd_pack) : a.groups);
    if constexpr (SINGLE_T > 0 && !DR) {
        if (single_lds) {
            constexpr int NL = SingleCu<T, S>::nl;
            if (a.stamps) hipLaunchKernelGGL((pcg_resident_kernel<T, S, SINGLE_T, NL, 1>), dim3(nblocks), dim3(a.threads), 0, st, a);
            else if (a.diag == 2) hipLaunchKernelGGL((pcg_resident_kernel<T, S, SINGLE_T, NL, 2>), dim3(nblocks), dim3(a.threads), 0, st, a);
            else hipLaunchKernelGGL((pcg_resident_kernel<T, S, SINGLE_T, NL, false>), dim3(nblocks), dim3(a.threads), 0, st, a);
            GATO_HIP_CHECK(hipGetLastError());
            if (a.ev_stop) GATO_HIP_CHECK(hipEventRecord(a.ev_stop, st));
            return GATO_OK;
        }
    }
    // Option coop_launch (A12: cudaLaunchCooperativeKernel + check_sms, gato_pcg.cuh:502-526, gato_utils.cuh:829-854): the
    // multi-workgroup persistent kernels through hipLaunchCooperativeKernel, so that the RUNTIME keeps the launch from
    // starting before all its workgroups can be resident (kernels of other streams and processes included), instead of this
    // library's own gate over its own launches.  Measured cost and verdict: DESIGN.md 3.1b.
#define GATO_LAUNCH_P(KERNEL, grid_, block_, st_, a_)                                                                          \
    do {                                                                                                                       \
        if ((a_).coop) {                                                                                                       \
            PcgLaunch arg_ = (a_);                                                                                             \
            void *args_[] = {(void *)&arg_};                                                                                   \
            (void)hipLaunchCooperativeKernel(reinterpret_cast<const void *>(&KERNEL), grid_, block_, args_, 0, st_);           \
        } else hipLaunchKernelGGL(KERNEL, grid_, block_, 0, st_, a_);                                                          \
    } while (0)
    // Hand-off form of the plain and the cluster launches (option wave_pub, default 1): ghost blocks in registers always;
    // per-wave published partials where a sweep - W << ceil(log2(waves)) granules - is at most 4 loads per lane (up to 32
    // workgroups of 8 waves).  wave_pub = 0: the gathered form with the ghost blocks staged in LDS (also what the cycle-stamp
    // build, DIAG = 1, runs).
    const int nw_ = a.threads / 64, wsh_ = nw_ <= 1 ? 0 : 32 - __builtin_clz((unsigned)(nw_ - 1));
    const bool rg = a.batch <= 1 && a.wave_pub != 0 && !a.stamps && (a.groups > 1 || mr);
    const bool wp = rg && a.groups > 1 && nw_ * (int)(sizeof(T) / 4) <= 16 && (a.groups << wsh_) <= 256 && a.wave_pub != 3;
    const dim3 grid(nblocks), block(a.threads);
    // launch bound of the instantiation: shapes whose bound is above 512 threads (fp32, S <= 16: 768 threads = 168 registers per
    // lane) also exist with a bound of 512 (256 registers) for the launches that fit it - most multi-workgroup launches are 512
    // threads, and the hand-off's loop-invariant offsets do not fit 168 registers beside the matrix rows (spills)
    auto go = [&](auto mtc) {
        constexpr int MT = decltype(mtc)::value;
        if (mr) {
            if (wp) GATO_LAUNCH_P((pcg_resident_kernel<T, S, MT, 0, 0, 0, false, true, 4, DR>), grid, block, st, a);
            else if (rg) GATO_LAUNCH_P((pcg_resident_kernel<T, S, MT, 0, 0, 0, false, true, -1, DR>), grid, block, st, a);
            else GATO_LAUNCH_P((pcg_resident_kernel<T, S, MT, 0, 0, 0, false, true, 0, DR>), grid, block, st, a);
        } else if (a.stamps) {
            if constexpr (!DR) GATO_LAUNCH_P((pcg_resident_kernel<T, S, MAXT0, 0, 1>), grid, block, st, a);
        } else if (a.diag == 2) {
            if (wp) GATO_LAUNCH_P((pcg_resident_kernel<T, S, MT, 0, 2, 0, false, false, 4, DR>), grid, block, st, a);
            else if (rg) GATO_LAUNCH_P((pcg_resident_kernel<T, S, MT, 0, 2, 0, false, false, -1, DR>), grid, block, st, a);
            else GATO_LAUNCH_P((pcg_resident_kernel<T, S, MT, 0, 2, 0, false, false, 0, DR>), grid, block, st, a);
        } else if (wp) GATO_LAUNCH_P((pcg_resident_kernel<T, S, MT, 0, 0, 0, false, false, 4, DR>), grid, block, st, a);
        else if (rg) GATO_LAUNCH_P((pcg_resident_kernel<T, S, MT, 0, 0, 0, false, false, -1, DR>), grid, block, st, a);
        else GATO_LAUNCH_P((pcg_resident_kernel<T, S, MT, 0, 0, 0, false, false, 0, DR>), grid, block, st, a);
    };
    constexpr bool HAS512 = MAXT0 > 512 && S >= 12;
    if constexpr (HAS512) {
        if (a.threads <= 512 && a.batch <= 1) go(std::integral_constant<int, 512>{});
        else go(std::integral_constant<int, MAXT0>{});
    } else go(std::integral_constant<int, MAXT0>{});
    GATO_HIP_CHECK(hipGetLastError());
    if (a.ev_stop) GATO_HIP_CHECK(hipEventRecord(a.ev_stop, st));
    return GATO_OK;
}

template <typename T, int S>
int pcg_resident_plan(PcgPlan *plan)
{
    plan->max_threads = MaxThreads<T, S>::v;
    plan->max_knots_per_wg = MaxThreads<T, S>::v / S;
    plan->single_max_threads = SingleCu<T, S>::threads;
    plan->pair_threads = (sizeof(T) == 4 && S % 2 == 0) ? PairThreads<S>::v : 0;
    plan->mixed_rows = sizeof(T) == 8 ? mixed_rows<S>() : 0;
    plan->mixed_threads = sizeof(T) == 8 ? 64 * MixedCfg<S>::wt : 0;
    plan->semi_threads = SemiThreads<T, S>::v;
    plan->semi_rows = SemiRows<T, S>::v;
    plan->nores_threads = NoresThreads<T, S>::v;
    plan->nores_rows = NoresRows<T, S>::v;
    plan->dpp_lanes = DppRows<S>::ok ? DppRows<S>::lanes : 0;
    return GATO_OK;
}

template <typename T, int S>
int launch_pcg_resident(const PcgLaunch &a0, hipStream_t st)
{
    // cluster launch (one rank of a multi-GPU solve): the MR instantiations, geometry over the rank's knot range
    const bool mr = a0.xslots != nullptr;
    PcgLaunch a = a0;
    if (!mr) { a.k_begin = 0; a.k_end = a.K; a.rank = 0; a.nranks = 1; }
    const int Kl = a.k_end - a.k_begin;                 // knots this launch works on
    if (mr && (a.pair || a.batch > 1 || a.xcd_pack || a.stamps || a.nranks < 1 || a.nranks > GATO_MAX_RANKS || a.rank < 0 ||
               a.rank >= a.nranks || a.k_begin < 0 || Kl < 1 || a.k_end > a.K || (a.rank == 0) != (a.k_begin == 0) ||
               (a.rank == a.nranks - 1) != (a.k_end == a.K))) {
        set_error("pcg_resident(cluster): bad shard rank=%d/%d knots [%d,%d) of %d", a.rank, a.nranks, a.k_begin, a.k_end, a.K);
        return GATO_EINVAL;
    }
    if (a.dpp_rows) {
        if (a.pair || a.semi) {
            set_error("pcg_resident: the DPP-row layout serves the plain and the cluster launches only");
            return GATO_EINVAL;
        }
        return launch_pcg_resident_dpp<T, S>(a, st);
    }
    // the one-workgroup kernels with two rows per lane (pair = 1: fp32; 2: fp64 mixed rows): a translation unit of their own
    if (a.pair) return launch_pcg_single<T, S>(a, mr, st);
    if constexpr (SemiThreads<T, S>::v > 0) {
        if (a.semi == 1) {
            constexpr int XT = SemiThreads<T, S>::v;
            const long long extra_rows = ((long long)a.knots_per_wg - a.threads / S) * S;
            if (a.batch > 1 || a.threads != XT || a.groups < (mr ? 1 : 2) || a.groups > 256 || a.threads / S < 2 ||
                extra_rows > (long long)SemiRows<T, S>::v * a.threads || (long long)a.groups * a.knots_per_wg < Kl ||
                (long long)(a.groups - 1) * a.knots_per_wg >= Kl) {
                set_error("pcg_resident(semi): bad launch geometry (K=%d groups=%d knots/wg=%d threads=%d)", a.K, a.groups,
                          a.knots_per_wg, a.threads);
                return GATO_EINVAL;
            }
            if (a.ev_start) GATO_HIP_CHECK(hipEventRecord(a.ev_start, st));
            if (mr) GATO_LAUNCH_P((pcg_resident_kernel<T, S, XT, 0, false, SemiRows<T, S>::v, false, true>), dim3(a.groups), dim3(a.threads), st, a);
            else GATO_LAUNCH_P((pcg_resident_kernel<T, S, XT, 0, false, SemiRows<T, S>::v>), dim3(a.groups), dim3(a.threads), st, a);
            GATO_HIP_CHECK(hipGetLastError());
            if (a.ev_stop) GATO_HIP_CHECK(hipEventRecord(a.ev_stop, st));
            return GATO_OK;
        }
    }
    if constexpr (NoresThreads<T, S>::v > 0) {
        if (a.semi == 2) {
            constexpr int NT = NoresThreads<T, S>::v, NX = NoresRows<T, S>::v;
            if (a.batch > 1 || a.threads != NT || a.groups < (mr ? 1 : 2) || a.groups > 256 ||
                (long long)a.knots_per_wg * S > (long long)NX * NT || (long long)a.groups * a.knots_per_wg < Kl ||
                (long long)(a.groups - 1) * a.knots_per_wg >= Kl) {
                set_error("pcg_resident(no resident rows): bad launch geometry (K=%d groups=%d knots/wg=%d threads=%d)", a.K,
                          a.groups, a.knots_per_wg, a.threads);
                return GATO_EINVAL;
            }
            if (a.ev_start) GATO_HIP_CHECK(hipEventRecord(a.ev_start, st));
            if (mr) GATO_LAUNCH_P((pcg_resident_kernel<T, S, NT, 0, false, NX, true, true>), dim3(a.groups), dim3(a.threads), st, a);
            else GATO_LAUNCH_P((pcg_resident_kernel<T, S, NT, 0, false, NX, true>), dim3(a.groups), dim3(a.threads), st, a);
            GATO_HIP_CHECK(hipGetLastError());
            if (a.ev_stop) GATO_HIP_CHECK(hipEventRecord(a.ev_stop, st));
            return GATO_OK;
        }
    }
    return launch_plain<T, S, false>(a, mr, Kl, st);
}

#if defined(GATO_RESIDENT_SINGLE_PART)
// gato_pcg_resident_single.hip: the launches of the one-workgroup two-rows-per-lane kernels (compile time: a third of this file's)
template <typename T, int S>
int launch_pcg_single(const PcgLaunch &a, bool mr, hipStream_t st)
{
    if constexpr (sizeof(T) == 4 && S % 2 == 0 && PairThreads<S>::v > 0) {
        if (a.pair) {
            constexpr int PT = PairThreads<S>::v;
            if (a.groups != 1 || a.threads > PT || a.threads % 64 != 0 || a.K * (S / 2) > a.threads) {
                set_error("pcg_resident(pair): bad geometry K=%d threads=%d max=%d", a.K, a.threads, PT);
                return GATO_EINVAL;
            }
            if (a.ev_start) GATO_HIP_CHECK(hipEventRecord(a.ev_start, st));
            // one system: + helper blocks (enough waves for one knot each: they also do dz), see pcg_single_f64m_kernel
            const int helpers = (a.K + a.threads / 64 - 1) / (a.threads / 64);
            const dim3 grid(a.batch > 1 ? a.batch : 1 + 8 * helpers);
            hipLaunchKernelGGL((pcg_single_f32x2_kernel<S, PT>), grid, dim3(a.threads), 0, st, a);
            GATO_HIP_CHECK(hipGetLastError());
            if (a.ev_stop) GATO_HIP_CHECK(hipEventRecord(a.ev_stop, st));
            return GATO_OK;
        }
    }
    if constexpr (sizeof(T) == 8 && MixedCfg<S>::wt > 0) {
        if (a.pair == 2) {
            constexpr int W2 = MixedCfg<S>::w2, WT = MixedCfg<S>::wt, K2 = MixedCfg<S>::k2, NPR = MixedCfg<S>::npr;
            if (mr || a.groups != 1 || a.threads != 64 * WT || a.K * S > mixed_rows<S>() || a.stamps) {
                set_error("pcg_resident(mixed): bad launch K=%d threads=%d", a.K, a.threads);
                return GATO_EINVAL;
            }
            if (a.ev_start) GATO_HIP_CHECK(hipEventRecord(a.ev_start, st));
            static_assert(GATO_L2_HELPERS * WT * S >= mixed_rows<S>(), "one helper wave per knot (they do dz)");
            const dim3 grid(a.batch > 1 ? a.batch : 1 + 8 * GATO_L2_HELPERS), block(64 * WT);          // one system: + helper blocks that warm the L2
            const int abl = a.diag == 2 ? a.ablate : 0;
            if (abl == 3) hipLaunchKernelGGL((pcg_single_f64m_kernel<S, W2, WT, 3, K2, NPR>), grid, block, 0, st, a);
            else if (abl == 4) hipLaunchKernelGGL((pcg_single_f64m_kernel<S, W2, WT, 4, K2, NPR>), grid, block, 0, st, a);
            else if (abl == 15) hipLaunchKernelGGL((pcg_single_f64m_kernel<S, W2, WT, 15, K2, NPR>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((pcg_single_f64m_kernel<S, W2, WT, GATO_F64M_DEV_ABL, K2, NPR>), grid, block, 0, st, a);
            GATO_HIP_CHECK(hipGetLastError());
            if (a.ev_stop) GATO_HIP_CHECK(hipEventRecord(a.ev_stop, st));
            return GATO_OK;
        }
    }
    set_error("pcg_resident: no one-workgroup two-rows-per-lane kernel for this shape and type (pair=%d)", a.pair);
    return GATO_EINVAL;
}
#define X(S_, C_)                                                          \
    template int launch_pcg_single<float, S_>(const PcgLaunch &, bool, hipStream_t); \
    template int launch_pcg_single<double, S_>(const PcgLaunch &, bool, hipStream_t);
GATO_SHAPES(X)
#undef X
#elif !defined(GATO_RESIDENT_DPP_PART)
#define X(S_, C_)                                                      \
    template int pcg_resident_plan<float, S_>(PcgPlan *);              \
    template int pcg_resident_plan<double, S_>(PcgPlan *);             \
    template int launch_pcg_resident<float, S_>(const PcgLaunch &, hipStream_t); \
    template int launch_pcg_resident<double, S_>(const PcgLaunch &, hipStream_t);
GATO_SHAPES(X)
#undef X
#else
// gato_pcg_resident_dpp.hip: the DPP-row instantiations of the plain and the cluster launches, a translation unit of their own
// (compile time).  `a` arrives checked and normalised by launch_pcg_resident.
template <typename T, int S>
int launch_pcg_resident_dpp(const PcgLaunch &a, hipStream_t st)
{
    if constexpr (DppRows<S>::ok) return launch_plain<T, S, true>(a, a.xslots != nullptr, a.k_end - a.k_begin, st);
    else {
        set_error("pcg_resident: no DPP-row layout for STATE_SIZE %d", S);
        return GATO_EINVAL;
    }
}
#define X(S_, C_)                                                          \
    template int launch_pcg_resident_dpp<float, S_>(const PcgLaunch &, hipStream_t); \
    template int launch_pcg_resident_dpp<double, S_>(const PcgLaunch &, hipStream_t);
GATO_SHAPES(X)
#undef X
#endif

}  // namespace gato
