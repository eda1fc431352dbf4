// Persistent PCG for K far beyond the register file, matrices streamed by LDS-DMA: the HBM-bound end of the range.
//
// Replaces parallelPCG_fixed (src/gato_pcg.cuh:17-268, the grid-stride variant that re-loads S and Pinv from global
// memory every iteration, D1 fixed) where even the semi-resident launch (gato_pcg_resident.hip, XR > 0) re-reads nearly
// every block row per product.  There the block rows reach the lanes as 4-byte loads at stride S (a lane walks its
// row), which keeps the L1 at 83 % of its sector rate and the HBM at 5.5 TB/s.  Here:
//  * one persistent launch, one workgroup of 8 waves per CU, two hand-offs per iteration - the protocol of
//    pcg_resident_kernel ({epoch, payload} granules, partial + boundary blocks, ghosts advanced locally);
//  * the block rows of a workgroup stream through a RING of three 37 KB tiles in LDS (16 block rows in fp32, 8 in fp64),
//    filled by global_load_lds_dwordx4 (1 KiB per wave-instruction, no
//    registers, whole cache lines), NB-1 tiles ahead of the one being multiplied.  The ring simply keeps going from the
//    last tile of S into the first tile of Pinv and back: the matrices do not depend on the hand-off, so the next
//    product's tiles land while the workgroups wait for each other (the register-load variants drain every CU's load
//    pipeline twice per iteration).  Measured: ~25 GB/s of LDS-DMA per CU whatever the ring shape -
//    smaller tiles with a deeper ring, or re-filling a slot as soon as its tile is in registers, are slower;
//  * ALL vector state lives in registers: worker lane l owns the rows l, l+448, l+896, ... of its workgroup (448 = 32 S
//    at S = 14, so a lane always has the same row of knots 32 apart) and keeps r, p, lambda and the product of each;
//    LDS holds only the operand window of the CURRENT product (p or r) beside the tile ring;
//  * wave 0 does not stream: it runs the hand-off (its polling loads must not queue behind a DMA ring).
// Same iteration, same boundary rules (gato_utils.cuh:157-174) and same exit test (gato_pcg.cuh:404) as everywhere.
#include "gato_pcg_device.h"

namespace gato {
namespace {

// largest divisor of n that is <= cap
constexpr int largest_divisor_upto(int n, int cap)
{
    int best = 1;
    for (int d = 1; d <= n && d <= cap; ++d)
        if (n % d == 0) best = d;
    return best;
}

// fp64 at S = 32: builds (-DGATO_DMA_F64_S32=1: 206 / 214 VGPRs, no spill, with the row read in chunks below) and
// is correct (rel diff 6e-16 against the launch without resident rows), but LOSES by 2.6x: a block row is 24 KB, so a tile is ONE
// block row (two would leave no LDS for the window), 32 of the 384 computing lanes work per tile step and a step is that half
// wave's chain of 96 dependent fp64 FMAs behind a barrier.  Measured (profiles/r05_ring_crossover.log, us per iteration): 32/16/4096
// ring 98.1 against 34.2 without resident rows, 8192: 187.6 / 76.3, 16384: 365.8 / 141.0 - the latter is 6.1 TB/s of algorithmic
// bytes (0.76 of 8 TB/s) already.  Not built by default; what would beat 0.76 is a tile step that uses all lanes (12 lanes per row
// and a cross-lane sum: another summation order), not this ring.
#ifndef GATO_DMA_F64_S32
#define GATO_DMA_F64_S32 0
#endif
template <typename T, int S>
struct DmaCfg {
    static constexpr int VW = VecOf<T>::W;
    static constexpr int SP = pad_to(S, VW);
    static constexpr int ROW = 3 * S * S;                               // elements of one block row [left|main|right]
    // computing lanes (of the 448 lanes of waves 1..7, all of which move DMA pieces): a whole number of knots, chosen so
    // that a round of them splits into tiles of ~37 KB - 448 = 32 knots at S = 14 (tiles of 16 / 8 block rows); at S = 32, 448
    // = 14 knots would leave tiles of 2 block rows (24 KB: step-bound, 166 us per iteration at K = 32 768 against 137 semi-
    // resident), 384 = 12 knots gives tiles of 3 (36 KB) and wave 7 only streams
    static constexpr int WL = S == 32 ? 384 : 448;
    static constexpr int KPR = WL / S;                                  // knots per round of computing lanes (S = 14: 32, S = 32: 12)
    // Tile = TK whole block rows, TK a divisor of KPR (so that a tile is a whole number of the workers' rows and a round a
    // whole number of tiles), as large as ~40 KB allow: 37 632 B at S = 14 in either type (fp32 16 block rows, fp64 8);
    // 24 576 B at S = 32 (fp32 2 block rows, fp64 1).  Ring of three.
    static constexpr int ROW_BYTES = ROW * (int)sizeof(T);
    static constexpr int TK = largest_divisor_upto(KPR, 40 * 1024 / ROW_BYTES > 0 ? 40 * 1024 / ROW_BYTES : 1);
    static constexpr int TROWS = TK * S;
    static constexpr int SUBS = WL / TROWS;                             // tiles per round of worker lanes
    static constexpr int NB = 3;
    static constexpr int TILE = TK * ROW;                               // elements
    static constexpr int TILE_BYTES = TILE * (int)sizeof(T);
    static constexpr int PIECES = (TILE_BYTES + 1023) / 1024;           // 1 KiB DMA pieces per tile
    static constexpr int NHI = (PIECES + 6) / 7, NLO = PIECES / 7;      // DMA instructions per tile of a worker wave: NHI or NLO
    // rounds = rows per worker lane: by registers (r, p, lambda, y per row beside the 3S matrix entries of the row being
    // multiplied) and by the LDS the operand window takes beside the ring
    static constexpr int XR_REG = S <= 16 ? (sizeof(T) == 4 ? 16 : 8) : 12;   // S = 32 fp32: 241 VGPRs at 12 rows, spills at 16
    static constexpr int XR_LDS = (150 * 1024 - NB * TILE_BYTES - 2 * SP * (int)sizeof(T)) / (KPR * SP * (int)sizeof(T));
    static constexpr int XR = XR_REG < XR_LDS ? XR_REG : XR_LDS;
    static constexpr int MAXK = XR * KPR;                               // knots per workgroup (S = 14: fp32 512 - K <= 131 072 on 256 CUs - fp64 256)
    static constexpr bool BLOCKWISE = 3 * S * (int)(sizeof(T) / 4) > 128;   // the row's entries leave the tile block by block (registers)
    static constexpr bool OK = WL % S == 0 && WL <= 448 && KPR % TK == 0 && 2 * S <= 64 && S >= 8 && TILE_BYTES <= 48 * 1024 && XR >= 2 &&
                               (GATO_DMA_F64_S32 || !(S > 16 && sizeof(T) == 8));   // fp64 at S = 32: see GATO_DMA_F64_S32
};

// MR: one rank of a cluster launch (gato_cluster_pcg) - the second hand-off level of pcg_resident_kernel<..., MR>: the rank's
// total into every rank's mirror, its edge blocks into the neighbouring ranks' mirrors, polls on the own mirror only.
template <typename T, int S, bool MR = false>
__global__ __launch_bounds__(512) void pcg_dma_kernel(PcgLaunch a)
{
    typedef DmaCfg<T, S> Cfg;
    typedef Granule<T> Gr;
    typedef GranuleSys<T> XGr;
    constexpr int SP = Cfg::SP, GPV = Gr::GPV, XR = Cfg::XR, TK = Cfg::TK, ROW = Cfg::ROW, NB = Cfg::NB;
    constexpr int AHEAD = NB - 1;                                          // tiles in flight behind the one being multiplied
    static_assert((AHEAD - 1) * Cfg::NHI <= 63, "vmcnt is a 6-bit counter");
    constexpr int KPR = Cfg::KPR, SUBS = Cfg::SUBS, PM = 4;
    static_assert(Cfg::OK, "shape not supported by the DMA variant");
    static_assert(SUBS * Cfg::TROWS == Cfg::WL && XR * SUBS >= 1, "a round of worker lanes is a whole number of tiles");

    __shared__ __attribute__((aligned(16))) T tiles[NB][Cfg::TILE];        // the ring
    __shared__ __attribute__((aligned(16))) T win[(Cfg::MAXK + 2) * SP];   // operand window of the current product
    __shared__ T ghost[2][2][S];           // [r | p][left | right]: ghost blocks of the neighbouring workgroups
    __shared__ T gh[2][32];                // neighbours' boundary blocks of the vector just formed
    __shared__ T yedge[2][S];              // own first / last block of it
    __shared__ __attribute__((aligned(32))) T wpart[2][32];
    __shared__ T bc[2];
    __shared__ int s_abort;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool worker = wave >= 1;
    const int wl = tid - 64;                                               // worker lane
    const int wg = blockIdx.x, W = gridDim.x;
    const int K = a.K;
    const int k_begin = MR ? a.k_begin : 0, k_end = MR ? a.k_end : K;      // this launch's knot range (a rank's shard)
    const int R = MR ? a.nranks : 1;
    const int k0 = k_begin + wg * a.knots_per_wg;
    const int nk = min(a.knots_per_wg, k_end - k0);
    const bool cw = worker && wl < Cfg::WL;                                // computing lane (owns rows)
    const int jl = cw ? wl / S : 0, rr = cw ? wl - jl * S : 0;             // the lane's knot within a round, its row
    const int NT = (nk + TK - 1) / TK;                                     // tiles per product
    const bool has_left = k0 > 0, has_right = k0 + nk < K;                 // a neighbouring block row exists in the SYSTEM ...
    const bool loc_left = MR ? wg > 0 : has_left, loc_right = MR ? wg < W - 1 : has_right;     // ... in a workgroup of this launch,
    const bool x_left = MR && wg == 0 && has_left, x_right = MR && wg == W - 1 && has_right;   // or on the neighbouring GPU

    const T *__restrict__ dS = static_cast<const T *>(a.S_bd);
    const T *__restrict__ dP = static_cast<const T *>(a.P_bd);
    const T *__restrict__ dG = static_cast<const T *>(a.gamma);
    T *__restrict__ dL = static_cast<T *>(a.lambda);

    const int slotG = pcg_slot_granules(S, (int)sizeof(T));
    gu64 *slots = (gu64 *)a.slots;
    gi32 *g_status = (gi32 *)a.status;
    const unsigned long long t_limit = a.timeout_ticks;
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

    // ---- state: r = gamma, lambda = 0 (gato_pcg.cuh:300-304) ----
    T r[XR], p[XR], lam[XR], y[XR];
#pragma unroll
    for (int e = 0; e < XR; ++e) {
        const int j = jl + KPR * e;
        const bool on = cw && j < nk;
        r[e] = on ? dG[(size_t)(k0 + j) * S + rr] : (T)0;
        p[e] = (T)0; lam[e] = (T)0; y[e] = (T)0;
    }
    if (tid < S) {
        ghost[0][0][tid] = has_left ? dG[(size_t)(k0 - 1) * S + tid] : (T)0;
        ghost[1][0][tid] = (T)0;
    } else if (tid < 2 * S) {
        ghost[0][1][tid - S] = has_right ? dG[(size_t)(k0 + nk) * S + (tid - S)] : (T)0;
        ghost[1][1][tid - S] = (T)0;
    }

    // ---- the ring: global tile G = tile (G % NT) of matrix (G / NT) & 1 ? S : Pinv (the solve starts with Pinv r) ----
    // wave w (1..7) moves the pieces w-1, w+6, w+13 of a tile: n_mine DMA instructions per tile
    const int n_mine = !worker ? 0 : (Cfg::PIECES - (wave - 1) + 6) / 7;
    auto issue = [&](unsigned G) {
        if (!worker) return;
        const unsigned prod = G / (unsigned)NT, t = G - prod * (unsigned)NT;
        const T *M = (prod & 1u) ? dS : dP;
        const int kt = (int)t * TK;
        const int bytes = min(TK, nk - kt) * ROW * (int)sizeof(T);
        const char *src = (const char *)(M + (size_t)(k0 + kt) * ROW);
        // LDS byte address of the slot: the ring's base (a constant: the local address of a __shared__ object) plus the slot
        // offset - formed arithmetically, a generic -> local cast of a run-time pointer trips the compiler in the S = 32 build
        const unsigned slot0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)&tiles[0][0] + (G % NB) * (unsigned)Cfg::TILE_BYTES;
        for (int q = wave - 1; q < Cfg::PIECES; q += 7) {
            const int off = q * 1024 + lane * 16;
            // Lanes past the end of a short last tile re-read the tile's first KiB (a block row is > 1 KiB) into the unused
            // tail of the slot: every wave issues the same number of DMA instructions for every tile, which is what the
            // counted waits below rely on.  Lanes past the slot itself (last piece) stay off.
            if (off < Cfg::TILE_BYTES) {
                const int so = off < bytes ? off : lane * 16;
                // inline asm: hipcc counts a builtin LDS-DMA against every later LDS read (s_waitcnt vmcnt(0) in front of the
                // tile reads = no ring at all); an asm load is outside its bookkeeping, the counted waits below are ours
                const char *gsrc = src + so;
                const unsigned ldst = __builtin_amdgcn_readfirstlane(slot0 + (unsigned)q * 1024u);
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(gsrc), "s"(ldst) : "memory");
            }
        }
    };
    unsigned G = 0;                                                        // next tile to be multiplied
    for (unsigned i = 0; i < (unsigned)AHEAD; ++i) issue(i);
    __syncthreads();

    unsigned epoch = a.epoch0;
    bool aborted = false;

    // One block-tridiagonal product with the ring's next NT tiles: x = p (which = 1) or r (which = 0) -> y, and this
    // lane's share of x . y.  Window layout as in pcg_resident_kernel: slot 0 = left ghost, slot j+1 = knot j.
    auto product = [&](const T (&x)[XR], int which) -> T {
#pragma unroll
        for (int e = 0; e < XR; ++e) {
            const int j = jl + KPR * e;
            if (cw && j < nk) win[(j + 1) * SP + rr] = x[e];
        }
        if (tid < S) win[tid] = ghost[which][0][tid];
        else if (tid < 2 * S) win[(nk + 1) * SP + (tid - S)] = ghost[which][1][tid - S];
        T dot = (T)0;
#pragma unroll
        for (int e = 0; e < XR; ++e) {
            if (e * SUBS >= NT) break;                                     // workgroup-uniform
#pragma unroll 1
            for (int sub = 0; sub < SUBS; ++sub) {
                const int t = e * SUBS + sub;
                if (t >= NT) break;
                // my pieces of tile G have landed once at most the DMA instructions of the NB-2 tiles behind it are
                // still in flight (loads return in order); the barrier then covers every wave's pieces - and says that
                // every lane has finished with the tile before, whose slot the tile three ahead may now take
                if (n_mine == Cfg::NHI) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((AHEAD - 1) * Cfg::NHI) : "memory");
                else if (n_mine == Cfg::NLO) asm volatile("s_waitcnt vmcnt(%0)" ::"i"((AHEAD - 1) * Cfg::NLO) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                issue(G + AHEAD);
                const int j = jl + KPR * e;
                const bool mine = cw && wl >= sub * Cfg::TROWS && wl < (sub + 1) * Cfg::TROWS && j < nk;
                if constexpr (Cfg::BLOCKWISE) {
                    // a row of 3S doubles at S = 32 is 192 VGPRs: in chunks of 16 columns instead (16 entries out of the tile + their
                    // window entries, then the 16 FMAs; the same left-to-right order, gato_utils.cuh:177-183).  A real loop: unrolled,
                    // the scheduler hoists every chunk's reads in front of the one dependent FMA chain and spills 160 registers
                    if (mine) {
                        typedef typename VecOf<T>::type V;
                        constexpr int VW = VecOf<T>::W, CH = 16;
                        static_assert(S % CH == 0 && SP == S, "whole chunks per block");
                        const unsigned ts = G % NB;
                        const T *mrow = &tiles[ts][(jl % TK) * ROW + rr];
                        const T *xw = &win[j * SP];
                        const bool nl = k0 + j == 0, nr = k0 + j == K - 1;
                        T acc = (T)0;
#pragma unroll 1
                        for (int c0 = 0; c0 < 3 * S; c0 += CH) {
                            T mb[CH];
#pragma unroll
                            for (int c = 0; c < CH; ++c) mb[c] = mrow[(c0 + c) * S];
                            const bool drop = (c0 < S && nl) || (c0 >= 2 * S && nr);     // never written blocks (gato_utils.cuh:157-174)
#pragma unroll
                            for (int i = 0; i < CH / VW; ++i) {
                                const V v = *reinterpret_cast<const V *>(xw + c0 + i * VW);
#pragma unroll
                                for (int e = 0; e < VW; ++e) acc = gato::fmaT(drop ? (T)0 : mb[i * VW + e], v[e], acc);
                            }
                        }
                        y[e] = acc;
                        dot = gato::fmaT(x[e], acc, dot);
                    }
                    ++G;
                    continue;
                }
                // all 3S entries of the row out of the tile first (independent LDS reads in flight together), then the FMAs
                T m[3 * S];
                if (mine) {
                    const unsigned ts = G % NB;
                    const int m0 = (jl % TK) * ROW + rr;
#pragma unroll
                    for (int c = 0; c < 3 * S; ++c) m[c] = tiles[ts][m0 + c * S];
                }
                if (mine) {
                    const T *xw = &win[j * SP];
                    const bool nl = k0 + j == 0, nr = k0 + j == K - 1;    // first / last block row of the system
#pragma unroll
                    for (int c = 0; c < S; ++c) {                                  // never written blocks (gato_utils.cuh:157-174)
                        m[c] = nl ? (T)0 : m[c];
                        m[2 * S + c] = nr ? (T)0 : m[2 * S + c];
                    }
                    const T acc = row_times_window<T, S, SP>(m, xw);
                    y[e] = acc;
                    dot = gato::fmaT(x[e], acc, dot);
                }
                ++G;
            }
        }
        return dot;
    };

    // Reduction + halo exchange: the protocol of pcg_resident_kernel (allreduce_and_halo), wave 0 only on the fabric.
    auto exchange = [&](T prod, T &total) {
        ++epoch;
        partials_store(wpart[epoch & 1], wave, lane, prod);
        // own first / last block of the vector just formed, for the neighbours
#pragma unroll
        for (int e = 0; e < XR; ++e) {
            const int j = jl + KPR * e;
            if (cw && j < nk) {
                if (j == 0) yedge[0][rr] = y[e];
                if (j == nk - 1) yedge[1][rr] = y[e];
            }
        }
        __syncthreads();
        if constexpr (MR) ++xepoch;
        if (wave == 0) {
            T tot = partials_total(wpart[epoch & 1], 8, lane);
            gu64 *mine = slots + ((size_t)(epoch & 1) * W + wg) * slotG;
            bool fail = false;
            if constexpr (MR) {             // the rank's edge blocks go straight into the neighbouring GPU's mirror
                if (x_left && lane < S) XGr::store(xp_prev + (size_t)(xepoch & 1) * xslotG + xghR + lane * GPV, xepoch, yedge[0][lane]);
                if (x_right && lane >= 32 && lane < 32 + S)
                    XGr::store(xp_next + (size_t)(xepoch & 1) * xslotG + xghL + (lane - 32) * GPV, xepoch, yedge[1][lane - 32]);
            }
            if (W > 1) {
                if (lane < S) Gr::store(mine + 16 + lane * GPV, epoch, yedge[0][lane]);
                else if (lane >= 32 && lane < 32 + S) Gr::store(mine + 16 + (S + lane - 32) * GPV, epoch, yedge[1][lane - 32]);
                if (lane == 0) Gr::store(mine, epoch, tot);
                gu64 *pbase = slots + (size_t)(epoch & 1) * W * slotG;
                const bool want_l = loc_left && lane < S;
                const bool want_r = loc_right && lane >= 32 && lane < 32 + S;
                gu64 *hptr = want_l ? pbase + (size_t)(wg - 1) * slotG + 16 + (S + lane) * GPV
                           : want_r ? pbase + (size_t)(wg + 1) * slotG + 16 + (lane - 32) * GPV
                                    : mine;
                gu64 *pptr[PM];
#pragma unroll
                for (int m = 0; m < PM; ++m) pptr[m] = pbase + (size_t)min(lane + 64 * m, W - 1) * slotG;
                const int pm_count = (W + 63) >> 6;
                unsigned long long raw[PM][GPV], hraw[GPV];
                const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                if (W > 32) __builtin_amdgcn_s_sleep(12);
                for (unsigned spin = 0;; ++spin) {
#pragma unroll
                    for (int m = 0; m < PM; ++m) {
                        if (m < pm_count) {
#pragma unroll
                            for (int g = 0; g < GPV; ++g) raw[m][g] = __hip_atomic_load(pptr[m] + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
#pragma unroll
                    for (int g = 0; g < GPV; ++g) hraw[g] = __hip_atomic_load(hptr + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    bool ok = true;
#pragma unroll
                    for (int m = 0; m < PM; ++m) {
                        if (m < pm_count) {
#pragma unroll
                            for (int g = 0; g < GPV; ++g) ok &= (unsigned)(raw[m][g] >> 32) == epoch;
                        }
                    }
#pragma unroll
                    for (int g = 0; g < GPV; ++g) ok &= (unsigned)(hraw[g] >> 32) == epoch;
                    if (__all(ok)) break;
                    if ((spin & 255u) == 255u) {
                        const bool late = __builtin_amdgcn_s_memrealtime() - t0 > t_limit;
                        const bool other = __hip_atomic_load(g_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.launch_id;
                        if (late || other) { fail = true; break; }
                    }
                }
                T acc = (T)0;
#pragma unroll
                for (int m = 0; m < PM; ++m) acc += (m < pm_count && lane + 64 * m < W) ? Gr::decode(raw[m]) : (T)0;
                const T hv = Gr::decode(hraw);
                tot = wave_sum(acc);
                if (lane < S) gh[0][lane] = want_l ? hv : (T)0;
                if (lane >= 32 && lane < 32 + S) gh[1][lane - 32] = want_r ? hv : (T)0;
            } else {                       // one workgroup on this GPU (cluster launch): the ghosts come from level 2 only
                if (lane < S) gh[0][lane] = (T)0;
                if (lane >= 32 && lane < 32 + S) gh[1][lane - 32] = (T)0;
            }
            if constexpr (MR) {
                if (R > 1 && !fail) {
                    // level 2, across the GPUs of the node (pcg_resident_kernel<..., MR>): tot = this rank's total
                    const size_t xo = (size_t)(xepoch & 1) * xslotG;
                    if (wg == 0 && lane < R) XGr::store((gu64 *)s_xpeer[lane] + xo + a.rank * 16, xepoch, tot);
                    gu64 *xl = (gu64 *)a.xslots + xo;                     // polls stay on THIS GPU's memory
                    const bool xw_l = x_left && lane < S;
                    const bool xw_r = x_right && lane >= 32 && lane < 32 + S;
                    gu64 *tptr = xl + (size_t)min(lane, R - 1) * 16;
                    gu64 *xhp = xw_l ? xl + xghL + lane * GPV : xw_r ? xl + xghR + (lane - 32) * GPV : tptr;
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
                    if (xw_l) gh[0][lane] = xv;
                    if (xw_r) gh[1][lane - 32] = xv;
                }
            }
            if (fail && lane == 0) {
                __hip_atomic_store(g_status, a.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_abort = 1;
            }
            if (lane == 0) bc[epoch & 1] = tot;
        }
        __syncthreads();
        total = bc[epoch & 1];
        aborted = s_abort != 0;
    };

    // ---- r~ = Pinv r ; p = r~ ; eta = r . r~   (gato_pcg.cuh:316-335) ----
    T eta = (T)0, eta_new = (T)0;
    int iters = a.max_iters;
    const T tol = (T)a.exit_tol;
    exchange(product(r, 0), eta);
    const bool rec = a.eta_hist != nullptr && wg == 0 && tid == 0;
    if (rec) a.eta_hist[0] = (double)eta;
    if (!aborted) {
#pragma unroll
        for (int e = 0; e < XR; ++e) p[e] = y[e];
        if (tid < S) ghost[1][0][tid] = gh[0][tid];
        else if (tid < 2 * S) ghost[1][1][tid - S] = gh[1][tid - S];
        for (int it = 0; it < a.max_iters; ++it) {                              // gato_pcg.cuh:348
            T v;
            exchange(product(p, 1), v);                                         // upsilon = S p ; v = p . upsilon   (:349-357)
            if (aborted) break;
            const T alpha = quotient(eta, v);                                    // :364
#pragma unroll
            for (int e = 0; e < XR; ++e) {                                      // :373-377
                lam[e] += alpha * p[e];
                r[e] -= alpha * y[e];
            }
            if (tid < S) ghost[0][0][tid] -= alpha * gh[0][tid];                // ghost r advances with the neighbours' upsilon
            else if (tid < 2 * S) ghost[0][1][tid - S] -= alpha * gh[1][tid - S];
            exchange(product(r, 0), eta_new);                                   // r~ = Pinv r ; eta' = r . r~       (:380-394)
            if (aborted) break;
            if (rec) a.eta_hist[it + 1] = (double)eta_new;
            if (fabs(eta_new) < tol) { iters = it; break; }                     // :404-411
            const T beta = quotient(eta_new, eta);                               // :415
#pragma unroll
            for (int e = 0; e < XR; ++e) p[e] = y[e] + beta * p[e];             // :416-419
            if (tid < S) ghost[1][0][tid] = gh[0][tid] + beta * ghost[1][0][tid];
            else if (tid < 2 * S) ghost[1][1][tid - S] = gh[1][tid - S] + beta * ghost[1][1][tid - S];
            eta = eta_new;                                                      // :420
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // no DMA may land after the workgroup has gone
    __syncthreads();
#pragma unroll
    for (int e = 0; e < XR; ++e) {
        const int j = jl + KPR * e;
        if (cw && j < nk) dL[(size_t)(k0 + j) * S + rr] = lam[e];               // :433-435
    }
    if constexpr (MR) (void)cluster_lambda_ghost<T, S>(a, wg, W, dL, aborted);      // lambda_{k_end} for this rank's dz launch
    if (wg == 0 && tid == 0) {
        a.iters[0] = aborted ? -1 : iters;
        if (a.final_eta) *a.final_eta = (double)eta_new;
    }
}

}  // namespace

template <typename T, int S>
int pcg_dma_max_knots()
{
    if constexpr (DmaCfg<T, S>::OK) return DmaCfg<T, S>::MAXK;
    else return 0;
}

template <typename T, int S>
int launch_pcg_dma(const PcgLaunch &a0, hipStream_t st)
{
    if constexpr (DmaCfg<T, S>::OK) {
        typedef DmaCfg<T, S> Cfg;
        const bool mr = a0.xslots != nullptr;                   // one rank of a cluster launch
        PcgLaunch a = a0;
        if (!mr) { a.k_begin = 0; a.k_end = a.K; a.rank = 0; a.nranks = 1; }
        const int Kl = a.k_end - a.k_begin;
        if (mr && (a.nranks < 1 || a.nranks > GATO_MAX_RANKS || a.rank < 0 || a.rank >= a.nranks || a.k_begin < 0 || Kl < 1 || a.k_end > a.K ||
                   (a.rank == 0) != (a.k_begin == 0) || (a.rank == a.nranks - 1) != (a.k_end == a.K) || a.flat)) {
            set_error("pcg_dma(cluster): bad shard rank=%d/%d knots [%d,%d) of %d", a.rank, a.nranks, a.k_begin, a.k_end, a.K);
            return GATO_EINVAL;
        }
        if (a.batch > 1 || a.lambda0 || a.groups < (mr ? 1 : 2) || a.groups > 256 || a.threads != 512 || a.knots_per_wg > Cfg::MAXK ||
            a.knots_per_wg < 1 || (long long)a.groups * a.knots_per_wg < Kl || (long long)(a.groups - 1) * a.knots_per_wg >= Kl ||
            (reinterpret_cast<uintptr_t>(a.S_bd) & 15) || (reinterpret_cast<uintptr_t>(a.P_bd) & 15)) {
            set_error("pcg_dma: bad launch (K=%d groups=%d knots/wg=%d threads=%d)", a.K, a.groups, a.knots_per_wg, a.threads);
            return GATO_EINVAL;
        }
        if (a.ev_start) GATO_HIP_CHECK(hipEventRecord(a.ev_start, st));
        if (mr) hipLaunchKernelGGL((pcg_dma_kernel<T, S, true>), dim3(a.groups), dim3(512), 0, st, a);
        else hipLaunchKernelGGL((pcg_dma_kernel<T, S, false>), dim3(a.groups), dim3(512), 0, st, a);
        GATO_HIP_CHECK(hipGetLastError());
        if (a.ev_stop) GATO_HIP_CHECK(hipEventRecord(a.ev_stop, st));
        return GATO_OK;
    } else {
        set_error("pcg_dma: STATE_SIZE %d has no DMA variant", S);
        return GATO_EINVAL;
    }
}

#define X(S_, C_)                                                              \
    template int pcg_dma_max_knots<float, S_>();                               \
    template int pcg_dma_max_knots<double, S_>();                              \
    template int launch_pcg_dma<float, S_>(const PcgLaunch &, hipStream_t);    \
    template int launch_pcg_dma<double, S_>(const PcgLaunch &, hipStream_t);
GATO_SHAPES(X)
#undef X

}  // namespace gato
