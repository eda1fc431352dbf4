// Internal header of libgato_hip.so (gfx950 only).  Not part of the C ABI (include/gato_hip.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/gato_hip.h"

// (S, C) instantiations compiled in.  The reference is compiled for exactly one triple
// (CMakeLists.txt:18); here K is a runtime value and (S, C) picks one of these.
// Add shapes at build time with  make EXTRA_SHAPES="X(12,6) X(6,3)"  (S <= 32, S even, C <= S).
#ifndef GATO_EXTRA_SHAPES
#define GATO_EXTRA_SHAPES(X)
#endif
// default build: pendulum (2/1, the reference's test), IIWA-14 (14/7, its default), 32/16 (BASELINE config 5) and a few
// common robot sizes (tools/devbuild.sh predefines GATO_SHAPES with fewer shapes for a quick A/B library)
#ifndef GATO_SHAPES
#define GATO_SHAPES(X) X(2, 1) X(14, 7) X(32, 16) X(4, 2) X(6, 3) X(12, 6) GATO_EXTRA_SHAPES(X)
#endif

#define GATO_MAX_RANKS 8     /* GPUs of one node a cluster launch spans (gato_cluster_*) */

namespace gato {

void set_error(const char *fmt, ...);

// fused multiply-add in the operand type (__builtin_fma alone would promote floats to double)
__device__ __forceinline__ float fmaT(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fmaT(double a, double b, double c) { return __builtin_fma(a, b, c); }

#define GATO_HIP_CHECK(expr)                                                                  \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            ::gato::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                              __LINE__);                                                      \
            return GATO_EHIP;                                                                 \
        }                                                                                     \
    } while (0)

// ---- wave64 sum on the DPP network (no LDS traffic): quad swaps, half-row / row mirrors, then the
// gfx9 row broadcasts; the total lands in lane 63 and is returned wave-uniform.  Fixed order, so the
// result is bitwise reproducible.  (A __shfl_xor butterfly compiles to six dependent ds_bpermute_b32,
// each an LDS-crossbar round trip.)
// FULL = every lane has a valid source and all rows take part (quad permutes, mirrors): the move is issued with
// bound_ctrl, which makes the `old` operand dead so that the compiler does not zero a register pair before every
// DPP move (matters for f64, which has no DPP add: 8 v_mov less per wave sum).  The row broadcasts keep old = 0.
template <int CTRL, int ROW_MASK, bool FULL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, FULL));
}
template <int CTRL, int ROW_MASK, bool FULL>
__device__ __forceinline__ double dpp_mov(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, 0xf, FULL);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROW_MASK, 0xf, FULL);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float lane63(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ double lane63(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float lane0(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
}
__device__ __forceinline__ double lane0(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, 0);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 0);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// Sum of the values in lanes 0..15 (the per-wave partials of a workgroup: at most 16 waves), wave-uniform: the four
// intra-row steps only - the two row broadcasts of the full wave sum are a third of its dependent chain.
// every lane ends with the sum of its row of 16 lanes
template <typename T>
__device__ __forceinline__ T row_sums_dpp(T v)
{
    v += dpp_mov<0xB1, 0xf, true>(v);
    v += dpp_mov<0x4E, 0xf, true>(v);
    v += dpp_mov<0x141, 0xf, true>(v);
    v += dpp_mov<0x140, 0xf, true>(v);
    return v;
}
// N: how many of the lanes 0..15 can hold a value (the others are zero): 8 values need three steps, 4 need two
template <typename T, int N = 16>
__device__ __forceinline__ T row0_sum_dpp(T v)
{
    v += dpp_mov<0xB1, 0xf, true>(v);
    v += dpp_mov<0x4E, 0xf, true>(v);
    if constexpr (N > 4) v += dpp_mov<0x141, 0xf, true>(v);
    if constexpr (N > 8) v += dpp_mov<0x140, 0xf, true>(v);
    return lane0(v);
}
template <typename T>
__device__ __forceinline__ T wave_sum_dpp(T v)
{
    v += dpp_mov<0xB1, 0xf, true>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E, 0xf, true>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141, 0xf, true>(v);   // row_half_mirror
    v += dpp_mov<0x140, 0xf, true>(v);   // row_mirror      -> every lane holds its row's sum
    v += dpp_mov<0x142, 0xa, false>(v);  // row_bcast:15 into rows 1,3
    v += dpp_mov<0x143, 0xc, false>(v);  // row_bcast:31 into rows 2,3 -> lane 63 holds the wave sum
    return lane63(v);
}

// Device layout sizes in elements (gato_defines.h:32-37, gpu_library.cu:40-45).
// B > 1: a batch of B independent systems of the same shape and sparsity pattern.  Every
// per-system array is stored B times back to back (system stride = its single-system size); the CSR structure
// (indptr / indices) is shared, the value arrays have nnzG / nnzC entries per system.
struct Dims {
    int S, C, K;
    int B = 1;
    int nnzG = 0, nnzC = 0;
    // stage-level entries only: knots [k_lo, k_hi) to work on (k_hi = 0: all).  A rank of a multi-GPU solve assembles
    // just the block rows its PCG shard reads (gato_python_amd/dist.py: assemble_shard)
    int k_lo = 0, k_hi = 0;
    int stair_follows = 0;      // internal (whole-solve stage path): the Schur launch need not zero the blocks the stair launch writes
    __host__ __device__ int lo() const { return k_lo; }
    __host__ __device__ int hi() const { return k_hi > 0 ? k_hi : K; }
    __host__ __device__ int n() const { return S + C; }
    __host__ __device__ size_t N() const { return (size_t)(S + C) * K - C; }
    __host__ __device__ size_t g_dense() const { return (size_t)(S * S + C * C) * K - C * C; }
    __host__ __device__ size_t c_dense() const { return (size_t)(S * S + S * C) * (K > 0 ? K - 1 : 0); }
    __host__ __device__ size_t bd() const { return (size_t)3 * S * S * K; }
    __host__ __device__ size_t sk() const { return (size_t)S * K; }
};

struct BatchStride {
    size_t g, c, bd, sk, n, nnzG, nnzC;     // elements per system: G_dense, C_dense, S/Pinv, gamma/lambda, g/dz, CSR values
    int k_lo, k_hi;                         // knots the launch works on
    int stair_follows;
};
inline BatchStride batch_stride(const Dims &d)
{
    return BatchStride{d.g_dense(), d.c_dense(), d.bd(), d.sk(), d.N(), (size_t)d.nnzG, (size_t)d.nnzC, d.lo(), d.hi(), 0};
}

// ---- persistent (resident) PCG launch description ------------------------------------------
struct PcgLaunch {
    const void *S_bd, *P_bd, *gamma;
    // optional (one-workgroup two-rows-per-lane kernels, one system): the same matrices TRANSPOSED - entry (row R = k S + r of
    // the system, column c of its block row) at img[c * img_ld + R], rows beyond K S and the blocks a first / last block row
    // lacks are zero - written by assemble_kernel beside S_bd / P_bd, so that a lane's 3S loads are coalesced
    const void *imgS, *imgP;
    int img_ld;
    void *lambda;
    const void *lambda0;         // initial guess (true warm start, r0 = gamma - S lambda0) or nullptr = cold start
    int K;
    int max_iters;
    double exit_tol;
    int batch;                   // > 1: blockIdx.x = system index, one workgroup per system (groups must be 1)
    int pair;                    // fp32 one-workgroup kernel with two rows per lane
    int xcd_pack;                // 2..32 workgroups: place them on one XCD (grid 8x oversubscribed, 7 of 8 blocks exit)
    int xcd_sel;                 // 0..7: the XCD (blockIdx % 8) that hosts them
    int wave_pub;                // launches of 2..32 workgroups: every wave publishes its own partial (no gather barrier); 0 = gathered form
    int knots_per_wg;            // contiguous knots owned by each workgroup (last may own fewer)
    int split_extra;             // pcg_cg1 only, set by its launcher: > 0 = balanced split - the first split_extra workgroups own
                                 // knots_per_wg knots, the others one less (the even split would leave the last workgroup ONE knot)
    int groups;                  // W = gridDim.x
    int threads;                 // blockDim.x (multiple of 64, >= knots_per_wg * S unless semi)
    int dpp_rows;                // 1: DPP-row layout (a knot owns whole 16-lane rows; threads >= knots_per_wg * 16 or 32), plain and cluster launches
    int semi;                    // 1: semi-resident launch (knots_per_wg exceeds the lanes, the rest are extra rows); 2: no resident rows; 3: LDS-DMA ring (gato_pcg_dma.hip)
    unsigned long long *slots;   // hand-off granules: every 8-byte word {epoch, payload}; epochs only grow, so no re-zeroing
    unsigned epoch0;             // this launch uses epochs epoch0+1 .. (the solver hands out disjoint ranges)
    // knot range [k_begin, k_end) of the system this launch works on (single GPU: 0, K).  Multi-GPU cluster launch
    // (gato_cluster_pcg): rank r of nranks runs this kernel on ITS range; K stays the global knot count, S_bd / P_bd /
    // gamma / lambda are full-system arrays of which only the rows of the range are touched.
    int k_begin, k_end;
    int rank, nranks;            // nranks > 1: second hand-off level through the cross-GPU mirrors below
    unsigned long long *xslots;                    // this rank's mirror (fine-grained memory, written by the peers)
    unsigned long long *const *xpeer;              // DEVICE table [nranks]: every rank's mirror as mapped into this process (xpeer[rank] == xslots)
    // optional: dz back-substitution (compute_dz, gato_schur.cuh:758-867) in the epilogue of a ONE-workgroup launch (the
    // workgroup holds every lambda_k): Ginv / C_dense / g in the layouts of gato_compute_dz, C = CONTROL_SIZE
    const void *dz_Ginv, *dz_Cd, *dz_g;
    void *dz;
    int C;
    int dz_helpers;              // 1: the helper blocks of the one-workgroup fp64 launch do the dz back-substitution once *dz_flag == launch_id
    int *dz_flag;                // device word: the solving workgroup stores launch_id there when lambda is complete
    // flat variant of the cluster exchange (ranks x workgroups <= 256): every workgroup stores its partial straight into
    // every rank's mirror and its boundary blocks into its own and (at the rank's edges) the neighbour's - ONE level, no
    // wait for the rank's own gather first.  The flat area follows the two-level area in every mirror.
    int flat, flat_groups, flat_base;
    size_t flat_off;             // granules from the start of a mirror to its flat area
    unsigned xepoch0;            // cross-GPU epochs: in lock-step on all ranks (only cluster launches draw from this counter)
    // cluster launches, at exit: the rank's first lambda block goes into the LEFT neighbour's mirror (granules at lam_off, tag
    // lam_tag: unique per launch, the same on every rank) and the rank's last workgroup stores the right neighbour's block at
    // lambda[k_end] - dz of the rank's last knot needs lambda_{k_end} (gato_schur.cuh:833-838), so no lambda all-reduce stands
    // between the PCG and the dz launch of a sharded solve.  lam_tag = 0: off.
    size_t lam_off;
    unsigned lam_tag;
    int launch_id;               // > 0; a timed-out hand-off stores it into *status (stale ids of earlier launches are ignored)
    int *iters;                  // device
    int *status;                 // device, 0 ok / 1 timeout
    double *final_eta;           // device (optional)
    double *eta_hist;            // device (optional): eta after init [0] and after iteration i [i+1] (system 0 of a batch)
    unsigned long long timeout_ticks;  // s_memrealtime ticks (100 MHz)
    hipEvent_t ev_start, ev_stop;      // optional: recorded right around the kernel launch
    int ablate;                        // diagnostic: timing-only ablation mask (0 in production)
    unsigned long long *stamps;        // optional: diagnostic cycle stamps (16 words), selects the DIAG = 1 build
    int diag;                          // 2: the build with the timing-only switches (ablate) but no stamps
    int coop;                          // multi-workgroup persistent launches through hipLaunchCooperativeKernel (option coop_launch)
};

// Cross-GPU mirror of a cluster launch, per epoch parity (granules): one 128-B line per rank for its total (written by
// that rank's workgroup 0 into EVERY rank's mirror), then the left and the right ghost S-block (written by the
// neighbouring rank's last / first workgroup).  Every line has exactly one writer.
// (two S-blocks per side: the single-reduction recurrence exchanges the first / last TWO blocks of w, gato_pcg_cg1.hip)
__host__ __device__ inline int pcg_xghost_granules(int S, int esz) { return ((2 * S * (esz / 4) + 15) / 16) * 16; }
__host__ __device__ inline int pcg_xslot_granules(int S, int esz) { return 16 * GATO_MAX_RANKS + 2 * pcg_xghost_granules(S, esz); }

// Granules (8 B: {epoch:32 | payload:32}) per workgroup and parity in the hand-off area.
// Line 0 (16 granules) holds the partial dot, then first-block and last-block halos.
__host__ __device__ inline int pcg_slot_granules(int S, int esz)
{
    int gpv = esz / 4;                       // granules per value
    int halo = 2 * S * gpv;
    return 16 + ((halo + 15) / 16) * 16;
}

// Hand-off slot of the single-reduction variant: line 0 = {gamma', delta}, then the first two and last two S-blocks of w.
__host__ __device__ inline int pcg_slot_granules_cg1(int S, int esz)
{
    int gpv = esz / 4;
    return 16 + ((4 * S * gpv + 15) / 16) * 16;
}

// A workgroup's slot in the FLAT area of a cluster mirror: room for either recurrence's slot.
__host__ __device__ inline int pcg_flat_slot_granules(int S, int esz)
{
    const int a = pcg_slot_granules(S, esz), b = pcg_slot_granules_cg1(S, esz);
    return a > b ? a : b;
}
// Granules of the lambda ghost block behind the flat area (one S-block, whole 128-B lines).
__host__ __device__ inline int pcg_lamghost_granules(int S, int esz) { return ((S * (esz / 4) + 15) / 16) * 16; }

struct PcgPlan {
    int max_threads;     // launch bound of the instantiation
    int max_knots_per_wg;
    int single_max_threads;   // > max_threads: a one-workgroup variant (Pinv rows partly in LDS) exists up to this size
    int pair_threads;         // > 0: fp32 one-workgroup kernel with two rows per lane, up to this many threads
    int mixed_rows, mixed_threads;   // > 0: fp64 one-workgroup kernel with two rows per lane in some waves (pcg_single_f64m_kernel)
    int semi_threads;         // > 0: semi-resident variant (extra rows re-read from memory): its workgroup size ...
    int semi_rows;            // ... and the extra rows a lane can take
    int nores_threads;        // > 0: variant without resident rows: workgroup size ...
    int nores_rows;           // ... and rows per lane
    int dpp_lanes;            // > 0: the plain and cluster launches exist in the DPP-row layout, with this many lanes per knot
};

// Per-(dtype, S, C) kernel launchers, defined in the .hip files and instantiated for GATO_SHAPES.
template <typename T, int S, int C>
int launch_convert(const Dims &d, const int *G_row, const int *G_col, const T *G_val, const int *C_row,
                   const int *C_col, const T *C_val, T rho, T *Gd, T *Cd, T *Ginv /* optional: also invert Q_k, R_k */,
                   hipStream_t st);
template <typename T, int S, int C>
int launch_add_rho(const Dims &d, const T *G_in, T rho, T *Gd, hipStream_t st);
template <typename T, int S, int C>
int launch_form_schur(const Dims &d, const T *Gd, const T *Cd, const T *g, const T *c, T *Sbd, T *Pbd,
                      T *gamma, T *Ginv, bool have_inverses, hipStream_t st);
// Fused assembly (A1 + A2 + A3) in one launch: see assemble_kernel in gato_assembly.hip.
struct AsmArgs {
    int mode;                        // 0: CSR in, G_dense/C_dense out; 1: G_dense/C_dense in; 2: G blocks in (+rho) -> G_dense, C_dense in
    const int *G_row, *G_col, *C_row, *C_col;
    const void *G_val, *C_val;       // mode 2: G_val = the caller's G blocks
    double rho;
    const void *g, *c;
    void *Gd, *Cd, *Ginv, *Sbd, *Pbd, *gamma;
    void *imgS, *imgP;               // optional (one system): S and Pinv also as column-major images over ALL rows (PcgLaunch::imgS), leading dimension img_ld
    int img_ld;
    unsigned long long *stamps;      // diagnostic (option stamp_asm): one workgroup's phase boundaries, 100 MHz ticks
};
template <typename T, int S, int C>
int launch_assemble(const Dims &d, const AsmArgs &a, hipStream_t st);

template <typename T, int S, int C>
int launch_form_ss(const Dims &d, const T *Sbd, T *Pbd, hipStream_t st);
template <typename T, int S, int C>
int launch_point_jacobi(const Dims &d, const T *Sbd, T *Pbd, hipStream_t st);
template <typename T, int S, int C>
int launch_compute_dz(const Dims &d, const T *Ginv, const T *Cd, const T *g, const T *lambda, T *dz,
                      hipStream_t st);
template <typename T, int S>
int pcg_resident_plan(PcgPlan *plan);
template <typename T, int S>
int launch_pcg_resident(const PcgLaunch &a, hipStream_t st);
template <typename T, int S>
int launch_pcg_resident_dpp(const PcgLaunch &a, hipStream_t st);      // gato_pcg_resident_dpp.hip; called by launch_pcg_resident
template <typename T, int S>
int launch_pcg_single(const PcgLaunch &a, bool mr, hipStream_t st);   // gato_pcg_resident_single.hip; called by launch_pcg_resident
template <typename T, int S> int pcg_dma_max_knots();          // knots per workgroup of the LDS-DMA variant (0: none for this shape)
template <typename T, int S>
int launch_pcg_dma(const PcgLaunch &a, hipStream_t st);
template <typename T, int S> int pcg_cg1_max_threads();
template <typename T, int S>
int launch_pcg_cg1(const PcgLaunch &a, hipStream_t st);

// Streaming PCG (two launches per iteration), gato_pcg_stream.hip
struct PcgStreamWork {
    void *vecs;                    // 6 consecutive S*K vectors: r[2], p[2], upsilon, r~
    void *partials;                // [4][max_groups] of T: eta' ring of 3 + v
    void *scalars;                 // final eta (double)
    int *done;                     // device flag
    int max_groups;
    int warm_start;                // lambda holds an initial guess on entry
    double *eta_hist;              // optional
};
// One launch of the streaming PCG on a shard of block rows (see gato_pcg_stream.hip).
struct StreamStep {
    const void *M;          // S_bd (phase A) or Pinv_bd (phase B), local block rows
    const void *a_old;      // p_old (A) / r_old (B) / gamma (init)
    const void *b;          // r~ (A) / upsilon (B)
    void *a_new;            // p_new (A) / r_new (B)
    void *y;                // upsilon (A) / r~ (B)
    void *lam;              // B and init
    const void *p_cur;      // B only: p of this iteration
    const void *part_num;   // coefficient = sum(num)/sum(den): partial arrays with element stride
    const void *part_den;
    int num_n, num_stride, den_n, den_stride;
    void *part_out;         // one slot per workgroup
    // ghosts of the neighbouring shards (multi-GPU): S-blocks of a_old and b, and where the advanced ghost goes
    const void *gh_a_left, *gh_a_right, *gh_b_left, *gh_b_right;
    void *gh_new_left, *gh_new_right;
    int first_global, last_global;
    int K;                  // local knots
    int it;
    int max_iters;
    double exit_tol;
    int *done;
    int *iters;
    double *eta_hist;       // optional history buffer (see PcgLaunch)
};
template <typename T, int S> int stream_grid(int K, int max_groups);
template <typename T, int S> int launch_stream_step(int phase, const StreamStep &a, int grid, hipStream_t st);
template <typename T, int S>
int launch_stream_pack(const void *slots, int nslots, const void *y, int K, void *send, hipStream_t st);
template <typename T, int S>
int launch_stream_finish(const void *part, int n, int stride, double exit_tol, int last_it, int *done, int *iters,
                         double *final_eta, double *eta_hist, hipStream_t st);

template <typename T, int S>
int launch_pcg_streaming(const Dims &d, const T *Sbd, const T *Pbd, const T *gamma, T *lambda,
                         T exit_tol, int max_iters, int *iters, const PcgStreamWork &w, hipStream_t st);

// Type-erased table used by the C ABI.
struct Ops {
    int S, C, dtype;
    int (*convert)(const Dims &, const int *, const int *, const void *, const int *, const int *,
                   const void *, double, void *, void *, void *, hipStream_t);
    int (*add_rho)(const Dims &, const void *, double, void *, hipStream_t);
    int (*form_schur)(const Dims &, const void *, const void *, const void *, const void *, void *, void *,
                      void *, void *, bool, hipStream_t);
    int (*form_ss)(const Dims &, const void *, void *, hipStream_t);
    int (*point_jacobi)(const Dims &, const void *, void *, hipStream_t);
    int (*assemble)(const Dims &, const AsmArgs &, hipStream_t);
    int (*compute_dz)(const Dims &, const void *, const void *, const void *, const void *, void *,
                      hipStream_t);
    int (*pcg_plan)(PcgPlan *);
    int (*pcg_resident)(const PcgLaunch &, hipStream_t);
    int (*pcg_dma_max_knots)();
    int (*pcg_dma)(const PcgLaunch &, hipStream_t);
    int (*pcg_cg1_max_threads)();
    int (*pcg_cg1)(const PcgLaunch &, hipStream_t);
    int (*pcg_streaming)(const Dims &, const void *, const void *, const void *, void *, double, int, int *,
                         const PcgStreamWork &, hipStream_t);
    int (*stream_grid)(int, int);
    int (*stream_step)(int, const StreamStep &, int, hipStream_t);
    int (*stream_pack)(const void *, int, const void *, int, void *, hipStream_t);
    int (*stream_finish)(const void *, int, int, double, int, int *, int *, double *, double *, hipStream_t);
};
const Ops *find_ops(int S, int C, int dtype);

}  // namespace gato
