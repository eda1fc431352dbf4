// Device helpers shared by the resident PCG kernels (gato_pcg_resident.hip, gato_pcg_cg1.hip).
#pragma once
#include "gato_common.h"

namespace gato {
namespace {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) int gi32;

template <typename T> struct VecOf;
template <> struct VecOf<float> { typedef float __attribute__((ext_vector_type(4))) type; static constexpr int W = 4; };
template <> struct VecOf<double> { typedef double __attribute__((ext_vector_type(2))) type; static constexpr int W = 2; };

__device__ __forceinline__ unsigned f2u(float x) { return __float_as_uint(x); }
__device__ __forceinline__ float u2f(unsigned x) { return __uint_as_float(x); }

template <typename T>
__device__ __forceinline__ T wave_sum(T v) { return wave_sum_dpp(v); }
// second level of a block sum: the per-wave partials (at most 16) sit in lanes 0..15, zeros elsewhere
template <typename T>
__device__ __forceinline__ T partials_sum(T v) { return row0_sum_dpp(v); }
// Block sum in two levels around ONE barrier, neither of which walks the whole wave: level 1 sums inside the four DPP rows
// of a wave (4 steps) and stores the four row sums (wp: 4 entries per wave, 32-byte aligned); after the barrier lane w
// reads the four of wave w with one vector read, adds them and the (at most 16) wave sums meet in row 0 (4 steps).  The
// two row-broadcast steps + readlane 63 that a full wave sum ends with cost more than all of this (14/7/50 fp64 on one
// CU: 2.37 -> 2.18 us per iteration with level 2 in row 0, -> see DESIGN.md 3.1 with level 1 in rows as well).
template <typename T>
__device__ __forceinline__ void partials_store(T *wp, int wave, int lane, T prod)
{
    const T rs = row_sums_dpp(prod);
    if ((lane & 15) == 0) wp[4 * wave + (lane >> 4)] = rs;
}
template <typename T, int MAXW = 16>
__device__ __forceinline__ T partials_total(const T *wp, int nwaves, int lane)
{
    typedef T __attribute__((ext_vector_type(4))) V4;
    T v = (T)0;
    if (lane < nwaves) {
        const V4 q = *reinterpret_cast<const V4 *>(wp + 4 * lane);
        v = (q[0] + q[1]) + (q[2] + q[3]);
    }
    return row0_sum_dpp<T, MAXW>(v);            // MAXW: compile-time bound of nwaves (8 waves: one DPP step less)
}

// Second level in EVERY lane, for workgroups of at most eight waves (round 4): lane l takes the four row sums of wave l & 7
// (slots of waves the launch does not have must hold zeros), so every group of eight lanes holds the eight wave sums and
// three DPP steps inside the group leave the total in all 64 lanes.  The same tree as partials_total<T, 8> - the same bits -
// without the exec mask around the read, the zero fill and the v_readlane at the end (the value is uniform but lives in a
// VGPR: a branch on it wants __builtin_amdgcn_readfirstlane).  14/7/50 fp64 on one CU: 1.593 -> 1.559 us per iteration.
template <typename T>
__device__ __forceinline__ T partials_total_all8(const T *wp, int lane)
{
    typedef T __attribute__((ext_vector_type(4))) V4;
    const V4 q = *reinterpret_cast<const V4 *>(wp + 4 * (lane & 7));
    T v = (q[0] + q[1]) + (q[2] + q[3]);
    v += dpp_mov<0xB1, 0xf, true>(v);     // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E, 0xf, true>(v);     // quad_perm [2,3,0,1]
    v += dpp_mov<0x141, 0xf, true>(v);    // row_half_mirror: the other quad of the group of eight
    return v;
}

// ---- granule transport -------------------------------------------------------------------
// SCOPE: __HIP_MEMORY_SCOPE_AGENT for hand-offs between the workgroups of one GPU (sc1 stores / loads),
// __HIP_MEMORY_SCOPE_SYSTEM for the cross-GPU mirrors of a cluster launch (sc0 sc1: through to memory / the fabric).
template <typename T, int SCOPE> struct GranuleT;
template <int SCOPE> struct GranuleT<float, SCOPE> {
    static constexpr int GPV = 1;
    __device__ static __forceinline__ void store(gu64 *g, unsigned ep, float v)
    {
        __hip_atomic_store(g, ((unsigned long long)ep << 32) | f2u(v), __ATOMIC_RELAXED, SCOPE);
    }
    __device__ static __forceinline__ float decode(const unsigned long long (&x)[1]) { return u2f((unsigned)x[0]); }
};
template <int SCOPE> struct GranuleT<double, SCOPE> {
    static constexpr int GPV = 2;
    __device__ static __forceinline__ void store(gu64 *g, unsigned ep, double v)
    {
        unsigned long long b = (unsigned long long)__double_as_longlong(v);
        __hip_atomic_store(g, ((unsigned long long)ep << 32) | (b & 0xffffffffull), __ATOMIC_RELAXED, SCOPE);
        __hip_atomic_store(g + 1, ((unsigned long long)ep << 32) | (b >> 32), __ATOMIC_RELAXED, SCOPE);
    }
    __device__ static __forceinline__ double decode(const unsigned long long (&x)[2])
    {
        return __longlong_as_double((long long)((x[1] << 32) | (x[0] & 0xffffffffull)));
    }
};
template <typename T> using Granule = GranuleT<T, __HIP_MEMORY_SCOPE_AGENT>;
// workgroup-scope stores (global_store ... sc0): the granule stays in the XCD's write-back L2 instead of being written through
// to memory.  Visible to agent-scope (sc1) loads of the SAME XCD only - 285 ns instead of 465 ns per one-way hand-off there
// (tools/micro/pingpong.hip: "st sc0 / ld sc1" 569 ns per round trip against 929), never across XCDs (stale reads): used by
// the one-XCD launches after they have CHECKED in the kernel that all their workgroups sit on one XCD.
template <typename T> using GranuleXcd = GranuleT<T, __HIP_MEMORY_SCOPE_WORKGROUP>;
template <typename T> using GranuleSys = GranuleT<T, __HIP_MEMORY_SCOPE_SYSTEM>;

constexpr int pad_to(int x, int m) { return (x + m - 1) / m * m; }

// ---- cluster launches (MR kernels), at exit: lambda_{k_end} for the dz back-substitution of the rank's last knot -------------
// dz_{x,k} = Q_k^-1 (q_k - lambda_k - A_k^T lambda_{k+1}) (gato_schur.cuh:833-852): the last knot of a rank needs the FIRST lambda
// block of the right neighbour rank.  Every rank but the first stores that block into its left neighbour's mirror (system-scope
// peer stores, {tag, payload} granules at a.lam_off, tag = a.lam_tag: unique per launch, identical on all ranks); the last
// workgroup of every rank but the last polls its OWN mirror for it (bounded) and writes it to dL[k_end] - the row behind the
// rank's own slice of the full-length lambda array - where the dz launch that follows on the stream finds it.  Call after the
// workgroup has stored its lambda rows, from every thread.  Returns false if the wait timed out (status word set).
template <typename T, int S>
__device__ __forceinline__ bool cluster_lambda_ghost(const PcgLaunch &a, int wg, int W, T *__restrict__ dL, bool aborted)
{
    typedef GranuleSys<T> XGr;
    constexpr int GPV = XGr::GPV;
    if (a.lam_tag == 0u || a.nranks <= 1) return true;
    const int tid = threadIdx.x;
    __syncthreads();                                    // the workgroup's lambda rows are stored (block 0 by lanes of any wave)
    bool ok = true;
    if (wg == 0 && a.rank > 0 && tid < S) {
        // (agent scope: read back from the L2 the stores above went through to - never a line this CU cached earlier, e.g. when
        //  lambda0 of a true warm start was loaded from the same array)
        const T v = __hip_atomic_load(dL + (size_t)a.k_begin * S + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        XGr::store((gu64 *)a.xpeer[a.rank - 1] + a.lam_off + tid * GPV, a.lam_tag, v);
    }
    if (wg == W - 1 && a.rank < a.nranks - 1 && tid < 64 && !aborted) {
        const int l = tid < S ? tid : 0;
        gu64 *src = (gu64 *)a.xslots + a.lam_off + l * GPV;
        unsigned long long raw[GPV];
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (unsigned spin = 0;; ++spin) {
            bool got = true;
#pragma unroll
            for (int g = 0; g < GPV; ++g) {
                raw[g] = __hip_atomic_load(src + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                got &= (unsigned)(raw[g] >> 32) == a.lam_tag;
            }
            if (__all(got)) break;
            if ((spin & 255u) == 255u && __builtin_amdgcn_s_memrealtime() - t0 > a.timeout_ticks) { ok = false; break; }
        }
        if (ok) {
            if (tid < S) dL[(size_t)a.k_end * S + tid] = XGr::decode(raw);
        } else if (tid == 0) __hip_atomic_store((gi32 *)a.status, a.launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return ok;
}

// LDS hand-over inside ONE wave (a lane reads what another lane of the same wave wrote): the wave's LDS operations execute in
// order, so all that is needed is that the compiler keeps them in order too.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// alpha = eta / v and beta = eta' / eta sit on the critical path of every iteration, in every lane; the IEEE division
// sequence (v_div_scale, v_rcp, 7 FMAs, v_div_fmas, v_div_fixup) costs 0.09 us each in fp64 on one CU.  Here: quotient
// of the MANTISSAS (both in [0.5, 1): no range problems whatever the operands) by hardware reciprocal + one Newton step
// (fp64 only) + the quotient's residual correction, then the exponents back in with ldexp - 8 (6) dependent
// instructions, branch-free.  Measured at 14/7/50 fp64 on one CU (2.37 us per iteration with IEEE divisions, 2.19 with
// the divisions replaced by products = the bound): this 2.34; the raw operands instead of the mantissas 2.29 but NaN
// once eta underflows; raw operands behind a range check with an IEEE fall-back 2.54 (even as a scalar branch); the
// parts that depend on eta alone hoisted a product ahead (three more live registers in a register-bound kernel) 2.46.
// Same bits as the IEEE quotient for every finite non-zero operand pair with a normal quotient (tools/micro/fastdiv.hip:
// 0 of 4 M random pairs per type differ, exponents over the whole range); 0 / d = 0; a zero or non-finite divisor gives
// NaN where IEEE may give an infinity (v = p.Sp = 0 means p = 0, i.e. eta = 0 and 0 / 0 either way).
__device__ __forceinline__ double quotient(double n, double d)
{
    const double mn = __builtin_amdgcn_frexp_mant(n), md = __builtin_amdgcn_frexp_mant(d);
    const int ex = __builtin_amdgcn_frexp_exp(n) - __builtin_amdgcn_frexp_exp(d);
    double x = __builtin_amdgcn_rcp(md);
    x = fma(x, fma(-md, x, 1.0), x);
    const double q = mn * x;
    return __builtin_amdgcn_ldexp(fma(fma(-md, q, mn), x, q), ex);
}
__device__ __forceinline__ float quotient(float n, float d)
{
    const float mn = __builtin_amdgcn_frexp_mantf(n), md = __builtin_amdgcn_frexp_mantf(d);
    const int ex = __builtin_amdgcn_frexp_expf(n) - __builtin_amdgcn_frexp_expf(d);
    const float x = __builtin_amdgcn_rcpf(md);
    const float q = mn * x;
    return __builtin_amdgcn_ldexpf(fmaf(fmaf(-md, q, mn), x, q), ex);
}

// y_row = [L M R]_row . window  - window read from LDS with 16-byte broadcast reads.
// fp64: a row's 3S products are added left to right, the reference's order (gato_utils.cuh:177-183).
// fp32 (GATO_PK_ROW, default on): packed FMAs - even and odd columns accumulate side by side in one v_pk_fma_f32 (the vector
// units issue a packed FP32 FMA at the rate of a plain one: half the FMA instructions of a product) and meet in one final
// add.  Another summation order than the reference's (columns 0,2,4,.. + columns 1,3,5,..); fp32 parity is measured against
// the fp64 oracle beside the reference order's own error (tests/f32_parity.py).
#ifndef GATO_PK_ROW
#define GATO_PK_ROW 1
#endif
template <typename T, int S, int SP>
__device__ __forceinline__ T row_times_window(const T (&m)[3 * S], const T *xw)
{
    typedef typename VecOf<T>::type V;
    constexpr int VW = VecOf<T>::W;
    constexpr int NV = SP / VW;
    if constexpr (sizeof(T) == 4 && GATO_PK_ROW && S % 2 == 0) {
        typedef float F2 __attribute__((ext_vector_type(2)));
        F2 acc = {0.f, 0.f};
#pragma unroll
        for (int b = 0; b < 3; ++b) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const V v = *reinterpret_cast<const V *>(xw + b * SP + i * VW);
#pragma unroll
                for (int e = 0; e < VW; e += 2) {
                    const int c = i * VW + e;                      // S even: a pair never straddles the end of a block
                    if (c < S) acc = __builtin_elementwise_fma(F2{m[b * S + c], m[b * S + c + 1]}, F2{v[e], v[e + 1]}, acc);
                }
            }
        }
        return acc[0] + acc[1];
    } else {
        T acc = (T)0;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                V v = *reinterpret_cast<const V *>(xw + b * SP + i * VW);
#pragma unroll
                for (int e = 0; e < VW; ++e) {
                    if (i * VW + e < S) acc = gato::fmaT(m[b * S + i * VW + e], v[e], acc);
                }
            }
        }
        return acc;
    }
}


// ---- row products with the operand window in REGISTERS (DPP-row layout, template parameter DR of pcg_resident_kernel) -----
// Layout: a knot owns whole 16-lane DPP rows (S <= 16: one row, lanes S..15 idle; 16 < S <= 32: two rows), so the entries of
// a knot's block of the operand vector sit in the lanes of that row and `v_fmac_f64_dpp / v_fmac_f32_dpp ... row_newbcast:c`
// (gfx90a+: lane c of each row feeds all 16 lanes of the row) reads them straight from the neighbouring lanes' registers: a
// product costs 2 scalar LDS reads per lane (the two neighbouring knots' entries of the lane's row index) instead of 3S/VW
// 16-byte reads, which is what bounds the LDS-window form (tools/micro/dppfma.hip: fp64 14-row knots 268 ns against 504 ns
// per product on 8 waves).  Same per-row summation order as row_times_window: fp64 left to right (the reference's,
// gato_utils.cuh:177-183), fp32 even columns + odd columns - bit-identical results.
// The chains are single asm statements: the compiler knows no DPP hazards inside inline asm (a VALU write of the DPP
// source needs 2 wait states, a VALU write of EXEC 5), so every statement starts with its own s_nop.
template <int S> struct DppRows {
    static constexpr bool ok = S == 12 || S == 14 || S == 16 || S == 32;      // block widths with a chain below
    static constexpr int lanes = S <= 16 ? 16 : 32;                            // lanes per knot
};
#define GATO_FM64(c, op) "v_fmac_f64_dpp %0, %1, %" #op " row_newbcast:" #c " row_mask:0xf bank_mask:0xf\n\t"
#define GATO_FM32(ac, c, op) "v_fmac_f32_dpp %" #ac ", %2, %" #op " row_newbcast:" #c " row_mask:0xf bank_mask:0xf\n\t"
#define GATO_FM64_12 GATO_FM64(0, 2) GATO_FM64(1, 3) GATO_FM64(2, 4) GATO_FM64(3, 5) GATO_FM64(4, 6) GATO_FM64(5, 7) \
    GATO_FM64(6, 8) GATO_FM64(7, 9) GATO_FM64(8, 10) GATO_FM64(9, 11) GATO_FM64(10, 12) GATO_FM64(11, 13)
#define GATO_FM32_12 GATO_FM32(0, 0, 3) GATO_FM32(1, 1, 4) GATO_FM32(0, 2, 5) GATO_FM32(1, 3, 6) GATO_FM32(0, 4, 7) GATO_FM32(1, 5, 8) \
    GATO_FM32(0, 6, 9) GATO_FM32(1, 7, 10) GATO_FM32(0, 8, 11) GATO_FM32(1, 9, 12) GATO_FM32(0, 10, 13) GATO_FM32(1, 11, 14)
#define GATO_M12(m) "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "v"(m[9]), "v"(m[10]), "v"(m[11])
// acc += sum_{c < N} m[c] * x(lane c of the lane's DPP row), c ascending
template <int N>
__device__ __forceinline__ void dpp_block(double &acc, double x, const double *m)
{
    static_assert(N == 12 || N == 14 || N == 16, "chain lengths built");
    if constexpr (N == 12) asm volatile("s_nop 4\n\t" GATO_FM64_12 : "+v"(acc) : "v"(x), GATO_M12(m));
    else if constexpr (N == 14)
        asm volatile("s_nop 4\n\t" GATO_FM64_12 GATO_FM64(12, 14) GATO_FM64(13, 15) : "+v"(acc) : "v"(x), GATO_M12(m), "v"(m[12]), "v"(m[13]));
    else
        asm volatile("s_nop 4\n\t" GATO_FM64_12 GATO_FM64(12, 14) GATO_FM64(13, 15) GATO_FM64(14, 16) GATO_FM64(15, 17)
                     : "+v"(acc) : "v"(x), GATO_M12(m), "v"(m[12]), "v"(m[13]), "v"(m[14]), "v"(m[15]));
}
// fp32: even columns into a0, odd columns into a1 (the order of the packed-FMA form)
template <int N>
__device__ __forceinline__ void dpp_block(float &a0, float &a1, float x, const float *m)
{
    static_assert(N == 12 || N == 14 || N == 16, "chain lengths built");
    if constexpr (N == 12) asm volatile("s_nop 4\n\t" GATO_FM32_12 : "+v"(a0), "+v"(a1) : "v"(x), GATO_M12(m));
    else if constexpr (N == 14)
        asm volatile("s_nop 4\n\t" GATO_FM32_12 GATO_FM32(0, 12, 15) GATO_FM32(1, 13, 16) : "+v"(a0), "+v"(a1) : "v"(x), GATO_M12(m), "v"(m[12]), "v"(m[13]));
    else
        asm volatile("s_nop 4\n\t" GATO_FM32_12 GATO_FM32(0, 12, 15) GATO_FM32(1, 13, 16) GATO_FM32(0, 14, 17) GATO_FM32(1, 15, 18)
                     : "+v"(a0), "+v"(a1) : "v"(x), GATO_M12(m), "v"(m[12]), "v"(m[13]), "v"(m[14]), "v"(m[15]));
}
// y_row = [L M R]_row . window, window block b given as x[b] (S <= 16: the entry of the lane's row index in knot j-1+b) or as
// x[2b], x[2b+1] (S = 32: entries (row & 15) and 16 + (row & 15) of that knot).  Call in uniform control flow (all lanes live).
template <typename T, int S>
__device__ __forceinline__ T row_times_dpp(const T (&m)[3 * S], const T (&x)[S <= 16 ? 3 : 6])
{
    static_assert(DppRows<S>::ok, "no DPP chain for this STATE_SIZE");
    if constexpr (sizeof(T) == 8) {
        T acc = (T)0;
        if constexpr (S <= 16) {
#pragma unroll
            for (int b = 0; b < 3; ++b) dpp_block<S>(acc, x[b], &m[b * S]);
        } else {
#pragma unroll
            for (int h = 0; h < 6; ++h) dpp_block<16>(acc, x[h], &m[h * 16]);
        }
        return acc;
    } else {
        T a0 = (T)0, a1 = (T)0;
        if constexpr (S <= 16) {
#pragma unroll
            for (int b = 0; b < 3; ++b) dpp_block<S>(a0, a1, x[b], &m[b * S]);
        } else {
#pragma unroll
            for (int h = 0; h < 6; ++h) dpp_block<16>(a0, a1, x[h], &m[h * 16]);
        }
        return a0 + a1;
    }
}

}  // namespace
}  // namespace gato
