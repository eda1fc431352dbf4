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
template <typename T> using GranuleSys = GranuleT<T, __HIP_MEMORY_SCOPE_SYSTEM>;

constexpr int pad_to(int x, int m) { return (x + m - 1) / m * m; }

// y_row = [L M R]_row . window  - window read from LDS with 16-byte broadcast reads.
template <typename T, int S, int SP>
__device__ __forceinline__ T row_times_window(const T (&m)[3 * S], const T *xw)
{
    typedef typename VecOf<T>::type V;
    constexpr int VW = VecOf<T>::W;
    constexpr int NV = SP / VW;
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

}  // namespace
}  // namespace gato
