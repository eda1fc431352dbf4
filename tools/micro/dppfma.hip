// Micro-benchmark: a row product y = [L M R]_row . window with the window taken from REGISTERS of the lanes of the row's own
// 16-lane DPP row (v_fmac_f64_dpp / v_fmac_f32_dpp row_newbcast:c - lane c of each row feeds all 16 lanes; gfx90a+) against the
// same product with the window read from LDS (16-byte broadcast reads), as the resident PCG kernels do today.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/dppfma.hip -o /tmp/dppfma && /tmp/dppfma
// Layout of the DPP form: 16 lanes = one knot (S <= 16 rows, the rest idle), 4 knots per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

constexpr int S = 14;

template <int C> __device__ __forceinline__ void fmac_bc(double &acc, double x, double m)
{
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(m), "n"(C));
}
template <int C> __device__ __forceinline__ void fmac_bc(float &acc, float x, float m)
{
    asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(m), "n"(C));
}
template <typename T, int C0, int C> struct Chain {
    __device__ static __forceinline__ void run(T &acc, T x, const T *m)
    {
        fmac_bc<C0>(acc, x, m[C0]);
        Chain<T, C0 + 1, C>::run(acc, x, m);
    }
};
template <typename T, int C> struct Chain<T, C, C> { __device__ static __forceinline__ void run(T &, T, const T *) {} };

// DPP form: rounds of (write own entry, barrier, read the two neighbours' entries, 3S fmacs)
template <typename T>
__global__ __launch_bounds__(512) void dpp_kernel(const T *mat, const T *x0, T *out, int rounds, unsigned long long *cyc)
{
    __shared__ T xs[2][(32 + 2) * 16];
    const int tid = threadIdx.x, knot = tid >> 4, row = tid & 15;
    const bool active = row < S;
    T m[3 * S];
    for (int c = 0; c < 3 * S; ++c) m[c] = active ? mat[((size_t)knot * 3 * S + c) * S + row] : (T)0;
    T x = active ? x0[knot * S + row] : (T)0;
    for (int i = tid; i < 2 * 34 * 16; i += 512) (&xs[0][0])[i] = (T)0;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    const unsigned long long c0 = clock64();
    for (int r = 0; r < rounds; ++r) {
        T *w = xs[r & 1];
        w[(knot + 1) * 16 + row] = x;
        __syncthreads();
        const T xl = w[knot * 16 + row], xr = w[(knot + 2) * 16 + row];
        T acc = (T)0;
        asm volatile("s_nop 4" ::: "memory");
        Chain<T, 0, S>::run(acc, xl, m);
        Chain<T, 0, S>::run(acc, x, m + S);
        Chain<T, 0, S>::run(acc, xr, m + 2 * S);
        x = active ? acc : (T)0;
    }
    const unsigned long long c1 = clock64();
    if (tid == 0) { cyc[0] = c1 - c0; cyc[1] = wall_clock64() - t0; }
    if (active) out[knot * S + row] = x;
}

// LDS form (today): lane = row of a knot, 14 lanes per knot, window of 3S entries read with 16-byte reads
template <typename T>
__global__ __launch_bounds__(512) void lds_kernel(const T *mat, const T *x0, T *out, int rounds, unsigned long long *cyc, int nk)
{
    constexpr int VW = 16 / sizeof(T), SP = (S + VW - 1) / VW * VW;
    typedef T V __attribute__((ext_vector_type(VW)));
    __shared__ __attribute__((aligned(16))) T xs[2][(40 + 2) * SP];
    const int tid = threadIdx.x, knot = tid / S, row = tid - knot * S;
    const bool active = knot < nk;
    T m[3 * S];
    for (int c = 0; c < 3 * S; ++c) m[c] = active ? mat[((size_t)knot * 3 * S + c) * S + row] : (T)0;
    T x = active ? x0[knot * S + row] : (T)0;
    for (int i = tid; i < 2 * 42 * SP; i += 512) (&xs[0][0])[i] = (T)0;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    const unsigned long long c0 = clock64();
    for (int r = 0; r < rounds; ++r) {
        T *w = xs[r & 1];
        if (active) w[(knot + 1) * SP + row] = x;
        __syncthreads();
        T acc = (T)0;
        const T *xw = w + (active ? knot : 0) * SP;
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int i = 0; i < SP / VW; ++i) {
                const V v = *reinterpret_cast<const V *>(xw + b * SP + i * VW);
#pragma unroll
                for (int e = 0; e < VW; ++e)
                    if (i * VW + e < S) acc = __builtin_fma(m[b * S + i * VW + e], v[e], acc);
            }
        x = active ? acc : (T)0;
    }
    const unsigned long long c1 = clock64();
    if (tid == 0) { cyc[0] = c1 - c0; cyc[1] = wall_clock64() - t0; }
    if (active) out[knot * S + row] = x;
}

template <typename T> void run(const char *name)
{
    const int NKD = 32, NKL = 512 / S;        // knots per workgroup: DPP form 32 (4 per wave), LDS form 36
    const int rounds = 2000;
    std::vector<T> mat((size_t)40 * 3 * S * S), x(40 * S);
    srand(1);
    for (auto &v : mat) v = (T)((rand() % 2001 - 1000) / 1000.0 * 0.15);
    for (auto &v : x) v = (T)((rand() % 2001 - 1000) / 1000.0);
    // reference for a few rounds (same left-to-right order), nk knots, zero beyond the ends
    auto ref = [&](int nk, int nr) {
        std::vector<T> cur(x.begin(), x.begin() + nk * S), nxt(nk * S);
        for (int r = 0; r < nr; ++r) {
            for (int k = 0; k < nk; ++k)
                for (int row = 0; row < S; ++row) {
                    T acc = 0;
                    for (int b = 0; b < 3; ++b)
                        for (int c = 0; c < S; ++c) {
                            const int kk = k + b - 1;
                            const T xv = (kk < 0 || kk >= nk) ? (T)0 : cur[kk * S + c];
                            acc = std::fma(mat[((size_t)k * 3 * S + b * S + c) * S + row], xv, acc);
                        }
                    nxt[k * S + row] = acc;
                }
            cur.swap(nxt);
        }
        return cur;
    };
    T *dm, *dx, *dout; unsigned long long *dc;
    hipMalloc(&dm, mat.size() * sizeof(T)); hipMalloc(&dx, x.size() * sizeof(T)); hipMalloc(&dout, x.size() * sizeof(T)); hipMalloc(&dc, 16);
    hipMemcpy(dm, mat.data(), mat.size() * sizeof(T), hipMemcpyHostToDevice);
    hipMemcpy(dx, x.data(), x.size() * sizeof(T), hipMemcpyHostToDevice);
    for (int form = 0; form < 2; ++form) {
        const int nk = form == 0 ? NKD : NKL;
        // correctness at 5 rounds
        if (form == 0) hipLaunchKernelGGL(dpp_kernel<T>, dim3(1), dim3(512), 0, 0, dm, dx, dout, 5, dc);
        else hipLaunchKernelGGL(lds_kernel<T>, dim3(1), dim3(512), 0, 0, dm, dx, dout, 5, dc, nk);
        hipDeviceSynchronize();
        std::vector<T> got(nk * S);
        hipMemcpy(got.data(), dout, got.size() * sizeof(T), hipMemcpyDeviceToHost);
        const std::vector<T> want = ref(nk, 5);
        int bad = 0;
        for (int i = 0; i < nk * S; ++i) bad += got[i] != want[i];
        if (form == 0) hipLaunchKernelGGL(dpp_kernel<T>, dim3(1), dim3(512), 0, 0, dm, dx, dout, rounds, dc);
        else hipLaunchKernelGGL(lds_kernel<T>, dim3(1), dim3(512), 0, 0, dm, dx, dout, rounds, dc, nk);
        hipDeviceSynchronize();
        unsigned long long h[2];
        hipMemcpy(h, dc, 16, hipMemcpyDeviceToHost);
        printf("%-8s %-22s %d knots per workgroup: %4d of %d entries differ from the host's left-to-right fma chain; %.0f cycles, %.0f ns per product (8 waves)\n",
               name, form == 0 ? "DPP row_newbcast" : "LDS 16-byte reads", nk, bad, nk * S, (double)h[0] / rounds, h[1] * 10.0 / rounds);
    }
    hipFree(dm); hipFree(dx); hipFree(dout); hipFree(dc);
}

int main()
{
    run<double>("fp64");
    run<float>("fp32");
    return 0;
}
