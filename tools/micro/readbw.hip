// Micro-benchmark: what a read-only stream of a buffer far larger than the Infinity Cache sustains on this chip, by
// workgroup count and loads in flight - the practical ceiling under the PCG kernels' HBM-bound runs.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/readbw.hip -o tools/micro/readbw && tools/micro/readbw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ void reader(const f4 *__restrict__ src, size_t n4, float *out)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    f4 acc = {0, 0, 0, 0};
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

// each workgroup streams its OWN contiguous chunk (what a knot-range partition does) instead of the grid-stride sweep
template <int U>
__global__ void reader_blocked(const f4 *__restrict__ src, size_t n4, float *out)
{
    const size_t chunk = n4 / gridDim.x;
    const f4 *p = src + (size_t)blockIdx.x * chunk;
    f4 acc = {0, 0, 0, 0};
    for (size_t i = threadIdx.x; i + (U - 1) * blockDim.x < chunk; i += U * blockDim.x) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + i + u * blockDim.x);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int U> void run_blocked(const f4 *buf, size_t n4, float *out, int blocks, int threads)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(reader_blocked<U>, dim3(blocks), dim3(threads), 0, 0, buf, n4, out);
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(reader_blocked<U>, dim3(blocks), dim3(threads), 0, 0, buf, n4, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("BLOCKED partition: blocks %5d x %4d threads, %d x 16 B in flight per lane: %.2f TB/s\n", blocks, threads, U, 5.0 * n4 * 16 / (ms * 1e-3) / 1e12);
}

template <int U> void run(const f4 *buf, size_t n4, float *out, int blocks, int threads)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(reader<U>, dim3(blocks), dim3(threads), 0, 0, buf, n4, out);
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(reader<U>, dim3(blocks), dim3(threads), 0, 0, buf, n4, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("blocks %5d x %4d threads, %d x 16 B in flight per lane: %.2f TB/s\n", blocks, threads, U, 5.0 * n4 * 16 / (ms * 1e-3) / 1e12);
}

int main()
{
    const size_t bytes = (size_t)1200 << 20;          // 1.2 GB: the K = 131072 matrices
    f4 *buf; float *out;
    hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes); hipMalloc(&out, 64);
    const size_t n4 = bytes / 16;
    for (int blocks : {256, 512, 1024, 2048, 4096}) {
        run<1>(buf, n4, out, blocks, 512);
        run<4>(buf, n4, out, blocks, 512);
        run<8>(buf, n4, out, blocks, 256);
    }
    for (int blocks : {256, 512}) {
        run_blocked<4>(buf, n4, out, blocks, 512);
        run_blocked<8>(buf, n4, out, blocks, 256);
        run_blocked<2>(buf, n4, out, blocks, 1024);
    }
    return 0;
}
