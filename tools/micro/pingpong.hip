// Micro-benchmark: round-trip time of an 8-byte {epoch,payload} hand-off between two workgroups, by cache-control
// variant of the store / poll, for workgroups on the SAME XCD (blocks 0 and 8) and on different XCDs (blocks 0 and 1).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/pingpong.hip -o /tmp/pingpong && /tmp/pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned long long u64;

// V = 10 * store variant + load variant
template <int V> __device__ __forceinline__ void st(u64 *p, u64 v)
{
    constexpr int SV = V / 10;
    if (SV == 0) asm volatile("global_store_dwordx2 %0, %1, off sc1\n" ::"v"(p), "v"(v) : "memory");            // agent scope (today)
    else if (SV == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0\n" ::"v"(p), "v"(v) : "memory");
    else if (SV == 2) asm volatile("global_store_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
    else if (SV == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)\n\tbuffer_wbl2 sc0\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
}
template <int V> __device__ __forceinline__ u64 ld(u64 *p)
{
    constexpr int LV = V % 10;
    u64 v;
    if (LV == 0) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (LV == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (LV == 2) asm volatile("buffer_inv sc0\n\tglobal_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (LV == 3) asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (LV == 4) asm volatile("global_load_dwordx2 %0, %1, off sc0 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("buffer_inv sc1\n\tglobal_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int V>
__global__ void pingpong(u64 *slots, int partner_block, int rounds, u64 *out, int *xcc)
{
    const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == partner_block ? 1 : -1);
    if (me < 0 || threadIdx.x != 0) return;
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[me] = (int)(id & 0xf);
    u64 *mine = slots + me * 32, *theirs = slots + (1 - me) * 32;      // 256 B apart: separate lines
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    int ok = 1;
    for (int r = 1; r <= rounds && ok; ++r) {
        if (me == 0) st<V>(mine, ((u64)r << 32) | 1u);
        u64 v = 0;
        for (unsigned spin = 0;; ++spin) {
            v = ld<V>(theirs);
            if ((unsigned)(v >> 32) == (unsigned)r) break;
            if ((spin & 1023u) == 1023u && __builtin_amdgcn_s_memrealtime() - t0 > 20000000ull) { ok = 0; break; }   // 0.2 s
        }
        if (me == 1) st<V>(mine, ((u64)r << 32) | 2u);
    }
    out[me * 2] = __builtin_amdgcn_s_memrealtime() - t0;
    out[me * 2 + 1] = (u64)ok;
}

// memory kind of the slots: 0 = hipMalloc (default), 1 = uncached device memory, 2 = fine-grained (the kinds of the cluster mirrors)
static int g_mem_kind = 0;
static void alloc_slots(u64 **p, size_t bytes)
{
    if (g_mem_kind == 1) hipExtMallocWithFlags((void **)p, bytes, hipDeviceMallocUncached);
    else if (g_mem_kind == 2) hipExtMallocWithFlags((void **)p, bytes, hipDeviceMallocFinegrained);
    else hipMalloc(p, bytes);
}

template <int V> void run(const char *name, int partner)
{
    u64 *slots, *out; int *xcc;
    alloc_slots(&slots, 4096); hipMemset(slots, 0, 4096);
    hipMalloc(&out, 64); hipMemset(out, 0, 64);
    hipMalloc(&xcc, 16);
    const int rounds = 5000;
    hipLaunchKernelGGL(pingpong<V>, dim3(partner + 1), dim3(64), 0, 0, slots, partner, rounds, out, xcc);
    hipDeviceSynchronize();
    u64 h[4]; int hx[2];
    hipMemcpy(h, out, 32, hipMemcpyDeviceToHost); hipMemcpy(hx, xcc, 8, hipMemcpyDeviceToHost);
    printf("%-34s partner block %d (XCC %d / %d): %s, %.0f ns per round trip (= 2 one-way hand-offs)\n", name, partner, hx[0], hx[1],
           h[1] && h[3] ? "completed" : "TIMED OUT (stale reads)", h[0] * 10.0 / rounds);
    hipFree(slots); hipFree(out); hipFree(xcc);
}

// All-to-all: W workgroups (one polling wave each, optionally a workgroup barrier per round as in the PCG kernel), every
// round each publishes one granule and waits for all W - the floor of the PCG hand-off, without any arithmetic.
__global__ void allgather(u64 *slots, int W, int pack, int rounds, int with_barrier, u64 *out)
{
    if (pack && (blockIdx.x & 7) != 0) return;
    const int wg = pack ? blockIdx.x >> 3 : blockIdx.x;
    if (wg >= W) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    int ok = 1;
    for (int r = 1; r <= rounds; ++r) {
        u64 *base = slots + (size_t)(r & 1) * W * 32;
        if (wave == 0) {
            if (lane == 0) st<0>(base + wg * 32, ((u64)r << 32) | (unsigned)wg);
            u64 *p = base + (size_t)(lane < W ? lane : W - 1) * 32;
            for (unsigned spin = 0;; ++spin) {
                const u64 v = ld<0>(p);
                if (__all((unsigned)(v >> 32) == (unsigned)r)) break;
                if ((spin & 1023u) == 1023u && __builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) { ok = 0; break; }
            }
        }
        if (with_barrier) __syncthreads();
        if (!ok) break;
    }
    if (wg == 0 && threadIdx.x == 0) { out[0] = __builtin_amdgcn_s_memrealtime() - t0; out[1] = (u64)ok; }
}

void run_allgather(int W, int pack, int threads, int with_barrier)
{
    u64 *slots, *out;
    alloc_slots(&slots, 2 * 64 * 256 + 4096); hipMemset(slots, 0, 2 * 64 * 256 + 4096);
    hipMalloc(&out, 64); hipMemset(out, 0, 64);
    const int rounds = 20000;
    hipLaunchKernelGGL(allgather, dim3(pack ? 8 * W : W), dim3(threads), 0, 0, slots, W, pack, rounds, with_barrier, out);
    hipDeviceSynchronize();
    u64 h[2];
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("all-to-all W=%2d %-8s %3d threads %-12s: %s, %.0f ns per round\n", W, pack ? "one XCD" : "spread", threads,
           with_barrier ? "with barrier" : "no barrier", h[1] ? "completed" : "TIMED OUT", h[0] * 10.0 / rounds);
    hipFree(slots); hipFree(out);
}

template <int V> void both(const char *name) { run<V>(name, 8); run<V>(name, 1); }

int main(int argc, char **argv)
{
    if (const char *mk = getenv("PINGPONG_MEM")) { g_mem_kind = atoi(mk); printf("slots in memory kind %d (0 hipMalloc, 1 uncached, 2 fine-grained)\n", g_mem_kind); }
    if (argc > 1) {
        for (int W : {2, 4, 15, 29, 32})
            for (int pack : {1, 0}) {
                run_allgather(W, pack, 64, 0);
                run_allgather(W, pack, 512, 1);
            }
        for (int W : {57, 64}) { run_allgather(W, 0, 64, 0); run_allgather(W, 0, 512, 1); }
        return 0;
    }
    both<0>("st sc1 / ld sc1 (agent, today)");
    both<1>("st sc1 / ld sc0");
    both<2>("st sc1 / inv sc0 + ld");
    both<3>("st sc1 / ld nt");
    both<4>("st sc1 / ld sc0 nt");
    both<10>("st sc0 / ld sc1");
    both<20>("st plain+wait / ld sc1");
    both<40>("st plain+wbl2 sc0 / ld sc1");
    both<11>("st sc0 / ld sc0");
    both<12>("st sc0 / inv sc0 + ld");
    both<22>("st plain+wait / inv sc0 + ld");
    both<42>("st plain+wbl2 / inv sc0 + ld");
    both<30>("st sc0 sc1 / ld sc1");
    both<5>("st sc1 / inv sc1 + ld");
    return 0;
}
