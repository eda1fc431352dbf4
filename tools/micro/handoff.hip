// Micro-benchmark: what one all-to-all hand-off round of W workgroups costs, by exchange TOPOLOGY and POLL style - the
// design space of the PCG hand-off (gato_pcg_resident.hip: allreduce_and_halo) without any arithmetic.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/handoff.hip -o tools/micro/handoff && tools/micro/handoff
// Every round each workgroup publishes one 8-byte {epoch, payload} granule in a 128-byte line of its own (one writer per
// line) and may go on when it has seen the granules of all W workgroups - directly (flat) or as G group totals (grouped).
//   flat      : every workgroup's polling wave loads all W lines (what the resident kernel does today)
//   grouped   : workgroups with equal (index % G) form a group (blocks are dealt round-robin over the 8 XCDs, so a group
//               sits on one XCD - a placement hint, nothing depends on it); the group's first member polls the members'
//               lines, publishes the group total in a line of its own; every workgroup polls the G total lines.
//               W*G + W loads per sweep instead of W*W; chain = store, same-XCD poll, store, cross-XCD poll.
//   pipelined : two poll sweeps in flight (the second is issued before the first has returned)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef unsigned long long u64;

__device__ __forceinline__ void st_sc1(u64 *p, u64 v) { asm volatile("global_store_dwordx2 %0, %1, off sc1\n" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void ld_issue(u64 &v, u64 *p) { asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory"); }
#define WAIT_VM(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// poll until the granules this lane watches (n of them: p0, p1) carry epoch r; pipelined: two sweeps in flight
template <int NPTR, bool PIPE>
__device__ __forceinline__ bool poll(u64 *p0, u64 *p1, unsigned r, u64 t0)
{
    u64 a0 = 0, a1 = 0, b0 = 0, b1 = 0;
    if (!PIPE) {
        for (unsigned spin = 0;; ++spin) {
            ld_issue(a0, p0);
            if (NPTR > 1) ld_issue(a1, p1);
            WAIT_VM(0);
            bool ok = (unsigned)(a0 >> 32) == r;
            if (NPTR > 1) ok &= (unsigned)(a1 >> 32) == r;
            if (__all(ok)) return true;
            if ((spin & 1023u) == 1023u && __builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) return false;
        }
    } else {
        ld_issue(a0, p0);
        if (NPTR > 1) ld_issue(a1, p1);
        for (unsigned spin = 0;; ++spin) {
            ld_issue(b0, p0);
            if (NPTR > 1) ld_issue(b1, p1);
            if (NPTR > 1) WAIT_VM(2); else WAIT_VM(1);
            bool ok = (unsigned)(a0 >> 32) == r;
            if (NPTR > 1) ok &= (unsigned)(a1 >> 32) == r;
            if (__all(ok)) { WAIT_VM(0); return true; }
            ld_issue(a0, p0);
            if (NPTR > 1) ld_issue(a1, p1);
            if (NPTR > 1) WAIT_VM(2); else WAIT_VM(1);
            ok = (unsigned)(b0 >> 32) == r;
            if (NPTR > 1) ok &= (unsigned)(b1 >> 32) == r;
            if (__all(ok)) { WAIT_VM(0); return true; }
            if ((spin & 1023u) == 1023u && __builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) { WAIT_VM(0); return false; }
        }
    }
}

// lines: [parity][W] partial lines, then [parity][G] group-total lines; LINE = 16 granules (128 B); STRIDE granules between
// the partial lines of two workgroups (the PCG slot: 48 at 14/7 f32)
template <bool GROUPED, bool PIPE>
__global__ void exchange(u64 *slots, int W, int G, int stride, int pack, int rounds, int sleep_first, u64 *out)
{
    if (pack && (blockIdx.x & 7) != 0) return;
    const int wg = pack ? blockIdx.x >> 3 : blockIdx.x;
    if (wg >= W) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 t0 = __builtin_amdgcn_s_memrealtime();
    int ok = 1;
    const int g = wg % G, mem = wg / G;                 // group and member index
    const int M = (W - g + G - 1) / G;                  // members of this group
    for (int r = 1; r <= rounds; ++r) {
        u64 *part = slots + (size_t)(r & 1) * W * stride;
        u64 *tot = slots + (size_t)2 * W * stride + (size_t)(r & 1) * G * 16;
        if (wave == 0) {
            if (lane == 0) st_sc1(part + (size_t)wg * stride, ((u64)r << 32) | (unsigned)wg);
            if (!GROUPED) {
                u64 *p0 = part + (size_t)(lane < W ? lane : W - 1) * stride;
                u64 *p1 = part + (size_t)(lane + 64 < W ? lane + 64 : W - 1) * stride;
                if (sleep_first) __builtin_amdgcn_s_sleep(12);
                if (W > 64) ok = poll<2, PIPE>(p0, p1, (unsigned)r, t0);
                else ok = poll<1, PIPE>(p0, p1, (unsigned)r, t0);
            } else {
                if (mem == 0) {                          // the group's first member gathers the group
                    u64 *p0 = part + (size_t)(g + G * (lane < M ? lane : M - 1)) * stride;
                    ok = poll<1, PIPE>(p0, p0, (unsigned)r, t0);
                    if (lane == 0) st_sc1(tot + (size_t)g * 16, ((u64)r << 32) | (unsigned)g);
                } else if (sleep_first) __builtin_amdgcn_s_sleep(12);
                u64 *q0 = tot + (size_t)(lane < G ? lane : G - 1) * 16;
                ok &= poll<1, PIPE>(q0, q0, (unsigned)r, t0) ? 1 : 0;
            }
        }
        if (blockDim.x > 64) __syncthreads();
        if (!ok) break;
    }
    if (wg == 0 && threadIdx.x == 0) { out[0] = __builtin_amdgcn_s_memrealtime() - t0; out[1] = (u64)ok; }
}

template <bool GROUPED, bool PIPE>
double run(int W, int G, int pack, int threads, int sleep_first, int stride = 48)
{
    u64 *slots, *out;
    const size_t bytes = ((size_t)2 * W * stride + 2 * 16 * 64 + 64) * 8;
    hipMalloc(&slots, bytes); hipMemset(slots, 0, bytes);
    hipMalloc(&out, 64); hipMemset(out, 0, 64);
    const int rounds = 20000;
    hipLaunchKernelGGL((exchange<GROUPED, PIPE>), dim3(pack ? 8 * W : W), dim3(threads), 0, 0, slots, W, G, stride, pack, rounds, sleep_first, out);
    hipDeviceSynchronize();
    u64 h[2];
    hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    hipFree(slots); hipFree(out);
    return h[1] ? h[0] * 10.0 / rounds : -1.0;
}

int main()
{
    printf("ns per all-to-all round (20000 rounds; -1 = timed out).  threads: 64 = polling wave alone, 512 = + a workgroup barrier per round\n");
    printf("%-28s %8s %8s %8s %8s\n", "case", "flat", "flat+pp", "grouped", "grp+pp");
    struct Case { int W, pack, threads, sleep; };
    const Case cases[] = {{2, 1, 64, 0}, {7, 1, 64, 0}, {15, 1, 64, 0}, {15, 1, 512, 0}, {29, 1, 64, 0}, {29, 1, 512, 0},
                          {15, 0, 64, 0}, {32, 0, 64, 0}, {57, 0, 64, 0}, {57, 0, 64, 1}, {64, 0, 512, 1},
                          {114, 0, 64, 0}, {114, 0, 64, 1}, {114, 0, 512, 1}, {128, 0, 512, 1}};
    for (const Case &c : cases) {
        char name[64];
        snprintf(name, sizeof(name), "W=%3d %-7s %3d thr%s", c.W, c.pack ? "one XCD" : "spread", c.threads, c.sleep ? " sleep" : "");
        const int G = c.pack ? (c.W < 4 ? 1 : 4) : 8;     // one XCD: groups of ~W/4 just to see the shape; spread: one group per XCD
        printf("%-28s %8.0f %8.0f %8.0f %8.0f\n", name, run<false, false>(c.W, G, c.pack, c.threads, c.sleep),
               run<false, true>(c.W, G, c.pack, c.threads, c.sleep), run<true, false>(c.W, G, c.pack, c.threads, c.sleep),
               run<true, true>(c.W, G, c.pack, c.threads, c.sleep));
        fflush(stdout);
    }
    return 0;
}
