// dev tool: can a strip kernel read every superblock TWICE -- once D steps ahead for the row tallies (from HBM), once
// for the accumulation (hoped to be served by the 256 MiB Infinity Cache) -- at the price of one read?
// 245 workgroups x 512 threads; workgroup w streams its own contiguous strip in 64 KiB superblocks (16 B per lane,
// 8 loads per thread and superblock).  D = 0: every superblock read once (baseline).  D > 0: at step k the
// workgroup loads superblock k + D (stream A) and superblock k again (stream B).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_mall.hip -o build/ubench_mall
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <bool NT>
static __device__ __forceinline__ v4u ld(const v4u *p) {
    return NT ? __builtin_nontemporal_load(p) : *p;
}

// NTA / NTB: non-temporal hint on the first / second read
template <bool TWICE, bool NTA, bool NTB>
__global__ __launch_bounds__(512, 2) void k_stream(const v4u *__restrict__ src, uint64_t sb_per_strip, uint32_t D,
                                                   uint32_t *out) {
    const v4u *base = src + (uint64_t)blockIdx.x * sb_per_strip * 4096 + threadIdx.x;  // 4096 x 16 B per superblock
    v4u a[8], b[8];
    uint32_t acc = 0, pop = 0;
    const uint32_t n = (uint32_t)sb_per_strip;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = ld<NTA>(base + (uint64_t)(D < n ? D : 0) * 4096 + i * 512);
        b[i] = TWICE ? ld<NTB>(base + i * 512) : v4u{0u, 0u, 0u, 0u};
    }
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t ka = k + 1 + D < n ? k + 1 + D : n - 1, kb = k + 1 < n ? k + 1 : n - 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const v4u va = a[i], vb = b[i];
            a[i] = ld<NTA>(base + (uint64_t)ka * 4096 + i * 512);
            if (TWICE) b[i] = ld<NTB>(base + (uint64_t)kb * 4096 + i * 512);
            pop += __popc(va.x) + __popc(va.y) + __popc(va.z) + __popc(va.w);
            acc ^= vb.x ^ vb.y ^ vb.z ^ vb.w;
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc + pop;
}

template <typename F>
static void run(const char *name, F launch, double bytes) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    launch();
    (void)hipEventRecord(a);
    for (int i = 0; i < 3; ++i) launch();
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    printf("%-58s %.3f ms per pass, %.2f TB/s algorithmic (%s)\n", name, ms / 3, bytes / (ms / 3 * 1e-3) / 1e12,
           hipGetErrorString(hipGetLastError()));
    fflush(stdout);
}

int main() {
    const uint32_t P = 245;
    const uint64_t sb_per_strip = 2048;  // 128 MiB per strip, 30.6 GiB in all
    const uint64_t bytes = (uint64_t)P * sb_per_strip * 65536;
    v4u *src;
    uint32_t *out;
    if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&out, 4 * 256 * 512) != hipSuccess) return 1;
    (void)hipMemset(src, 0x5a, bytes);
    const dim3 g(P), blk(512);
    printf("245 workgroups x 512 threads, %.1f GiB, superblock = 64 KiB per workgroup (15.3 MiB chip-wide per step)\n",
           bytes / 1073741824.0);
    run("read once", [&] { hipLaunchKernelGGL((k_stream<false, false, false>), g, blk, 0, 0, src, sb_per_strip, 0u, out); }, (double)bytes);
    run("read once, nt", [&] { hipLaunchKernelGGL((k_stream<false, true, false>), g, blk, 0, 0, src, sb_per_strip, 0u, out); }, (double)bytes);
    for (uint32_t D : {1u, 2u, 4u, 8u, 12u, 16u, 32u}) {
        char nm[128];
        snprintf(nm, sizeof nm, "read twice, D = %u (%.0f MiB apart), plain / plain", D, D * 15.3);
        run(nm, [&] { hipLaunchKernelGGL((k_stream<true, false, false>), g, blk, 0, 0, src, sb_per_strip, D, out); }, (double)bytes);
        snprintf(nm, sizeof nm, "read twice, D = %u, plain / nt", D);
        run(nm, [&] { hipLaunchKernelGGL((k_stream<true, false, true>), g, blk, 0, 0, src, sb_per_strip, D, out); }, (double)bytes);
    }
    return 0;
}
