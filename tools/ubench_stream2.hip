// dev tool (round 4): mx_tally_kernel streams the strip layout at 7.0 TB/s, the persistent strip kernel's data path at 6.1.
// Which of the differences is it?  occupancy (32 small waves per CU vs 8 big ones), the nt hint, or the access pattern
// (many short-lived workgroups reading 1 MiB each vs 245 long-lived ones reading 4 GB each)?
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_stream2.hip -o build/ubench_stream2
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <bool NT>
static __device__ __forceinline__ v4u ld(const v4u *p) {
    return NT ? __builtin_nontemporal_load(p) : *p;
}
static __device__ __forceinline__ uint32_t fold(v4u w) { return __popc(w.x) + __popc(w.y) + __popc(w.z) + __popc(w.w); }

// A: the tally kernel's shape.  grid (superblocks, groups of 16 strips), 256 threads; wave w reads units w, w+4, ..
template <bool NT>
__global__ __launch_bounds__(256) void k_tally(const v4u *__restrict__ src, uint64_t n_sb, uint32_t P, uint32_t *out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t acc = 0;
    const uint32_t s_end = min(P, blockIdx.y * 16 + 16);
    for (uint32_t strip = blockIdx.y * 16; strip < s_end; ++strip) {
        const v4u *base = src + ((uint64_t)strip * 64 * n_sb + (uint64_t)blockIdx.x * 64) * 64 + lane;
        for (uint32_t u = wave; u < 64; u += 4) acc += fold(ld<NT>(base + (uint64_t)u * 64));
    }
    if (acc == 0x12345678u) out[0] = acc;
}

// B: short-lived workgroups on ONE strip each: grid (chunks of C superblocks, strips), 256 threads
template <bool NT, int C>
__global__ __launch_bounds__(256) void k_chunk(const v4u *__restrict__ src, uint64_t n_sb, uint32_t *out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t acc = 0;
    const v4u *base = src + ((uint64_t)blockIdx.y * 64 * n_sb + (uint64_t)blockIdx.x * C * 64) * 64 + lane;
    for (uint32_t u = wave; u < 64 * C; u += 4) acc += fold(ld<NT>(base + (uint64_t)u * 64));
    if (acc == 0x12345678u) out[0] = acc;
}

// C: persistent, one workgroup of T threads per strip, every wave keeps R loads in flight
template <bool NT, int R>
__global__ __launch_bounds__(1024) void k_persist(const v4u *__restrict__ src, uint64_t n_sb, uint32_t *out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const v4u *base = src + (uint64_t)blockIdx.x * 64 * n_sb * 64 + lane;
    const uint64_t n_units = 64 * n_sb;
    v4u ring[R];
    uint32_t acc = 0;
    uint64_t u = wave;
#pragma unroll
    for (int r = 0; r < R; ++r) ring[r] = ld<NT>(base + (u + (uint64_t)r * nw) * 64);
    for (; u + (uint64_t)R * nw < n_units; u += (uint64_t)R * nw) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const v4u v = ring[r];
            const uint64_t q = u + (uint64_t)(R + r) * nw;
            ring[r] = ld<NT>(base + (q < n_units ? q : wave) * 64);
            acc += fold(v);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
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
    printf("%-72s %.3f ms per pass, %.2f TB/s (%s)\n", name, ms / 3, bytes / (ms / 3 * 1e-3) / 1e12,
           hipGetErrorString(hipGetLastError()));
    fflush(stdout);
}

int main() {
    const uint32_t P = 245;
    const uint64_t n_sb = 2048;  // 128 MiB per strip, 30.6 GiB in all
    const uint64_t bytes = (uint64_t)P * n_sb * 65536;
    v4u *src;
    uint32_t *out;
    if (hipMalloc(&src, bytes) != hipSuccess || hipMalloc(&out, 4096) != hipSuccess) return 1;
    (void)hipMemset(src, 0x5a, bytes);
    printf("strip layout, 245 strips x 2048 superblocks x 64 KiB = %.1f GiB\n", bytes / 1073741824.0);
    run("A tally shape: grid (2048 sb, 16 groups of 16 strips) x 256 thr, nt", [&] { hipLaunchKernelGGL(k_tally<true>, dim3(n_sb, 16), dim3(256), 0, 0, src, n_sb, P, out); }, (double)bytes);
    run("A tally shape, plain loads", [&] { hipLaunchKernelGGL(k_tally<false>, dim3(n_sb, 16), dim3(256), 0, 0, src, n_sb, P, out); }, (double)bytes);
    run("B one strip per workgroup, 16 sb (1 MiB) each, 256 thr, nt", [&] { hipLaunchKernelGGL((k_chunk<true, 16>), dim3(n_sb / 16, P), dim3(256), 0, 0, src, n_sb, out); }, (double)bytes);
    run("B one strip per workgroup, 1 sb (64 KiB) each, 256 thr, nt", [&] { hipLaunchKernelGGL((k_chunk<true, 1>), dim3(n_sb, P), dim3(256), 0, 0, src, n_sb, out); }, (double)bytes);
    run("B one strip per workgroup, 128 sb (8 MiB) each, 256 thr, nt", [&] { hipLaunchKernelGGL((k_chunk<true, 128>), dim3(n_sb / 128, P), dim3(256), 0, 0, src, n_sb, out); }, (double)bytes);
    run("C persistent 245 x 512 thr, ring 9, nt", [&] { hipLaunchKernelGGL((k_persist<true, 9>), dim3(P), dim3(512), 0, 0, src, n_sb, out); }, (double)bytes);
    run("C persistent 245 x 512 thr, ring 9, plain", [&] { hipLaunchKernelGGL((k_persist<false, 9>), dim3(P), dim3(512), 0, 0, src, n_sb, out); }, (double)bytes);
    run("C persistent 245 x 512 thr, ring 18, nt", [&] { hipLaunchKernelGGL((k_persist<true, 18>), dim3(P), dim3(512), 0, 0, src, n_sb, out); }, (double)bytes);
    run("C persistent 245 x 1024 thr, ring 4, nt", [&] { hipLaunchKernelGGL((k_persist<true, 4>), dim3(P), dim3(1024), 0, 0, src, n_sb, out); }, (double)bytes);
    run("C persistent 245 x 1024 thr, ring 9, nt", [&] { hipLaunchKernelGGL((k_persist<true, 9>), dim3(P), dim3(1024), 0, 0, src, n_sb, out); }, (double)bytes);
    run("C persistent 245 x 1024 thr, ring 2, nt", [&] { hipLaunchKernelGGL((k_persist<true, 2>), dim3(P), dim3(1024), 0, 0, src, n_sb, out); }, (double)bytes);
    return 0;
}
