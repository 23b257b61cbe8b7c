// dev tool: how fast can a persistent grid (one workgroup of W waves per CU) stream a large buffer from HBM
//   R: 16-byte loads into a register ring (depth D), consumed by an XOR
//   L: LDS-DMA (global_load_lds_dwordx4) into a per-wave LDS ring (depth D), read back with ds_read_b128
// hipcc -O3 --offload-arch=gfx950 tools/ubench_stream.hip -o build/ubench_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

static __device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// every wave walks its own contiguous 1 KiB pieces: piece index = (iteration * total_waves + global wave)
template <int D>
__global__ __launch_bounds__(1024) void k_reg(const uint4 *__restrict__ src, uint64_t n_pieces, uint32_t *out) {
    const int lane = threadIdx.x & 63;
    const uint64_t gw = (uint64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const uint64_t tw = (uint64_t)gridDim.x * (blockDim.x >> 6);
    uint4 ring[D];
    uint32_t acc = 0;
    uint64_t p = gw;
#pragma unroll
    for (int d = 0; d < D; ++d) ring[d] = src[(p + d * tw < n_pieces ? p + d * tw : gw) * 64 + lane];
    for (; p < n_pieces; p += D * tw) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const uint4 v = ring[d];
            const uint64_t q = p + (uint64_t)(D + d) * tw;
            ring[d] = src[(q < n_pieces ? q : gw) * 64 + lane];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int D>
__global__ __launch_bounds__(1024) void k_lds(const uint4 *__restrict__ src, uint64_t n_pieces, uint32_t *out) {
    __shared__ uint4 ring[D][16][64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t gw = (uint64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    const uint64_t tw = (uint64_t)gridDim.x * (blockDim.x >> 6);
    uint32_t acc = 0;
    uint64_t p = gw;
#pragma unroll
    for (int d = 0; d < D; ++d)
        glds16(src + (p + d * tw < n_pieces ? p + d * tw : gw) * 64 + lane, (uint32_t)(uintptr_t)&ring[d][wave][0]);
    for (; p < n_pieces; p += D * tw) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(D - 1) : "memory");
            const uint4 v = ring[d][wave][lane];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const uint64_t q = p + (uint64_t)(D + d) * tw;
            glds16(src + (q < n_pieces ? q : gw) * 64 + lane, (uint32_t)(uintptr_t)&ring[d][wave][0]);
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <typename F>
static void run(const char *name, F launch, uint64_t bytes) {
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
    printf("%-40s %.3f ms per pass, %.2f TB/s  (%s)\n", name, ms / 3, bytes / (ms / 3 * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
}

int main() {
    const uint64_t bytes = 32ull << 30;
    uint4 *src;
    uint32_t *out;
    (void)hipMalloc(&src, bytes);
    (void)hipMalloc(&out, 4 * 256 * 1024);
    (void)hipMemset(src, 1, bytes);
    const uint64_t n_pieces = bytes / 1024;
    for (int threads : {1024, 512}) {
        const dim3 g(256), blk(threads);
        printf("-- %d threads per workgroup, 256 workgroups, %llu GiB\n", threads, (unsigned long long)(bytes >> 30));
        run("registers, ring 4", [&] { hipLaunchKernelGGL(k_reg<4>, g, blk, 0, 0, src, n_pieces, out); }, bytes);
        run("registers, ring 8", [&] { hipLaunchKernelGGL(k_reg<8>, g, blk, 0, 0, src, n_pieces, out); }, bytes);
        run("LDS-DMA, ring 4", [&] { hipLaunchKernelGGL(k_lds<4>, g, blk, 0, 0, src, n_pieces, out); }, bytes);
        run("LDS-DMA, ring 8", [&] { hipLaunchKernelGGL(k_lds<8>, g, blk, 0, 0, src, n_pieces, out); }, bytes);
    }
    {
        const dim3 g(2048), blk(256);
        printf("-- 256 threads per workgroup, 2048 workgroups (8 per CU)\n");
        run("registers, ring 4", [&] { hipLaunchKernelGGL(k_reg<4>, g, blk, 0, 0, src, n_pieces, out); }, bytes);
        run("registers, ring 8", [&] { hipLaunchKernelGGL(k_reg<8>, g, blk, 0, 0, src, n_pieces, out); }, bytes);
    }
    return 0;
}
