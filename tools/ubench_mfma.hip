// dev tool: how fast does a SIMD run "8 int8 MFMAs (4 accumulators) + V vector instructions" per step, with
// W waves per SIMD and everything in registers?  (the instruction mix of multi_mfma_kernel's word step,
// without LDS / global memory).     hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma.hip -o build/ubench_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// L = 1: the four B operands of a step come from LDS (ds_read_b128 at a rotating offset of a 64 KiB table
// of random bytes), as the kernel's digit fragments do; A is remade from x every step
template <int V, int L>
__global__ __launch_bounds__(1024) void k(int iters, unsigned seed, int *out, long long *cycles) {
    __shared__ uint4 tab[L ? 4096 : 1];
    if (L) {
        for (int i = threadIdx.x; i < 4096; i += blockDim.x) {
            unsigned z = (i + 1) * 2654435761u ^ seed;
            tab[i] = make_uint4(z, z * 3u, z * 5u, z * 7u);
        }
        __syncthreads();
    }
    v16i acc[4];
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0;
    unsigned x = seed + threadIdx.x * 2654435761u, y = x ^ 0x9e3779b9u;
    v4i A = {(int)(x & 0x03030303u), (int)((x >> 2) & 0x03030303u), (int)((x >> 4) & 0x03030303u), (int)((x >> 6) & 0x03030303u)};
    v4i B = {(int)y, (int)(y * 3u), (int)(y * 5u), (int)(y * 7u)};
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (L) {
            const int base = ((i * 4) & 63) * 64 + (threadIdx.x & 63);
            v4i A2 = {(int)(x & 0x03030303u), (int)((x >> 2) & 0x03030303u), (int)((x >> 4) & 0x03030303u), (int)((x >> 6) & 0x03030303u)};
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const uint4 bd = tab[base + (2 * t) * 64], bm = tab[base + (2 * t + 1) * 64];
                const v4i BD = {(int)bd.x, (int)bd.y, (int)bd.z, (int)bd.w}, BM = {(int)bm.x, (int)bm.y, (int)bm.z, (int)bm.w};
                acc[2 * t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, BD, acc[2 * t], 0, 0, 0);
                acc[2 * t + 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A2, BD, acc[2 * t + 1], 0, 0, 0);
                acc[2 * t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A2, BM, acc[2 * t], 0, 0, 0);
                acc[2 * t + 1] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, BM, acc[2 * t + 1], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                acc[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A, B, acc[a], 0, 0, 0);
                acc[a] = __builtin_amdgcn_mfma_i32_32x32x32_i8(B, A, acc[a], 0, 0, 0);
            }
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {  // dependent-free-ish VALU filler (bitop3 / lshl_or flavour)
            x = (x << 1 | 1) & (y + v);
            asm volatile("" : "+v"(x));
        }
        A[0] ^= (int)(x & 3);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int a = 0; a < 4; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (int)x;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int V, int L = 0>
void run(int threads, int iters) {
    int *out;
    long long *cyc;
    const int blocks = 256;
    hipMalloc(&out, sizeof(int) * blocks * threads);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    hipLaunchKernelGGL((k<V, L>), dim3(blocks), dim3(threads), 0, 0, iters / 10, 1u, out, cyc);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<V, L>), dim3(blocks), dim3(threads), 0, 0, iters, 1u, out, cyc);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    const double waves_per_simd = threads / 64 / 4.0;
    // s_memtime ticks at 100 MHz: use wall time and count per-SIMD work instead
    const double mfma_per_simd = (double)iters * 8 * waves_per_simd;
    printf("%sV=%3d VALU per 8 MFMA, %4d threads (%.0f waves/SIMD): %.3f ms, %.1f ns per MFMA per SIMD  (32 cycles = %.1f ns at 2.1 GHz)\n",
           L ? "B from LDS, " : "", V, threads, waves_per_simd, ms, ms * 1e6 / mfma_per_simd, 32 / 2.1);
    hipFree(out);
    hipFree(cyc);
}

int main() {
    const int iters = 20000;
    for (int threads : {256, 512, 1024}) {
        run<0>(threads, iters);
        run<16>(threads, iters);
        run<32>(threads, iters);
        run<48>(threads, iters);
        run<64>(threads, iters);
        run<96>(threads, iters);
    }
    for (int threads : {512, 1024}) {
        run<0, 1>(threads, iters);
        run<8, 1>(threads, iters);
        run<16, 1>(threads, iters);
        run<32, 1>(threads, iters);
    }
    return 0;
}
