// dev tool (round 6): what would a SPARSE correction pass for the missing genotypes of the multi-score
// kernel cost?  The pass that nps_multi.hip's is-missing matrix would be replaced by: per missing genotype (sample, row) add
// the row's S = 8 is-missing weights (49-bit integers) to that sample's S sums.  The only formulation without a divergent
// per-lane loop: a workgroup owns 1 024 samples, keeps their 8 x 1 024 int64 sums in LDS (64 KiB), walks the
// superblocks of its row chunk, stages the superblock's 128 x 8 weights in LDS (8 KiB) and lets every lane take ONE entry of
// the packed list of the tile's missing positions: 4 x ds_read_b128 (the row's weights) + 8 x 64-bit LDS atomic adds.
// This measures the upper bound of that core: the entries come from a hash (no list is read: at 1 % missing the list would
// be another 10 GB = 1.7 ms), 1 310 per (tile of 1 024 samples, superblock of 128 rows) = 1 % of 131 072 genotypes.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_sparse.hip -o build/ubench_sparse && build/ubench_sparse
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

static __device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}

template <int ENTRIES>
__global__ __launch_bounds__(1024) void k_sparse(const longlong2 *__restrict__ weights /* [n_sb][128][4] x 16 B */,
                                                 uint32_t n_sb, uint32_t sb_per_chunk, long long *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(smem);   // [1024][8]
    longlong2 *W = reinterpret_cast<longlong2 *>(smem + 65536);               // [2][128][4]
    const uint32_t t = threadIdx.x, tile = blockIdx.x, chunk = blockIdx.y;
    const uint32_t sb_a = chunk * sb_per_chunk, sb_b = min(n_sb, sb_a + sb_per_chunk);
#pragma unroll
    for (int s = 0; s < 8; ++s) acc[t * 8 + s] = 0ull;
    if (sb_a >= sb_b) return;
    if (t < 512) W[t] = weights[(uint64_t)sb_a * 512 + t];
    __syncthreads();
    for (uint32_t sb = sb_a; sb < sb_b; ++sb) {
        const int buf = (sb - sb_a) & 1;
        if (t < 512 && sb + 1 < sb_b) W[(buf ^ 1) * 512 + t] = weights[(uint64_t)(sb + 1) * 512 + t];
        for (uint32_t e = t; e < ENTRIES; e += 1024) {
            const uint32_t h = mix((tile * 7919u + sb) * 2053u + e);
            const uint32_t sample = h & 1023u, row = (h >> 10) & 127u;
            const longlong2 *w = W + buf * 512 + row * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const longlong2 v = w[q];
                atomicAdd(&acc[sample * 8 + 2 * q], (unsigned long long)v.x);
                atomicAdd(&acc[sample * 8 + 2 * q + 1], (unsigned long long)v.y);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int s = 0; s < 8; ++s) out[(((uint64_t)chunk * gridDim.x + tile) * 1024 + t) * 8 + s] = (long long)acc[t * 8 + s];
}

int main() {
    const uint32_t n_tiles = 489, n_sb = 7813, n_chunks = 12, per = (n_sb + n_chunks - 1) / n_chunks;
    longlong2 *d_w;
    long long *d_out;
    if (hipMalloc(&d_w, (size_t)n_sb * 512 * 16) != hipSuccess) return 1;
    if (hipMalloc(&d_out, (size_t)n_chunks * n_tiles * 1024 * 8 * 8) != hipSuccess) return 1;
    hipMemset(d_w, 1, (size_t)n_sb * 512 * 16);
    hipFuncSetAttribute((const void *)k_sparse<1310>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 16384);
    hipFuncSetAttribute((const void *)k_sparse<2620>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 16384);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
        for (int dens = 1; dens <= 2; ++dens) {
            hipEventRecord(a);
            if (dens == 1)
                hipLaunchKernelGGL(k_sparse<1310>, dim3(n_tiles, n_chunks), dim3(1024), 65536 + 16384, 0, d_w, n_sb, per, d_out);
            else
                hipLaunchKernelGGL(k_sparse<2620>, dim3(n_tiles, n_chunks), dim3(1024), 65536 + 16384, 0, d_w, n_sb, per, d_out);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            const double entries = (double)n_tiles * n_sb * (dens == 1 ? 1310 : 2620);
            printf("sparse core: %d %% missing, %.3g entries x 8 scores: %.2f ms (%.1f G entries/s); %s\n", dens, entries, ms,
                   entries / ms * 1e-6, hipGetErrorString(hipGetLastError()));
        }
    }
    return 0;
}
