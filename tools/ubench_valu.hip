// dev tool: issue cost of the instructions the fused kernel is made of, on gfx950.
//   hipcc -O3 --offload-arch=gfx950 -o ubench_valu tools/ubench_valu.hip && ./ubench_valu
// One 1024-thread workgroup per CU (4 waves per SIMD, like fused_cw_kernel); every wave runs ITERS x 32
// independent instructions of one kind between two s_memtime stamps.  Printed: shader cycles per
// wave-instruction per SIMD (elapsed / (ITERS*32*4)).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            return 1;                                                              \
        }                                                                          \
    } while (0)

constexpr int ITERS = 2000;

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define REP32(S) REP8(S) REP8(S) REP8(S) REP8(S)

template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned long long *out, unsigned int *sink, unsigned seed,
                                           const double *gtab) {
    const double *mytab = gtab + blockIdx.x * 256;  // this workgroup's 2 KiB table in global memory
    __shared__ double tab[2048 + 64];
    unsigned r[8];
    double d[8];
    for (int i = 0; i < 8; ++i) {
        r[i] = (threadIdx.x * 2654435761u + i * 40503u + seed) | 1u;
        d[i] = (double)r[i];
    }
    for (int i = threadIdx.x; i < 2048; i += 1024) tab[i] = i;
    __syncthreads();
    unsigned sm = 0x55555555u ^ seed;
    // random table offsets (bytes, 8-aligned) inside a 256-entry table
    unsigned idx[8];
    for (int i = 0; i < 8; ++i) {
        unsigned h = (threadIdx.x + 977u * i + seed) * 2246822519u;
        h ^= h >> 15;
        idx[i] = (OP == 21 || OP == 32) ? ((threadIdx.x & 31) * 8u + 256u * (i & 7)) : ((h >> 8) & 0xFFu) * 8u;
    }
    const unsigned long long cond = 0x5555555555555555ull ^ seed;
    if (OP == 16) asm volatile("s_mov_b64 vcc, %0" : : "s"(cond) : "vcc");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
        if (OP == 0) {
#define S(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(r[i]) : "s"(sm));
            REP32(S)
#undef S
        } else if (OP == 1) {
#define S(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r[i]) : "s"(sm), "v"(r[(i + 1) & 7]));
            REP32(S)
#undef S
        } else if (OP == 2) {
#define S(i) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(r[i]) : "v"(3u));
            REP32(S)
#undef S
        } else if (OP == 3) {
#define S(i) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
            REP32(S)
#undef S
        } else if (OP == 4) {
#define S(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            REP32(S)
#undef S
        } else if (OP == 5) {
#define S(i) asm volatile("v_add_u32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
            REP32(S)
#undef S
        } else if (OP == 6) {
#define S(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
            REP32(S)
#undef S
        } else if (OP == 7) {
#define S(i) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
            REP32(S)
#undef S
        } else if (OP == 8) {
#define S(i) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(r[i]));
            REP32(S)
#undef S
        } else if (OP == 9) {
#define S(i) asm volatile("v_add_u32_dpp %0, %1, %0 row_shr:4 row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
            REP32(S)
#undef S
        } else if (OP == 10) {
#define S(i) asm volatile("v_alignbit_b32 %0, %0, %0, 4" : "+v"(r[i]));
            REP32(S)
#undef S
        } else if (OP == 11) {
#define S(i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            REP32(S)
#undef S
        } else if (OP == 12) {
#define S(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            REP32(S)
#undef S
        } else if (OP == 13) {
#define S(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 1) & 7]), "s"(0x07050301u));
            REP32(S)
#undef S
        } else if (OP == 14) {
#define S(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
            REP32(S)
#undef S
        } else if (OP == 15) {  // select with the condition in an SGPR pair
#define S(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(r[(i + 1) & 7]), "s"(cond));
            REP32(S)
#undef S
        } else if (OP == 16) {  // select with vcc written once before the loop
#define S(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
            REP32(S)
#undef S
        } else if (OP == 17) {  // DPP add writing half of the lanes (bank mask)
#define S(i) asm volatile("v_add_u32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0x5" : "+v"(r[i]) : "v"(r[(i + 1) & 7]));
            REP32(S)
#undef S
        } else if (OP == 18) {
#define S(i) asm volatile("s_nop 1");
            REP32(S)
#undef S
        } else if (OP == 19) {  // LDS atomic add, 16 distinct addresses per wave
#define S(i) asm volatile("ds_add_u32 %0, %1" : : "v"((threadIdx.x & 15u) * 4u + 16384u), "v"(r[i]));
            REP32(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (OP == 20 || OP == 21) {  // ds_read_b64: random table index / conflict-free
#define S(i) asm volatile("ds_read_b64 %0, %1" : "=v"(d[i]) : "v"(idx[i]));
            REP32(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (OP == 30) {  // half of the waves (by SIMD pairing) VALU only, the other half LDS only
            if ((threadIdx.x >> 8) & 1) {
#define S(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r[i]) : "s"(sm), "v"(r[(i + 1) & 7]));
                REP32(S)
#undef S
            } else {
#define S(i) asm volatile("ds_read_b64 %0, %1" : "=v"(d[i]) : "v"(idx[i]));
                REP8(S)
#undef S
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        } else if (OP == 31) {  // every wave: 8 lookups, then 32 VALU while they are in flight, then the wait
#define S(i) asm volatile("ds_read_b64 %0, %1" : "=v"(d[i]) : "v"(idx[i]));
            REP8(S)
#undef S
#define S(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r[i]) : "s"(sm), "v"(r[(i + 1) & 7]));
            REP32(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (OP == 32) {  // as 31 with conflict-free lookups
#define S(i) asm volatile("ds_read_b64 %0, %1" : "=v"(d[i]) : "v"(idx[i]));
            REP8(S)
#undef S
#define S(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(r[i]) : "s"(sm), "v"(r[(i + 1) & 7]));
            REP32(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (OP == 33) {  // 8 lookups alone (reference for 31)
#define S(i) asm volatile("ds_read_b64 %0, %1" : "=v"(d[i]) : "v"(idx[i]));
            REP8(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (OP == 40) {  // 8 random 8-byte gathers from the workgroup's 2 KiB global table (L1 hits)
#define S(i) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(d[i]) : "v"(idx[i]), "s"(mytab));
            REP8(S)
#undef S
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (OP == 41) {  // 6 LDS lookups + 2 global gathers per block of 8
#define S(i) asm volatile("ds_read_b64 %0, %1" : "=v"(d[i]) : "v"(idx[i]));
            S(0) S(1) S(2) S(4) S(5) S(6)
#undef S
#define S(i) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(d[i]) : "v"(idx[i]), "s"(mytab));
            S(3) S(7)
#undef S
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        } else if (OP == 42) {  // 6 LDS lookups alone (reference for 41)
#define S(i) asm volatile("ds_read_b64 %0, %1" : "=v"(d[i]) : "v"(idx[i]));
            S(0) S(1) S(2) S(4) S(5) S(6)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (OP == 43) {  // 2 gathers alone
#define S(i) asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(d[i]) : "v"(idx[i]), "s"(mytab));
            S(3) S(7)
#undef S
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (OP == 50) {  // is the front end (bytes fetched) or the VALU the limit?  v_bfi (8 B) + s_mov (4 B)
#define S(i) asm volatile("v_bfi_b32 %0, %1, %0, %2\n\ts_mov_b32 s40, s41" : "+v"(r[i]) : "s"(sm), "v"(r[(i + 1) & 7]) : "s40");
            REP32(S)
#undef S
        } else if (OP == 51) {  // v_bfi (8 B) + 2 x s_mov (8 B)
#define S(i) asm volatile("v_bfi_b32 %0, %1, %0, %2\n\ts_mov_b32 s40, s41\n\ts_mov_b32 s42, s43" : "+v"(r[i]) : "s"(sm), "v"(r[(i + 1) & 7]) : "s40", "s42");
            REP32(S)
#undef S
        } else if (OP == 52) {  // s_mov alone
#define S(i) asm volatile("s_mov_b32 s40, s41" ::: "s40");
            REP32(S)
#undef S
        } else if (OP == 53) {  // v_bfi + s_waitcnt (4 B, no-op wait)
#define S(i) asm volatile("v_bfi_b32 %0, %1, %0, %2\n\ts_waitcnt lgkmcnt(15)" : "+v"(r[i]) : "s"(sm), "v"(r[(i + 1) & 7]));
            REP32(S)
#undef S
        } else if (OP == 54) {  // v_bfi with a 32-bit literal (12 B)
#define S(i) asm volatile("v_and_b32 %0, 0x12345678, %0" : "+v"(r[i]));
            REP32(S)
#undef S
        } else if (OP == 22) {  // the fused mix: sdwa + ds_read_b64 + v_add_f64
#define S(i)                                                                                          \
    asm volatile("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD "  \
                 "src1_sel:BYTE_1"                                                                    \
                 : "=v"(r[i])                                                                         \
                 : "v"(3u), "v"(idx[i]));                                                             \
    asm volatile("ds_read_b64 %0, %1" : "=v"(d[i]) : "v"(idx[i]));
            REP8(S)
#undef S
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#define S(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
            REP8(S)
#undef S
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned acc = 0;
    for (int i = 0; i < 8; ++i) acc += r[i] + (unsigned)d[i] + idx[i];
    if (acc == 0x12345u) sink[0] = acc;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + threadIdx.x / 64] = t1 - t0;
}

static const double *g_tab = nullptr;
template <int OP>
static int run(const char *name, int per_iter, unsigned long long *d_out, unsigned *d_sink, int cus) {
    std::vector<unsigned long long> h(cus * 16);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<OP>, dim3(cus), dim3(1024), 0, 0, d_out, d_sink, 12345u + rep, g_tab);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
    double s = 0;
    for (auto v : h) s += (double)v;
    s /= h.size();
    printf("%-44s %7.2f cycles per wave-instruction per SIMD (4 waves/SIMD)\n", name,
           s / ((double)ITERS * per_iter * 4));
    return 0;
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    unsigned long long *d_out;
    unsigned *d_sink;
    CK(hipMalloc(&d_out, cus * 16 * 8));
    CK(hipMalloc(&d_sink, 64));
    double *d_tab;
    CK(hipMalloc(&d_tab, (size_t)cus * 2048));
    CK(hipMemset(d_tab, 0, (size_t)cus * 2048));
    g_tab = d_tab;
    printf("%s, %d CUs\n", p.name, cus);
    run<0>("v_and_b32 (VOP2, sgpr)", 32, d_out, d_sink, cus);
    run<1>("v_bfi_b32 (VOP3, sgpr mask)", 32, d_out, d_sink, cus);
    run<2>("v_lshlrev_b32_sdwa BYTE_1", 32, d_out, d_sink, cus);
    run<3>("v_bcnt_u32_b32", 32, d_out, d_sink, cus);
    run<4>("v_add_f64", 32, d_out, d_sink, cus);
    run<5>("v_add_u32_dpp quad_perm", 32, d_out, d_sink, cus);
    run<9>("v_add_u32_dpp row_shr:4", 32, d_out, d_sink, cus);
    run<6>("v_cndmask_b32 (vcc)", 32, d_out, d_sink, cus);
    run<7>("v_lshl_or_b32", 32, d_out, d_sink, cus);
    run<8>("v_lshrrev_b32 (VOP2)", 32, d_out, d_sink, cus);
    run<10>("v_alignbit_b32", 32, d_out, d_sink, cus);
    run<11>("v_lshl_add_u64", 32, d_out, d_sink, cus);
    run<12>("v_pk_add_f32", 32, d_out, d_sink, cus);
    run<13>("v_perm_b32", 32, d_out, d_sink, cus);
    run<14>("v_add_f32", 32, d_out, d_sink, cus);
    run<15>("v_cndmask_b32_e64 (sgpr pair)", 32, d_out, d_sink, cus);
    run<16>("v_cndmask_b32_e32 (vcc set before)", 32, d_out, d_sink, cus);
    run<17>("v_add_u32_dpp quad_perm bank_mask:0x5", 32, d_out, d_sink, cus);
    run<18>("s_nop 1", 32, d_out, d_sink, cus);
    run<19>("ds_add_u32 (16 addresses per wave)", 32, d_out, d_sink, cus);
    run<20>("ds_read_b64 random 256-entry table", 32, d_out, d_sink, cus);
    run<21>("ds_read_b64 conflict-free", 32, d_out, d_sink, cus);
    run<33>("8 random lookups + wait (per block of 8)", 1, d_out, d_sink, cus);
    run<31>("8 random lookups + 32 v_bfi + wait (per block)", 1, d_out, d_sink, cus);
    run<32>("8 conflict-free lookups + 32 v_bfi + wait (per block)", 1, d_out, d_sink, cus);
    run<30>("half the waves 32 v_bfi, half 8 random lookups (per block)", 1, d_out, d_sink, cus);
    run<52>("s_mov_b32 alone", 32, d_out, d_sink, cus);
    run<50>("v_bfi_b32 + s_mov_b32 (per pair)", 32, d_out, d_sink, cus);
    run<51>("v_bfi_b32 + 2 s_mov_b32 (per triple)", 32, d_out, d_sink, cus);
    run<53>("v_bfi_b32 + s_waitcnt (per pair)", 32, d_out, d_sink, cus);
    run<54>("v_and_b32_e32 with a literal (8 bytes)", 32, d_out, d_sink, cus);
    run<40>("8 random 8-byte global gathers, 2 KiB table (per block of 8)", 1, d_out, d_sink, cus);
    run<42>("6 random LDS lookups (per block)", 1, d_out, d_sink, cus);
    run<43>("2 global gathers (per block)", 1, d_out, d_sink, cus);
    run<41>("6 LDS lookups + 2 global gathers (per block)", 1, d_out, d_sink, cus);
    run<22>("mix: sdwa + ds_read_b64(random) + v_add_f64 (per triple)", 8, d_out, d_sink, cus);
    return 0;
}
