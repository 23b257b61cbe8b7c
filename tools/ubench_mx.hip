// dev tool (round 3): the building blocks of the matrix-core single-score kernel, probed on the device.
//   A  what ds_read_b64_tr_b4 delivers (a 16 x 16 transpose of 4-bit elements per 16-lane group)
//   B  v_mfma_scale_f32_16x16x128_f8f6f4 with FP4 (e2m1) A and FP8 (e4m3) / FP6 (e2m3) B: operand layout,
//      block scales, exact integer sums up to 2^24
//   C  its issue rate by B format, with and without vector instructions beside it
//   D  a prototype of the data path without the inter-workgroup hand-over: strip-major 1 KiB units ->
//      register ring -> popcount tally -> LDS image -> transposed reads -> FP4 operands -> MFMA
// hipcc -O3 --offload-arch=gfx950 tools/ubench_mx.hip -o exp/ubench_mx ;  exp/ubench_mx [n_sb]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define LDSP __attribute__((address_space(3)))
#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                   \
        }                                                                              \
    } while (0)

static __device__ __forceinline__ v2i tr4(const void *p) {
    return __builtin_amdgcn_ds_read_tr4_b64_v2i32((LDSP v2i *)p);
}

// ------------------------------------------------------------------------------------------- A
__global__ void probe_tr(const unsigned long long *in, unsigned long long *out) {
    __shared__ unsigned long long l[64];
    l[threadIdx.x] = in[threadIdx.x];
    __syncthreads();
    const v2i t = tr4(&l[threadIdx.x]);
    out[threadIdx.x] = (unsigned long long)(unsigned)t[0] | ((unsigned long long)(unsigned)t[1] << 32);
}

// ------------------------------------------------------------------------------------------- B
template <int BF>  // blgp: 0 = fp8 e4m3, 2 = fp6 e2m3, 4 = fp4
__global__ void probe_mfma(const int *a, const int *b, const float *c, const int *sa, const int *sb, float *d) {
    const int l = threadIdx.x;
    v8i A = {a[l * 4], a[l * 4 + 1], a[l * 4 + 2], a[l * 4 + 3], 0, 0, 0, 0};
    v8i B;
    for (int r = 0; r < 8; ++r) B[r] = b[l * 8 + r];
    v4f C = {c[l * 4], c[l * 4 + 1], c[l * 4 + 2], c[l * 4 + 3]};
    C = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, C, 4, BF, 0, sa[l], 0, sb[l]);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = C[r];
}

// ------------------------------------------------------------------------------------------- C
template <int BF, int V>
__global__ __launch_bounds__(512) void rate_mfma(int iters, unsigned seed, float *out) {
    v4f acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = v4f{0, 0, 0, 0};
    unsigned x = seed + threadIdx.x * 2654435761u, y = x ^ 0x9e3779b9u;
    v8i A = {(int)(x & 0x33333333u), (int)((x >> 2) & 0x33333333u), (int)(y & 0x33333333u), (int)((y >> 2) & 0x33333333u), 0, 0, 0, 0};
    v8i B = {(int)(y & 0x47474747u), (int)((y * 3u) & 0x47474747u), (int)((y * 5u) & 0x47474747u), (int)((y * 7u) & 0x47474747u),
             (int)((y * 9u) & 0x47474747u), (int)((y * 11u) & 0x47474747u), (int)((y * 13u) & 0x47474747u), (int)((y * 15u) & 0x47474747u)};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            acc[t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(A, B, acc[t], 4, BF, 0, 127, 0, 127);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                x = (x >> 1) & (y + v);
                asm volatile("" : "+v"(x));
            }
        }
        A[0] ^= (int)(x & 1);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)x;
}

// ------------------------------------------------------------------------------------------- D
// unit = 128 rows x 32 samples = 1 KiB, row-major [row][8 bytes]; nibble i of a row = samples 2i (bits 0,1) and
// 2i+1 (bits 2,3); codes 0,1,2 = dosage, 3 = missing.  cohort = [strip][superblock][64 units].
constexpr int UPW = 8;  // units per wave and superblock (8 waves x 8 units x 32 samples = 2048 samples per strip)
struct ProtoArgs {
    const v4u *data;
    uint32_t n_sb;
    const uint4 *btab;  // [n_sb][3 tables][2][64 lanes] uint4: MFMA B fragments (fp8) of the superblock's weights
    float *cout;        // [strip][wave][unit][2][64 lanes][4]
    unsigned *tout;     // [strip][n_sb][128 rows][3]
    int mode;           // bit 0: skip the tally, bit 1: skip the accumulation
};

static __host__ __device__ inline int rowoff(int r) {  // LDS image of a unit: where row r's 8 bytes live
    return 8 * (r & 15) + 128 * ((r >> 5) & 1) + 256 * ((r >> 4) & 1) + 512 * (r >> 6);
}

__global__ __launch_bounds__(512, 2) void proto(const ProtoArgs a) {
    extern __shared__ char smem[];  // [2 slots][64 units][1 KiB] + [2][3 tables][2][64] uint4 + tally scratch
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = lane >> 4, q = lane & 15;
    char *slot0 = smem + (size_t)(wave * UPW) * 1024;
    char *slot1 = smem + (size_t)(64 + wave * UPW) * 1024;
    uint4 *tabs = reinterpret_cast<uint4 *>(smem + 128 * 1024);               // [2][6][64]
    unsigned *tsc = reinterpret_cast<unsigned *>(smem + 128 * 1024 + 12288);  // [8 waves][128 rows][3]
    const int woff = rowoff(2 * lane);
    const int r1off = rowoff(32 * g + q), r2off = rowoff(32 * g + 16 + q);
    const v4u *base = a.data + ((size_t)blockIdx.x * a.n_sb * 64 + wave * UPW) * 64 + lane;
    v4f C[UPW][2];
#pragma unroll
    for (int u = 0; u < UPW; ++u) C[u][0] = C[u][1] = v4f{0, 0, 0, 0};
    v4u ring[2][UPW];

    auto load_sb = [&](uint32_t k, v4u(&dst)[UPW]) {
        if (k < a.n_sb) {
            const v4u *p = base + (size_t)k * 4096;
#pragma unroll
            for (int u = 0; u < UPW; ++u) dst[u] = __builtin_nontemporal_load(p + u * 64);
        }
    };
    // tally superblock k (rows 2*lane, 2*lane+1 of every unit of this wave) and park it in an LDS slot
    auto tally_park = [&](uint32_t k, const v4u(&src)[UPW], char *slot) {
        if (k >= a.n_sb) return;
        unsigned p1a = 0, pha = 0, p3a = 0, p1b = 0, phb = 0, p3b = 0;
#pragma unroll
        for (int u = 0; u < UPW; ++u) {
            const v4u w = src[u];
            if (!(a.mode & 1)) {
                const unsigned sx = w.x >> 1, sy = w.y >> 1, sz = w.z >> 1, sw = w.w >> 1;
                p1a += __popc(w.x) + __popc(w.y);
                pha += __popc((w.x & 0xAAAAAAAAu) | (sy & 0x55555555u));
                p3a += __popc((w.x & sx & 0x55555555u) | ((w.y & sy & 0x55555555u) << 1));
                p1b += __popc(w.z) + __popc(w.w);
                phb += __popc((w.z & 0xAAAAAAAAu) | (sw & 0x55555555u));
                p3b += __popc((w.z & sz & 0x55555555u) | ((w.w & sw & 0x55555555u) << 1));
            }
            *reinterpret_cast<v4u *>(slot + u * 1024 + woff) = w;
        }
        if (!(a.mode & 1)) {
            unsigned *t = tsc + (wave * 128 + 2 * lane) * 3;
            t[0] = p1a; t[1] = pha; t[2] = p3a; t[3] = p1b; t[4] = phb; t[5] = p3b;
        }
    };
    auto accumulate = [&](uint32_t k, const char *slot) {
        if (k >= a.n_sb || (a.mode & 2)) return;
        const uint4 *tb = tabs + (k & 1) * 384 + lane;
        v8i Bc, Bme, Bmo;
        {
            const uint4 x0 = tb[0], x1 = tb[64], y0 = tb[128], y1 = tb[192], z0 = tb[256], z1 = tb[320];
            Bc = v8i{(int)x0.x, (int)x0.y, (int)x0.z, (int)x0.w, (int)x1.x, (int)x1.y, (int)x1.z, (int)x1.w};
            Bme = v8i{(int)y0.x, (int)y0.y, (int)y0.z, (int)y0.w, (int)y1.x, (int)y1.y, (int)y1.z, (int)y1.w};
            Bmo = v8i{(int)z0.x, (int)z0.y, (int)z0.z, (int)z0.w, (int)z1.x, (int)z1.y, (int)z1.z, (int)z1.w};
        }
#pragma unroll
        for (int u = 0; u < UPW; ++u) {
            const v2i t01 = tr4(slot + u * 1024 + r1off), t23 = tr4(slot + u * 1024 + r2off);
            const unsigned w[4] = {(unsigned)t01[0], (unsigned)t01[1], (unsigned)t23[0], (unsigned)t23[1]};
            v8i ce = {0, 0, 0, 0, 0, 0, 0, 0}, co = ce, me = ce, mo = ce;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned s1 = w[r] >> 1;
                ce[r] = (int)(w[r] & 0x33333333u);
                co[r] = (int)(s1 & 0x66666666u);
                me[r] = (int)(w[r] & s1 & 0x11111111u);
                mo[r] = (int)(w[r] & s1 & 0x44444444u);
            }
            // block scales: even c x2 -> {0,1,2,3}; odd c (exponent form) -> {0,1,2,4}; missing bit -> 1
            C[u][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ce, Bc, C[u][0], 4, 0, 0, 128, 0, 127);
            C[u][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(co, Bc, C[u][1], 4, 0, 0, 127, 0, 127);
            C[u][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(me, Bme, C[u][0], 4, 0, 0, 128, 0, 127);
            C[u][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(mo, Bmo, C[u][1], 4, 0, 0, 126, 0, 127);
        }
    };
    // stand-in for the control work: wave 0 copies the tables of superblock k into table buffer k&1 and sums
    // the waves' tallies of superblock kt
    auto control = [&](uint32_t k, uint32_t kt) {
        if (wave != 0) return;
        if (k < a.n_sb) {
            const uint4 *src = a.btab + (size_t)k * 384 + lane;
            uint4 *dst = tabs + (k & 1) * 384 + lane;
#pragma unroll
            for (int i = 0; i < 6; ++i) dst[i * 64] = src[i * 64];
        }
        if (kt < a.n_sb && !(a.mode & 1)) {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int row = lane + 64 * rr;
                unsigned s0 = 0, s1 = 0, s2 = 0;
                for (int w8 = 0; w8 < 8; ++w8) {
                    const unsigned *t = tsc + (w8 * 128 + row) * 3;
                    s0 += t[0]; s1 += t[1]; s2 += t[2];
                }
                unsigned *o = a.tout + (((size_t)blockIdx.x * a.n_sb + kt) * 128 + row) * 3;
                o[0] = s0; o[1] = s1; o[2] = s2;
            }
        }
    };

    load_sb(0, ring[0]);
    load_sb(1, ring[1]);
    control(0, 0xffffffffu);
    tally_park(0, ring[0], slot0);
    __syncthreads();
    const uint32_t n_steps = (a.n_sb + 1) / 2 * 2;
    // step k: loads of superblock k+2 go into the register slot that superblock k occupied until it was parked
    auto step = [&](uint32_t k, v4u(&r_load)[UPW], const v4u(&r_tal)[UPW], char *s_tal, const char *s_acc) {
        load_sb(k + 2, r_load);
        control(k + 1, k);   // tables of k+1; tallies of k were parked before the last barrier
        accumulate(k, s_acc);
        __syncthreads();     // tally scratch of k consumed
        tally_park(k + 1, r_tal, s_tal);
        __syncthreads();
    };
    for (uint32_t k = 0; k < n_steps; k += 2) {
        step(k + 0, ring[0], ring[1], slot1, slot0);
        step(k + 1, ring[1], ring[0], slot0, slot1);
    }
    float *co = a.cout + ((size_t)(blockIdx.x * 8 + wave) * UPW * 2) * 256 + lane * 4;
#pragma unroll
    for (int u = 0; u < UPW; ++u)
#pragma unroll
        for (int eo = 0; eo < 2; ++eo) *reinterpret_cast<v4f *>(co + (u * 2 + eo) * 256) = C[u][eo];
}

__global__ void fill_random(uint4 *p, size_t n, unsigned seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        unsigned long long y = z * 0xD1342543DE82EF95ull + 1;
        y ^= y >> 29;
        p[i] = make_uint4((unsigned)z, (unsigned)(z >> 32), (unsigned)y, (unsigned)(y >> 32));
    }
}

// ---- host models ---------------------------------------------------------------------------
static float fp4val(int n) {
    const float m[8] = {0.f, .5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
    return (n & 8) ? -m[n & 7] : m[n & 7];
}
static float fp8val(int b) {  // e4m3fn
    const int e = (b >> 3) & 15, m = b & 7;
    const float v = e ? ldexpf(1.f + m / 8.f, e - 7) : ldexpf(m / 8.f, -6);
    return (b & 0x80) ? -v : v;
}
static float fp6val(int b) {  // e2m3
    const int e = (b >> 3) & 3, m = b & 7;
    const float v = e ? ldexpf(1.f + m / 8.f, e - 1) : m / 8.f;
    return (b & 0x20) ? -v : v;
}
static int fp8_of_int(int d) {  // |d| <= 16
    static const int enc[17] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E, 0x50, 0x51, 0x52, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58};
    return d < 0 ? (enc[-d] | 0x80) : enc[d];
}
static int fp6_of_int(int d) {  // |d| <= 7
    static const int enc[8] = {0, 8, 16, 20, 24, 26, 28, 30};
    return d < 0 ? (enc[-d] | 0x20) : enc[d];
}

static unsigned rng_state = 12345;
static unsigned rnd() {
    rng_state = rng_state * 1664525u + 1013904223u;
    return rng_state >> 8;
}

static int test_mfma(int bf) {
    std::vector<int> a(256), b(512, 0), sa(64), sb(64);
    std::vector<float> c(256), d(256), ref(256);
    int bad = 0;
    for (int trial = 0; trial < 4; ++trial) {
        // A nibble [i][k], B value [k][j]
        std::vector<float> Av(16 * 128), Bv(128 * 16);
        for (int l = 0; l < 64; ++l) {
            const int i = l & 15, gq = l >> 4;
            for (int r = 0; r < 4; ++r) {
                unsigned w = 0;
                for (int n = 0; n < 8; ++n) {
                    const int nib = trial == 3 ? (int)(rnd() & 15) : (int)(rnd() & 3);
                    w |= (unsigned)nib << (4 * n);
                    Av[i * 128 + 32 * gq + 8 * r + n] = fp4val(nib);
                }
                a[l * 4 + r] = (int)w;
            }
            // scales: per lane byte 0
            const int ea = trial >= 2 ? 125 + (int)(rnd() % 5) : 127, eb = trial >= 2 ? 126 + (int)(rnd() % 3) : 127;
            sa[l] = ea | 0x55000000;  // junk in the other bytes: only byte 0 may matter (op_sel 0)
            sb[l] = eb | 0x00330000;
            for (int n = 0; n < 32; ++n) Av[i * 128 + 32 * gq + n] *= ldexpf(1.f, ea - 127);
            const int j = l & 15;
            unsigned char bytes[32];
            memset(bytes, 0, sizeof bytes);
            for (int e = 0; e < 32; ++e) {
                const int dgt = (int)(rnd() % 15) - 7;
                float v;
                if (bf == 0) {
                    bytes[e] = (unsigned char)fp8_of_int(dgt);
                    v = fp8val(bytes[e]);
                } else {  // fp6: 32 x 6 bits packed little-endian
                    const int code = fp6_of_int(dgt);
                    const int bit = 6 * e;
                    for (int t = 0; t < 6; ++t)
                        if (code >> t & 1) bytes[(bit + t) >> 3] |= 1u << ((bit + t) & 7);
                    v = fp6val(code);
                }
                Bv[(32 * gq + e) * 16 + j] = v * ldexpf(1.f, eb - 127);
            }
            memcpy(&b[l * 8], bytes, 32);
        }
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) c[l * 4 + r] = trial == 1 ? (float)(8388608 - 20000 + (int)(rnd() % 1000)) : (float)((int)(rnd() % 2001) - 1000);
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const int i = 4 * (l >> 4) + r, j = l & 15;
                double s = c[l * 4 + r];
                for (int k = 0; k < 128; ++k) s += (double)Av[i * 128 + k] * Bv[k * 16 + j];
                ref[l * 4 + r] = (float)s;
            }
        int *da, *db, *dsa, *dsb;
        float *dc, *dd;
        CK(hipMalloc(&da, 1024)); CK(hipMalloc(&db, 2048)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256));
        CK(hipMalloc(&dc, 1024)); CK(hipMalloc(&dd, 1024));
        CK(hipMemcpy(da, a.data(), 1024, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice));
        CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
        CK(hipMemcpy(dc, c.data(), 1024, hipMemcpyHostToDevice));
        if (bf == 0)
            hipLaunchKernelGGL(probe_mfma<0>, dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb, dd);
        else
            hipLaunchKernelGGL(probe_mfma<2>, dim3(1), dim3(64), 0, 0, da, db, dc, dsa, dsb, dd);
        CK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
        int nb = 0;
        double maxd = 0;
        for (int x = 0; x < 256; ++x) {
            if (d[x] != ref[x]) ++nb;
            maxd = fmax(maxd, fabs((double)d[x] - ref[x]));
        }
        printf("B  mfma fp4 x %s trial %d (%s): %d of 256 differ, max |d| %.3g   e.g. got %.1f want %.1f\n", bf ? "fp6" : "fp8", trial,
               trial == 0 ? "codes 0..3, unit scales" : trial == 1 ? "C near 2^23" : trial == 2 ? "block scales" : "any fp4 nibble, block scales",
               nb, maxd, d[5], ref[5]);
        bad += nb;
        hipFree(da); hipFree(db); hipFree(dsa); hipFree(dsb); hipFree(dc); hipFree(dd);
    }
    return bad;
}

template <int BF, int V>
static void time_rate(const char *name, int threads) {
    float *out;
    CK(hipMalloc(&out, 256 * 512 * 4));
    const int iters = 4000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((rate_mfma<BF, V>), dim3(256), dim3(threads), 0, 0, 200, 1u, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((rate_mfma<BF, V>), dim3(256), dim3(threads), 0, 0, iters, 1u, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double per_simd = (double)iters * 8 * (threads / 256);
    printf("C  %-34s %d waves/SIMD: %.2f ns per MFMA and SIMD (%.1f cycles at 2.4 GHz)\n", name, threads / 256, ms * 1e6 / per_simd,
           ms * 1e6 / per_simd * 2.4);
    hipFree(out);
}

int main(int argc, char **argv) {
    const uint32_t n_sb_big = argc > 1 ? (uint32_t)atoi(argv[1]) : 2000;
    // ---- A
    {
        std::vector<unsigned long long> in(64), out(64);
        unsigned long long *di, *dout;
        CK(hipMalloc(&di, 512)); CK(hipMalloc(&dout, 512));
        for (int probe = 0; probe < 3; ++probe) {
            for (int l = 0; l < 64; ++l) {
                unsigned long long v = 0;
                for (int n = 0; n < 16; ++n) v |= (unsigned long long)(probe == 0 ? (l & 15) : probe == 1 ? n : (l >> 4)) << (4 * n);
                in[l] = v;
            }
            CK(hipMemcpy(di, in.data(), 512, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 0, 0, di, dout);
            CK(hipMemcpy(out.data(), dout, 512, hipMemcpyDeviceToHost));
            printf("A  tr_b4 probe %d (nibble = %s):", probe, probe == 0 ? "supplying lane & 15" : probe == 1 ? "position in the 8 bytes" : "lane group");
            for (int l : {0, 1, 5, 15, 16, 37, 63}) printf("  lane %d: %016llx", l, out[l]);
            printf("\n");
        }
        hipFree(di); hipFree(dout);
    }
    // ---- B
    int bad = test_mfma(0) + test_mfma(2);
    // ---- C
    time_rate<4, 0>("fp4 x fp4", 256);
    time_rate<2, 0>("fp4 x fp6", 256);
    time_rate<0, 0>("fp4 x fp8", 256);
    time_rate<2, 0>("fp4 x fp6", 512);
    time_rate<0, 0>("fp4 x fp8", 512);
    time_rate<2, 4>("fp4 x fp6 + 4 VALU per MFMA", 512);
    time_rate<0, 4>("fp4 x fp8 + 4 VALU per MFMA", 512);
    time_rate<2, 10>("fp4 x fp6 + 10 VALU per MFMA", 512);
    time_rate<0, 10>("fp4 x fp8 + 10 VALU per MFMA", 512);
    time_rate<0, 10>("fp4 x fp8 + 10 VALU per MFMA", 1024);

    // ---- D: small run checked on the host, then a big timed one
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int lds_bytes = 128 * 1024 + 12288 + 8 * 128 * 3 * 4;
    CK(hipFuncSetAttribute((const void *)proto, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    {
        const uint32_t n_sb = 7, strips = 3;
        const size_t n16 = (size_t)strips * n_sb * 64 * 64;
        std::vector<uint4> data(n16), bt((size_t)n_sb * 384);
        for (auto &x : data) x = make_uint4(rnd() ^ (rnd() << 12), rnd() ^ (rnd() << 12), rnd() ^ (rnd() << 12), rnd() ^ (rnd() << 12));
        std::vector<int> W((size_t)n_sb * 3 * 128 * 16);  // digit [sb][table][row][col]
        for (auto &x : W) x = (int)(rnd() % 16) - 8;
        for (uint32_t sb = 0; sb < n_sb; ++sb)
            for (int t = 0; t < 3; ++t)
                for (int l = 0; l < 64; ++l) {
                    unsigned char bytes[32];
                    for (int e = 0; e < 32; ++e) bytes[e] = (unsigned char)fp8_of_int(W[(((size_t)sb * 3 + t) * 128 + 32 * (l >> 4) + e) * 16 + (l & 15)]);
                    memcpy(&bt[(size_t)sb * 384 + (t * 2) * 64 + l], bytes, 16);
                    memcpy(&bt[(size_t)sb * 384 + (t * 2 + 1) * 64 + l], bytes + 16, 16);
                }
        uint4 *dd, *dbt;
        float *dco;
        unsigned *dto;
        CK(hipMalloc(&dd, n16 * 16)); CK(hipMalloc(&dbt, bt.size() * 16));
        CK(hipMalloc(&dco, (size_t)strips * 8 * UPW * 2 * 256 * 4)); CK(hipMalloc(&dto, (size_t)strips * n_sb * 128 * 3 * 4));
        CK(hipMemcpy(dd, data.data(), n16 * 16, hipMemcpyHostToDevice));
        CK(hipMemcpy(dbt, bt.data(), bt.size() * 16, hipMemcpyHostToDevice));
        ProtoArgs a{reinterpret_cast<const v4u *>(dd), n_sb, dbt, dco, dto, 0};
        hipLaunchKernelGGL(proto, dim3(strips), dim3(512), lds_bytes, 0, a);
        CK(hipDeviceSynchronize());
        std::vector<float> co((size_t)strips * 8 * UPW * 2 * 256);
        std::vector<unsigned> to((size_t)strips * n_sb * 128 * 3);
        CK(hipMemcpy(co.data(), dco, co.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(to.data(), dto, to.size() * 4, hipMemcpyDeviceToHost));
        size_t badc = 0, badt = 0;
        const unsigned char *bytes = reinterpret_cast<const unsigned char *>(data.data());
        auto code = [&](uint32_t p, uint32_t sb, int unit, int row, int s) {
            const unsigned char *r = bytes + ((((size_t)p * n_sb + sb) * 64 + unit) * 128 + row) * 8;
            return (r[s >> 2] >> (2 * (s & 3))) & 3;
        };
        for (uint32_t p = 0; p < strips; ++p) {
            for (uint32_t sb = 0; sb < n_sb; ++sb)
                for (int row = 0; row < 128; ++row) {
                    unsigned p1 = 0, ph = 0, p3 = 0;
                    for (int unit = 0; unit < 64; ++unit)
                        for (int s = 0; s < 32; ++s) {
                            const int c = code(p, sb, unit, row, s);
                            p1 += (c & 1) + (c >> 1);
                            ph += c >> 1;
                            p3 += c == 3;
                        }
                    const unsigned *t = &to[(((size_t)p * n_sb + sb) * 128 + row) * 3];
                    if (t[0] != p1 || t[1] != ph || t[2] != p3) ++badt;
                }
            for (int wv = 0; wv < 8; ++wv)
                for (int u = 0; u < UPW; ++u)
                    for (int eo = 0; eo < 2; ++eo)
                        for (int l = 0; l < 64; ++l)
                            for (int r = 0; r < 4; ++r) {
                                const int i = 4 * (l >> 4) + r, j = l & 15, s = 2 * i + eo, unit = wv * UPW + u;
                                double ex = 0;
                                for (uint32_t sb = 0; sb < n_sb; ++sb)
                                    for (int row = 0; row < 128; ++row) {
                                        const int c = code(p, sb, unit, row, s);
                                        const int *w0 = &W[(((size_t)sb * 3 + 0) * 128 + row) * 16], *wm = &W[(((size_t)sb * 3 + 1 + eo) * 128 + row) * 16];
                                        ex += (eo && c == 3 ? 4 : c) * w0[j] + (c == 3 ? wm[j] : 0);
                                    }
                                const float got = co[((((size_t)p * 8 + wv) * UPW + u) * 2 + eo) * 256 + l * 4 + r];
                                if (got != (float)ex) {
                                    if (badc < 5) printf("   C mismatch strip %u wave %d unit %d eo %d lane %d r %d: got %.1f want %.1f\n", p, wv, u, eo, l, r, got, ex);
                                    ++badc;
                                }
                            }
        }
        printf("D  small prototype run (%u strips x %u superblocks): %zu wrong tallies, %zu wrong sums\n", strips, n_sb, badt, badc);
        bad += (int)(badc + badt);
        hipFree(dd); hipFree(dbt); hipFree(dco); hipFree(dto);
    }
    {
        const uint32_t n_sb = n_sb_big, strips = 245;
        const size_t n16 = (size_t)strips * n_sb * 64 * 64;
        uint4 *dd, *dbt;
        float *dco;
        unsigned *dto;
        CK(hipMalloc(&dd, n16 * 16)); CK(hipMalloc(&dbt, (size_t)n_sb * 384 * 16));
        CK(hipMalloc(&dco, (size_t)strips * 8 * UPW * 2 * 256 * 4)); CK(hipMalloc(&dto, (size_t)strips * n_sb * 128 * 3 * 4));
        hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, dd, n16, 7u);
        hipLaunchKernelGGL(fill_random, dim3(256), dim3(256), 0, 0, dbt, (size_t)n_sb * 384, 9u);
        CK(hipDeviceSynchronize());
        const double bytes = (double)n16 * 16;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int mode : {0, 0, 1, 2, 3, 0}) {
            ProtoArgs a{reinterpret_cast<const v4u *>(dd), n_sb, dbt, dco, dto, mode};
            CK(hipEventRecord(e0));
            for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(proto, dim3(strips), dim3(512), lds_bytes, 0, a);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            ms /= 3;
            printf("D  prototype, %u strips x %u superblocks (%.1f GB), %s: %.3f ms per pass = %.2f TB/s -> %.2f ms for 125.04 GB\n", strips, n_sb, bytes / 1e9,
                   mode == 0 ? "tally + accumulate" : mode == 1 ? "accumulate only" : mode == 2 ? "tally only" : "load + park only", ms, bytes / ms / 1e9,
                   125.04e9 / (bytes / ms));
        }
        hipFree(dd); hipFree(dbt); hipFree(dco); hipFree(dto);
    }
    printf(bad ? "FAILED (%d)\n" : "ok\n", bad);
    return bad ? 1 : 0;
}
