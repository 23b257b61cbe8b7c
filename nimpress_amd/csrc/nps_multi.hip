// nps_multi.hip -- several score definitions evaluated in ONE pass over a resident cohort, on the
// matrix cores (reference loop nimpress.nim:634-641 run S times).
//
// For ONE score the inner loop is a weighted reduction: one table lookup per four genotypes, bound by
// the LDS (nps_fused.hip).  For S scores over the same rows it is a matrix product
//
//        scores[N x S]  =  dosage[N x M] . beta[M x S]   (+ imputed values of the missing genotypes)
//
// and that belongs on MFMA -- provided the arithmetic stays exact enough for the 1e-6 bar.  float64
// MFMA runs at the vector rate and would need every genotype converted to a double.  Instead:
//
//   * the per-row weights are turned into FIXED-POINT integers (49 bits, 41 on request; times 128 and
//     recombined for the operand extraction, see weight_digits) and split into ND = seven (six) signed base-256
//     digits: w = sum_k d_k 256^k / 2^(F+7).  code (int8) x digit (int8) accumulated in int32 by
//     v_mfma_i32_16x16x64_i8 is EXACT integer arithmetic, independent of the summation order; the
//     digit sums of a sample are recombined in float64 at the very end.  The only error is the
//     quantisation of the weights (2^-49 of the largest weight per row: ~1e-15 of a score; 2^-41 with six digits).
//   * a missing genotype has code 3 in this layout: it contributes 3 x beta through the dosage
//     matrix, and (imputed - 3) x beta through a second 0/1 matrix "is missing" (same accumulators).
//     A weight that is NaN in the reference (imp-sample fail / int_fail below --mincs, NaN eaf) sets a
//     flag digit instead: any sample that meets it comes out NaN, as in the reference.
//   * rows the reference imputes as a whole locus (over --maxmis, uncovered, absent, FILTER) add the
//     same constant to every sample: summed exactly in fixed point on the side, no matrix work.
//
// Data layout NPS_FMT_GT2M (the A operand wants 16 ROWS of one sample per lane): superblocks of 128
// rows x groups of 32 samples; unit (superblock, group) = 64 lanes x 16 bytes = 1 KiB contiguous, made for the
// 16 x 16 x 64 shape (16 samples x 64 rows per instruction; measured a fifth faster than 32 x 32 x 32 on this
// chip, tools/ubench_mfma_shape.hip): lane l = sample (l & 15) of either half of the group, row quarter
// g = l >> 4; word w = 0..3 belongs to the sample half w >> 1 (sample 16 (w >> 1) + (l & 15) of the group) and
// holds its rows 32 g + 16 (w & 1) + j (j = 0..15, code of row j in bits 2j, 2j+1); codes 0, 1, 2 = dosage,
// 3 = missing.
// Whole-row tallies (tallyAlleles, nimpress.nim:32-47) are produced when the cohort is packed (by the
// generator / the converter, as the streaming decode kernel does for pushed rows) and kept with it.
#include <algorithm>
#include <cmath>

#include "nps_kernels.h"

namespace nps {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------
// packing: synthetic generator, converter from the 2-bit row-major cohort layout, tallies
static __device__ __forceinline__ uint64_t mmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// NPS_CODE_* (0, 1, 3 = dosage 2, 2 = missing) -> layout code (0, 1, 2, 3 = missing)
static __device__ __forceinline__ uint32_t m_code(uint32_t c) { return c ^ (c >> 1); }

// one thread = one lane of one unit: two samples x 32 rows.  grid = (units chunks, superblocks)
__global__ __launch_bounds__(256) void synth_gt2m_kernel(uint4 *__restrict__ units, uint64_t n_groups,
                                                         uint64_t n_samples, uint64_t sb0,
                                                         uint64_t gen_row0, uint64_t n_rows, uint64_t seed,
                                                         const uint32_t *__restrict__ t_het,
                                                         const uint32_t *__restrict__ t_hom,
                                                         const uint32_t *__restrict__ t_miss) {
    const uint64_t u = (uint64_t)blockIdx.x * 256 + threadIdx.x;  // lane index inside the superblock
    const uint64_t sb = blockIdx.y;                                // relative to sb0
    if (u >= n_groups * 64) return;
    const uint64_t g = u >> 6;
    const uint32_t lane = (uint32_t)(u & 63);
    const uint32_t rq = lane >> 4;
    uint32_t out[4] = {0, 0, 0, 0};
    {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint64_t s = g * 32 + 16 * (w >> 1) + (lane & 15);
            uint32_t word = 0;
            for (int j = 0; j < 16; ++j) {
                const uint64_t r = sb * 128 + 32 * rq + 16 * (w & 1) + j;  // row relative to the first row written
                if (r < n_rows && s < n_samples) {
                    const uint64_t key = mmix64(seed ^ ((gen_row0 + r) * 0xD1B54A32D192ED03ull));
                    const uint64_t hsh = mmix64(key + s);
                    const uint32_t gq = (uint32_t)hsh, ms = (uint32_t)(hsh >> 32);
                    const uint32_t c = ms < t_miss[r] ? NPS_CODE_MISSING
                                                      : (gq < t_hom[r] ? NPS_CODE_DOSAGE2 : (gq < t_het[r] ? 1u : 0u));
                    word |= m_code(c) << (2 * j);
                }
            }
            out[w] = word;
        }
    }
    units[(sb0 + sb) * n_groups * 64 + u] = make_uint4(out[0], out[1], out[2], out[3]);
}

// whole-row tallies of generator rows, recomputed row-major (one thread = 16 samples of one row):
// tally[r] = nmissing << 32 | neffect.  grid = (sample chunks of 4096, rows)
__global__ __launch_bounds__(256) void synth_tally_kernel(unsigned long long *__restrict__ tally,
                                                          uint64_t n_samples, uint64_t gen_row0, uint64_t seed,
                                                          const uint32_t *__restrict__ t_het,
                                                          const uint32_t *__restrict__ t_hom,
                                                          const uint32_t *__restrict__ t_miss) {
    __shared__ uint32_t red[8];
    const uint64_t r = blockIdx.y;
    const uint64_t key = mmix64(seed ^ ((gen_row0 + r) * 0xD1B54A32D192ED03ull));
    const uint32_t th = t_het[r], tm = t_hom[r], tmi = t_miss[r];
    uint32_t nm = 0, ne = 0;
    const uint64_t s0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    for (int k = 0; k < 16; ++k) {
        const uint64_t s = s0 + k;
        if (s < n_samples) {
            const uint64_t hsh = mmix64(key + s);
            const uint32_t gq = (uint32_t)hsh, ms = (uint32_t)(hsh >> 32);
            if (ms < tmi)
                nm += 1;
            else
                ne += gq < tm ? 2u : (gq < th ? 1u : 0u);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        nm += __shfl_down(nm, o, 64);
        ne += __shfl_down(ne, o, 64);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[wv * 2] = nm;
        red[wv * 2 + 1] = ne;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long m = (unsigned long long)red[0] + red[2] + red[4] + red[6];
        const unsigned long long e = (unsigned long long)red[1] + red[3] + red[5] + red[7];
        if (m | e) atomicAdd(&tally[r], (m << 32) | e);
    }
}

// rows [0, n_rows) of a 2-bit row-major cohort (group-interleaved device layout of nps_kernels.h) ->
// units of superblocks [0, ceil(n_rows/128)).  A one-time repack.  Workgroup = one superblock x 32 word columns
// (512 samples = 16 groups): the 128 x 32 word tile comes in as 32 row groups x 512 contiguous bytes and goes
// through LDS; every thread takes the 16 rows x 16 samples of one word column and one block of 16 rows, turns the
// 16 words into plain 2-bit order and TRANSPOSES them in registers (four butterfly stages on 2-bit elements: 24
// vector ops per word; gathering the codes one by one took 112), and the 16 words it ends up with -- 16 rows of one
// sample each -- go back through LDS in unit order, so that the units leave as whole 16-byte lanes.
__global__ __launch_bounds__(256) void convert_gt2m_kernel(const uint4 *__restrict__ src /* first row group */,
                                                           uint64_t src_stride_words, uint64_t n_row_groups,
                                                           uint64_t n_words, uint4 *__restrict__ units,
                                                           uint64_t n_groups,
                                                           unsigned long long *__restrict__ tally /* zeroed */) {
    __shared__ uint32_t tile[128][33];       // [row][word column], padded
    __shared__ uint32_t outw[16][64 * 4 + 4];  // [group of the tile][lane x word], padded
    const int t = threadIdx.x;
    const uint64_t sb = blockIdx.y, c0 = (uint64_t)blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = t + 256 * i;  // (row group 0..31, column 0..31)
        const uint64_t rg = sb * 32 + (e >> 5), c = c0 + (e & 31);
        uint4 q = make_uint4(0, 0, 0, 0);
        if (rg < n_row_groups && c < n_words) q = src[rg * src_stride_words + c];
        const int r = 4 * (e >> 5);
        tile[r][e & 31] = q.x, tile[r + 1][e & 31] = q.y, tile[r + 2][e & 31] = q.z, tile[r + 3][e & 31] = q.w;
    }
    __syncthreads();
    {   // whole-row tallies (tallyAlleles, nimpress.nim:32-47) while the tile is here: two threads per row
        const int r = t >> 1;
        uint32_t cw = 0, cm = 0;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const uint32_t w = tile[r][16 * (t & 1) + c];
            cw += __popc(w);
            cm += __popc((w >> 4) & ~w & 0x0F0F0F0Fu);  // (nps_kernels.hip tally_word)
        }
        cw += __shfl_xor(cw, 1, 64);
        cm += __shfl_xor(cm, 1, 64);
        if ((t & 1) == 0 && (cw | cm))
            atomicAdd(&tally[sb * 128 + r], ((unsigned long long)cm << 32) | (unsigned long long)(cw - cm));
    }
    {
        const int cc = t & 31, rb = t >> 5;  // word column (16 samples), block of 16 rows
        uint32_t a[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            uint32_t x = word_from_planes(tile[16 * rb + j][cc]);  // sample s in bits 2s, 2s+1 (NPS_CODE_*)
            a[j] = x ^ ((x >> 1) & 0x55555555u);                   // -> layout codes (m_code on all 16 fields)
        }
        // a[j] = row j, element s = sample s  ->  a[s] = sample s, element j = row j
#define NPS_T2(d, mask)                                               \
    _Pragma("unroll") for (int j = 0; j < 16; ++j) if (!(j & d)) {    \
        const uint32_t u = ((a[j] >> (2 * d)) ^ a[j + d]) & mask;     \
        a[j + d] ^= u;                                                \
        a[j] ^= u << (2 * d);                                         \
    }
        NPS_T2(8, 0x0000FFFFu)
        NPS_T2(4, 0x00FF00FFu)
        NPS_T2(2, 0x0F0F0F0Fu)
        NPS_T2(1, 0x33333333u)
#undef NPS_T2
        // unit layout: lane = 16 (row quarter) + (sample & 15), word = 2 (sample half) + (row block & 1)
        uint32_t *o = &outw[cc >> 1][((rb >> 1) * 16) * 4 + 2 * (cc & 1) + (rb & 1)];
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) o[s2 * 4] = a[s2];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = t + 256 * i;  // (group 0..15, lane 0..63)
        const uint64_t g = (uint64_t)blockIdx.x * 16 + (e >> 6);
        if (g >= n_groups) continue;
        const uint32_t *o = &outw[e >> 6][(e & 63) * 4];
        units[(sb * n_groups + g) * 64 + (e & 63)] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

hipError_t launch_synth_gt2m(hipStream_t st, void *d_units, uint64_t n_samples, uint64_t row0, uint64_t gen_row0,
                             uint64_t n_rows, uint64_t seed, const uint32_t *d_t_het, const uint32_t *d_t_hom,
                             const uint32_t *d_t_miss, unsigned long long *d_tally /* [row0 ..) */) {
    if (n_rows == 0 || n_samples == 0) return hipSuccess;
    if (row0 % 128) return hipErrorInvalidValue;
    const uint64_t n_groups = (n_samples + 31) / 32, n_sb = (n_rows + 127) / 128;
    if (n_sb > 65535) return hipErrorInvalidValue;  // caller splits row ranges
    (void)hipGetLastError();
    hipLaunchKernelGGL(synth_gt2m_kernel, dim3((uint32_t)((n_groups * 64 + 255) / 256), (uint32_t)n_sb), dim3(256),
                       0, st, (uint4 *)d_units, n_groups, n_samples, row0 / 128, gen_row0, n_rows, seed, d_t_het,
                       d_t_hom, d_t_miss);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // whole 1 KiB units are written: the rows of the last superblock past n_rows are zero genotypes now, so their
    // tallies go back to zero as well (the tally array is padded to whole superblocks)
    e = hipMemsetAsync(d_tally + row0, 0, sizeof(unsigned long long) * n_sb * 128, st);
    if (e != hipSuccess) return e;
    for (uint64_t r = 0; r < n_rows; r += 65535) {
        const uint64_t k = std::min<uint64_t>(65535, n_rows - r);
        hipLaunchKernelGGL(synth_tally_kernel, dim3((uint32_t)((n_samples + 4095) / 4096), (uint32_t)k), dim3(256), 0,
                           st, d_tally + row0 + r, n_samples, gen_row0 + r, seed, d_t_het + r, d_t_hom + r,
                           d_t_miss + r);
    }
    return hipGetLastError();
}

hipError_t launch_convert_gt2m(hipStream_t st, const uint32_t *d_src, uint64_t src_stride_words,
                               uint64_t n_samples, uint64_t n_rows, void *d_units, unsigned long long *d_tally) {
    if (n_rows == 0 || n_samples == 0) return hipSuccess;
    const uint64_t n_groups = (n_samples + 31) / 32, n_sb = (n_rows + 127) / 128;
    const uint64_t n_words = (n_samples + 15) / 16, n_row_groups = (n_rows + 3) / 4;
    // (the tally array of a GT2M cohort is padded to whole superblocks: rows past n_rows read as zeros and add nothing)
    hipError_t e = hipMemsetAsync(d_tally, 0, sizeof(unsigned long long) * n_rows, st);
    if (e != hipSuccess) return e;
    (void)hipGetLastError();
    for (uint64_t sb = 0; sb < n_sb; sb += 65535) {
        const uint64_t k = std::min<uint64_t>(65535, n_sb - sb);
        hipLaunchKernelGGL(convert_gt2m_kernel, dim3((uint32_t)((n_words + 31) / 32), (uint32_t)k), dim3(256), 0, st,
                           reinterpret_cast<const uint4 *>(d_src) + sb * 32 * src_stride_words, src_stride_words,
                           n_row_groups - sb * 32, n_words, (uint4 *)d_units + sb * n_groups * 64, n_groups,
                           d_tally + sb * 128);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// per (position, score): the decision chain of getImputedDosages for this row (nimpress.nim:523-585)
// and the weights as signed base-256 digits in MFMA-fragment order.
//
// The product kernel never spreads 2-bit codes to bytes.  A byte of a unit word holds the codes of four rows
// (fields f = 0..3 at bits 2f) of one sample: byte = c0 + 4 c1 + 16 c2 + 64 c3.  Its PREFIXES
//      v0 = byte & 3,  v1 = byte & 15,  v2 = byte & 63,  v3 = byte          (one AND each, v3 none)
// are int8 operands as they stand, and the four codes are linear in them (c1 = (v1 - v0) / 4, ...), so
//      sum_f c_f W_f  =  v0 (W0 - W1/4) + v1 (W1/4 - W2/16) + v2 (W2/16 - W3/64) + v3 W3/64 .
// The change of basis goes into the weights: all weights are kept as V x 128 (|V| < 2^47, the combined
// coefficients 128 V0 - 32 V1, 32 V1 - 8 V2, 8 V2 - 2 V3, 2 V3 are exact integers below 2^55, which seven signed
// base-256 digits hold).  v3 can reach 255, int8 operands are signed: the kernel uses v3 - 128 (`word ^
// 0x80808080`, one op) and the missing 128 x 2 V3 = V3 x 2 (in units of 2^-F) is the same for every sample --
// it joins the whole-locus constants of the score, summed exactly in fixed point.
// The is-missing operand is the same construction on m = w & (w >> 1) & 0x55555555 (bit 2f of a byte: field f is
// code 3): prefixes m & 0x01.., m & 0x05.., m & 0x15.., m itself (at most 85, no sign problem); the NaN flag of a
// row rides along as a digit 64 x flag in the same basis.  10 VALU ops per 16 genotypes for both matrices
// (24 for spreading codes to bytes; 15 for masking the fields one by one, the first version of this kernel).
//
// One MFMA (16 samples x 64 rows) takes the prefixes f = 2p and 2p + 1 (pair p) of the two words a lane holds for
// one half of the group: K index k = 8 e + 4 i + b  <->  byte b of word i, prefix 2 p + e  (rows 32 g + 16 i + 4 b + ..).
//   table bytes: index(sb, p, t, dm, lane, k) = ((((sb*2 + p)*T + t)*2 + dm)*64 + lane)*16 + k   (T tiles of 16 columns)
//   column c = multi_col(S, ND, s, digit) ; t = c / 16 ; lane = c % 16 + 16 g
//
// Columns, most significant first: digit d of score s in column (ND - 1 - d) S + s, the NaN flags behind all
// digits in columns ND S + s; T = ceil((ND + 1) S / 16) tiles.  So
//   * the dosage matrix has no flags and needs the tiles below TD = ceil(ND S / 16) only; eight scores of six
//     digits: 3 tiles, their flags alone in the fourth; of seven digits: 4 tiles, the flags in the last;
//   * the is-missing matrix needs the same, plus the tiles from TF0 = floor(ND S / 16) on -- those holding flag
//     columns -- in calls in which a NaN imputation value occurs at all (MultiState.m_flag);
//   * a caller who accepts 32-bit is-missing weights (nps_multi_set_missing_weight_bits) needs the tiles below
//     ceil(4 S / 16) of that matrix: the coefficients are rounded to a multiple of 256^(ND-4), their low digits are zero.
static __host__ __device__ constexpr __forceinline__ int multi_col(int S, int ND, int s, int d /* 0..ND-1, ND = flag */) {
    return d < ND ? (ND - 1 - d) * S + s : ND * S + s;
}

// the signed base-256 digits of a coefficient (already scaled: see above; |V| < 2^(8 ND - 1))
static __device__ __forceinline__ void weight_digits(long long V, int ND, int keep, int (&d)[8]) {
    if (keep > 0 && keep < ND) {  // nearest multiple of 256^(ND-keep): only the `keep` leading digits remain
        const int sh = 8 * (ND - keep);
        V = ((V + (1ll << (sh - 1))) >> sh) << sh;
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        d[k] = (int)((V + 128) & 255) - 128;
        V = (V - d[k]) >> 8;
    }
    d[7] = 0;
}

struct MultiState {          // per score, on the device
    unsigned long long nloci;  // rows for which getImputedDosages returned true
    long long const_lo, const_hi;  // whole-locus constants of this call in fixed point: (hi << 32) + lo
    unsigned long long const_nan;  // a whole-locus constant was NaN (imp-locus fail / NaN eaf)
    double const_sum;          // float64 sum over the calls so far
    unsigned long long m_low;  // (score 0 only) m_flag: a NaN imputation value occurred in this call (flag tiles needed)
    double pad[2];
};

// One thread per (superblock, row quarter g, word i, score slot s): the 16 rows 128 sb + 32 g + 16 i + 4 b + f
// (b = byte, f = field) -- whole bytes, because the prefix basis couples the four fields of a byte.  It writes
// 32-bit pieces of the fragments: word 2 e + i of lane (column, g) in the tables of pair p holds prefix 2 p + e.
// blockIdx.y = score; one more (y = S) writes zeros to the columns behind the flags, rows past n_desc get zeros,
// so the table needs no memset.
__global__ __launch_bounds__(256) void multi_params_kernel(
    const unsigned long long *__restrict__ tally /* first cohort row of this call */,
    const nps_row_desc *__restrict__ desc /* [S][n_desc] */, uint64_t n_desc, int S, int ND, int T, uint64_t n_samples,
    DevParams p, const int *__restrict__ F /* [S] */, uint32_t *__restrict__ table, MultiState *__restrict__ state,
    int coarse_missing, uint32_t n_sb) {
    const uint64_t u = (uint64_t)blockIdx.x * 256 + threadIdx.x;  // (sb, g, i)
    const int s = blockIdx.y;
    if (u >= (uint64_t)n_sb * 8) return;
    const uint64_t sb = u >> 3;
    const int g = (int)((u >> 1) & 3), wi = (int)(u & 1);
    if (s == S) {  // the unused columns of the last tile
        for (int c = (ND + 1) * S; c < 16 * T; ++c)
            for (int dm = 0; dm < 2; ++dm)
                for (int f = 0; f < 4; ++f)
                    table[(((((sb * 2 + (f >> 1)) * T + (c >> 4)) * 2 + dm) * 64 + (c & 15) + 16 * g) << 2) + 2 * (f & 1) + wi] = 0;
        return;
    }
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    const int fx = F[s];
    uint32_t frag[2][8][4];  // [matrix][digit][prefix f]: byte b = the coefficient digit of (byte b, prefix f)
#pragma unroll
    for (int dm = 0; dm < 2; ++dm)
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int w = 0; w < 4; ++w) frag[dm][k][w] = 0;
    unsigned long long n_used = 0, c_lo = 0, c_hi = 0;
    bool c_nan = false, any_m_nan = false;
    auto add_const = [&](long long V) {  // exact, order-independent sums of both halves
        c_lo += (unsigned long long)(V & 0xffffffffll);
        c_hi += (unsigned long long)(V >> 32);
    };
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        long long VD[4] = {0, 0, 0, 0}, VM[4] = {0, 0, 0, 0};
        int flag[4] = {0, 0, 0, 0};
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            const uint64_t j = sb * 128 + 32 * g + 16 * wi + 4 * b + f;
            if (j >= n_desc) continue;
            const nps_row_desc d = desc[(uint64_t)s * n_desc + j];
            double wD = 0.0, wM = 0.0, cst = 0.0;  // weights of the dosage / the is-missing matrix; constant
            int used = 0, has_const = 0;
            const bool rie = d.ref_is_effect != 0;
            auto locus = [&]() {  // imputeLocusDosages nimpress.nim:417-447
                if (p.imp_locus == NPS_LOCUS_IGNORE) return;
                used = 1;
                has_const = 1;
                cst = (p.imp_locus == NPS_LOCUS_PS ? d.eaf * 2.0 : p.imp_locus == NPS_LOCUS_HOMREF ? (rie ? 2.0 : 0.0) : nan) *
                      d.beta;
            };
            if (d.kind == NPS_ROW_PRESENT) {
                const unsigned long long t = tally[j];
                const uint64_t nmiss = t >> 32, neff = t & 0xffffffffull, ngen = n_samples - nmiss;
                const double missingrate = (double)nmiss / (double)n_samples;
                if (missingrate > p.max_missing_rate) {  // :565-571
                    locus();
                } else {  // :582-585 -> imputeSampleDosages :450-481
                    used = 1;
                    double imp;
                    switch (p.imp_sample) {
                    case NPS_SAMPLE_PS: imp = d.eaf * 2.0; break;
                    case NPS_SAMPLE_HOMREF: imp = rie ? 2.0 : 0.0; break;
                    case NPS_SAMPLE_FAIL: imp = nan; break;
                    default:
                        if ((double)ngen >= p.min_cs)
                            imp = (double)neff / (double)ngen;
                        else
                            imp = p.imp_sample == NPS_SAMPLE_INT_PS ? d.eaf * 2.0 : nan;
                        break;
                    }
                    wD = d.beta;
                    // a missing genotype has code 3: it already got 3 x beta from the dosage matrix
                    wM = imp * d.beta - 3.0 * d.beta;
                }
            } else if (d.kind == NPS_ROW_ABSENT) {  // :536-551
                if (p.imp_missing == NPS_MISSING_HOMREF) {
                    used = 1;
                    has_const = 1;
                    cst = (rie ? 2.0 : 0.0) * d.beta;
                }
            } else if (d.kind == NPS_ROW_UNCOVERED || d.kind == NPS_ROW_FILTERED) {  // :526-531, :553-558
                locus();
            }  // else: the row is not part of this score
            const bool m_nan = !(fabs(wM) < __builtin_huge_val());  // NaN, or an infinite eaf: llrint(inf) is undefined
            VD[f] = llrint(ldexp(wD, fx));
            VM[f] = m_nan ? 0ll : llrint(ldexp(wM, fx));
            flag[f] = m_nan ? 1 : 0;
            any_m_nan |= m_nan;
            n_used += used;
            if (has_const) {
                if (!(fabs(cst) < __builtin_huge_val()))
                    c_nan = true;
                else
                    add_const(llrint(ldexp(cst, fx)));
            }
        }
        add_const(2 * VD[3]);  // the kernel's top operand is v3 - 128 (see above)
        // codes -> prefixes: the coefficients of v0 .. v3, scaled by 128
        const long long cD[4] = {128 * VD[0] - 32 * VD[1], 32 * VD[1] - 8 * VD[2], 8 * VD[2] - 2 * VD[3], 2 * VD[3]};
        const long long cM[4] = {128 * VM[0] - 32 * VM[1], 32 * VM[1] - 8 * VM[2], 8 * VM[2] - 2 * VM[3], 2 * VM[3]};
        const int cF[4] = {64 * flag[0] - 16 * flag[1], 16 * flag[1] - 4 * flag[2], 4 * flag[2] - flag[3], flag[3]};
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            int dd[8], dmm[8];
            weight_digits(cD[f], ND, 0, dd);
            weight_digits(cM[f], ND, coarse_missing, dmm);
            dmm[7] = cF[f];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                frag[0][q][f] |= (uint32_t)(dd[q] & 255) << (8 * b);
                frag[1][q][f] |= (uint32_t)(dmm[q] & 255) << (8 * b);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {  // digits 0..6 (those below ND), 7 = the flag
        if (q < 7 && q >= ND) continue;
        const int c = multi_col(S, ND, s, q < 7 ? q : ND), t = c >> 4, lane = (c & 15) + 16 * g;
#pragma unroll
        for (int dm = 0; dm < 2; ++dm)
#pragma unroll
            for (int f = 0; f < 4; ++f)
                table[(((((sb * 2 + (f >> 1)) * T + t) * 2 + dm) * 64 + lane) << 2) + 2 * (f & 1) + wi] =
                    frag[dm][q][f];
    }
    if (any_m_nan) atomicOr(&state[0].m_low, 1ull);
    if (n_used) atomicAdd(&state[s].nloci, n_used);
    if (c_nan) atomicOr(&state[s].const_nan, 1ull);
    if (c_lo | c_hi) {
        atomicAdd((unsigned long long *)&state[s].const_lo, c_lo);
        atomicAdd((unsigned long long *)&state[s].const_hi, c_hi);
    }
}

// one flag per superblock of the call: does any of its 128 rows have a missing genotype at all (the packer's
// whole-row tallies know)?  Where none has, the is-missing matrix of the superblock is zero and the product
// kernel skips its MFMAs -- cohorts of imputed hard calls have no missing genotypes anywhere.
__global__ __launch_bounds__(256) void multi_sbflag_kernel(const unsigned long long *__restrict__ tally, uint64_t n_rows,
                                                           uint32_t n_sb, uint32_t *__restrict__ flag) {
    const uint32_t sb = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (sb >= n_sb) return;
    const int lane = threadIdx.x & 63;
    bool any = false;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const uint64_t r = (uint64_t)sb * 128 + k * 64 + lane;
        any |= r < n_rows && (tally[r] >> 32) != 0;
    }
    const bool w = __any(any);
    if (lane == 0) flag[sb] = w ? 1u : 0u;
}

// ------------------------------------------------------------------------------------------
// The product.  Workgroup = WAVES waves; wave v owns GW groups of 32 samples; the workgroup walks the
// superblocks of its row chunk.  Everything the loop reads comes in by LDS-DMA (global_load_lds_dwordx4:
// 1 KiB per wave instruction, no VGPR destination), because the kernel sits at the 128-VGPR cap of a
// 16-wave workgroup and a register prefetch had nowhere to live (the compiler sank the loads to the end of
// the step and every wave then sat out a full HBM round trip per superblock: 10 ms of a 48 ms pass):
//   * the superblock's digit tables (T x 4 KiB, shared by the 16 waves): one 1 KiB piece per wave, staged
//     kStage superblocks ahead into the other half of a double buffer, one workgroup barrier per stage;
//   * the wave's own units (GW x 1 KiB per superblock): a private ring, kStage superblocks ahead.
// The wave counts its DMAs itself (s_waitcnt vmcnt(N), in order): per step its table pieces, then GW units.
// Per superblock and group of 32 samples: 40 VALU ops make the 8 operand register sets (2 sample halves x 2 prefix
// pairs x {dosage, is-missing}; see weight_digits) for up to 2 x 2 x T x 2 MFMAs of 16 x 16 x 64.  An MFMA of this
// shape holds the SIMD's vector issue for half of its 16 cycles: what the other waves' vector work may cost is
// two ops per MFMA, and the first version of this kernel (60 ops per group) sat exactly on that limit -- 69 % of the
// matrix pipe's cycles (profiles/r04_pmc_multi.txt).
//
// The vector work of one wave overlaps the matrix work of the other three waves of its SIMD only while
// the waves are out of step; a workgroup barrier puts them back in step, hence kStage > 1.
constexpr int kStage = 2;  // superblocks per barrier = prefetch distance of the units

template <int T, int GW, int WAVES>
struct __attribute__((aligned(16))) MultiLds {
    uint4 tab[2][kStage][2 * T * 2 * 64];
    uint4 unit[kStage][WAVES][GW][64];
};

// one LDS-DMA: lane l's 16 bytes at gsrc -> LDS byte address lds_dst + 16 l (lds_dst wave-uniform).  M0 is
// the destination base and compiler-reserved: written and restored inside the statement.
static __device__ __forceinline__ void glds16(const void *gsrc, uint32_t lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
template <int N>
static __device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
static __device__ __forceinline__ uint32_t lds_addr(const void *p) {  // LDS byte address of a __shared__ object
    return (uint32_t)(uintptr_t)p;
}

#ifndef NPS_MULTI_GW
#define NPS_MULTI_GW 2  // sample groups per wave; the workgroup has 32 / GW waves (-DNPS_MULTI_GW=4: measured equal)
#endif
// T tiles of 16 columns; the dosage matrix uses the tiles below TD, the is-missing matrix those below TM and, in
// calls with a NaN imputation value, those from TF0 on (all compile-time: the MFMAs of a step must stay one
// basic block for the scheduler to interleave the fragment reads with them).
template <int T, int TD, int TM, int TF0, int GW, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void multi_mfma_kernel(const uint4 *__restrict__ units, uint64_t n_groups,
                                                          uint64_t sb_first, uint32_t n_sb, uint32_t sb_per_chunk,
                                                          const uint4 *__restrict__ table,
                                                          int32_t *__restrict__ partial,
                                                          const MultiState *__restrict__ state,
                                                          const uint32_t *__restrict__ sb_has_missing) {
    constexpr int kTab = 2 * T * 2 * 64;   // uint4 per superblock: T x 4 KiB
    constexpr int kPieces = kTab / 64;     // 1 KiB pieces of a superblock's tables
    constexpr int kPW = (kPieces + WAVES - 1) / WAVES;  // table DMAs per wave and step (some pieces twice)
    __shared__ MultiLds<T, GW, WAVES> lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t chunk = blockIdx.y;
    const uint32_t sb_a = chunk * sb_per_chunk, sb_b = min(n_sb, sb_a + sb_per_chunk);
    const uint64_t g0 = ((uint64_t)blockIdx.x * WAVES + wave) * GW;
    if (sb_a >= sb_b) return;  // (whole workgroup)
    // the flag tiles of the is-missing matrix: only in calls with a NaN imputation value
    const bool m_flag = __builtin_amdgcn_readfirstlane((int)state[0].m_low) != 0;

    constexpr int NT16 = T;  // tiles of 16 columns
    v4i acc[GW][2][NT16];    // [group][sample half][tile]
#pragma unroll
    for (int a = 0; a < GW; ++a)
#pragma unroll
        for (int hs = 0; hs < 2; ++hs)
#pragma unroll
            for (int t = 0; t < NT16; ++t) acc[a][hs][t] = v4i{0, 0, 0, 0};

    // Every wave issues the same DMAs in every step (the counted waits rely on it): addresses past the
    // chunk / past the last group are clamped to valid ones, what they fetch is never used.
    const int piece0 = wave * kPW;  // (more slots than pieces: some pieces are fetched twice)
    auto dma_table = [&](uint32_t sb, int buf, int slot) {
        const uint32_t s = min(sb, sb_b - 1);
#pragma unroll
        for (int i = 0; i < kPW; ++i) {
            const int e = ((piece0 + i) % kPieces) * 64;
            glds16(table + (uint64_t)s * kTab + e + lane, lds_addr(&lds.tab[buf][slot][e]));
        }
    };
    auto dma_units = [&](uint32_t sb, int slot) {
        const uint32_t s = min(sb, sb_b - 1);
#pragma unroll
        for (int a = 0; a < GW; ++a) {
            const uint64_t g = min(g0 + a, n_groups - 1);
            glds16(units + ((sb_first + s) * n_groups + g) * 64 + lane, lds_addr(&lds.unit[slot][wave][a][0]));
        }
    };

#pragma unroll
    for (int k = 0; k < kStage; ++k) {
        dma_table(sb_a + k, 0, k);
        dma_units(sb_a + k, k);
    }
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    int buf = 0;
    for (uint32_t s0 = sb_a; s0 < sb_b; s0 += kStage, buf ^= 1) {
#pragma unroll
        for (int k = 0; k < kStage; ++k) {
            const uint32_t sb = s0 + k;
            // the units of this superblock were fetched one stage ago: younger DMAs = the later steps of that stage
            wait_vm<(kPW + GW) * (kStage - 1)>();
            uint32_t wd[GW][4];
#pragma unroll
            for (int a = 0; a < GW; ++a) {
                const uint4 u = lds.unit[k][wave][a][lane];
                wd[a][0] = u.x, wd[a][1] = u.y, wd[a][2] = u.z, wd[a][3] = u.w;
            }
            // the ring slot is free again once the read has returned
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dma_table(sb + kStage, buf ^ 1, k);
            dma_units(sb + kStage, k);
            if (sb < sb_b) {
                const bool has_m = sb_has_missing[sb] != 0;  // (uniform: a scalar load)
                uint32_t miss[GW][4];  // bit 2f of a byte: field f is code 3
                if (has_m) {
#pragma unroll
                    for (int a = 0; a < GW; ++a)
#pragma unroll
                        for (int i = 0; i < 4; ++i) miss[a][i] = wd[a][i] & (wd[a][i] >> 1) & 0x55555555u;
                }
#pragma unroll
                for (int p = 0; p < 2; ++p) {  // prefix pair: prefixes 2p, 2p + 1 (see weight_digits)
                    // first all of the dosage matrix, then all of the is-missing matrix: one set of operand
                    // registers alive at a time (the kernel sits at the 128-VGPR cap of a 16-wave workgroup)
#pragma unroll
                    for (int dm = 0; dm < 2; ++dm) {
                        if (dm == 1 && !has_m) break;
                        v4i A[GW][2];
#pragma unroll
                        for (int a = 0; a < GW; ++a)
#pragma unroll
                            for (int hs = 0; hs < 2; ++hs)
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    const int f = 2 * p + (i >> 1);
                                    const uint32_t w = dm == 0 ? wd[a][2 * hs + (i & 1)] : miss[a][2 * hs + (i & 1)];
                                    if (dm == 0)
                                        A[a][hs][i] = f == 0 ? (int)(w & 0x03030303u) : f == 1 ? (int)(w & 0x0F0F0F0Fu)
                                                    : f == 2 ? (int)(w & 0x3F3F3F3Fu) : (int)(w ^ 0x80808080u);
                                    else
                                        A[a][hs][i] = f == 0 ? (int)(w & 0x01010101u) : f == 1 ? (int)(w & 0x05050505u)
                                                    : f == 2 ? (int)(w & 0x15151515u) : (int)w;
                                }
#pragma unroll
                        for (int t = 0; t < NT16; ++t) {
                            if (dm == 0 ? t >= TD : (t >= TM && t < TF0)) continue;  // (compile-time)
                            if (dm == 1 && t >= TM && !m_flag) continue;             // (flag tiles: uniform, rare)
                            const uint4 bq = lds.tab[buf][k][((p * NT16 + t) * 2 + dm) * 64 + lane];
                            const v4i B = {(int)bq.x, (int)bq.y, (int)bq.z, (int)bq.w};
#pragma unroll
                            for (int a = 0; a < GW; ++a)
#pragma unroll
                                for (int hs = 0; hs < 2; ++hs)
                                    acc[a][hs][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[a][hs], B, acc[a][hs][t], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // the next stage's tables have landed (only this step's unit DMAs may still be in flight); every wave
        // has finished reading this stage's
        wait_vm<GW>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    wait_vm<0>();  // (clamped prefetches of the last stage: nothing may land after the workgroup has ended)
    // C/D map of the 16x16 MFMA shapes: column = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
    for (int a = 0; a < GW; ++a) {
        if (g0 + a >= n_groups) continue;
        int32_t *dst = partial + (((uint64_t)chunk * n_groups + g0 + a) * 32) * (T * 16);
#pragma unroll
        for (int hs = 0; hs < 2; ++hs)
#pragma unroll
            for (int t = 0; t < NT16; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int samp = 16 * hs + 4 * (lane >> 4) + r;
                    dst[(uint64_t)samp * (T * 16) + 16 * t + (lane & 15)] = acc[a][hs][t][r];
                }
    }
}

// per sample: the digit sums of all row chunks -> float64 per score, added to the running sums
template <int S, int ND>
__global__ __launch_bounds__(256) void multi_fold_kernel(const int32_t *__restrict__ partial, uint32_t n_chunks,
                                                         uint64_t n_groups, uint64_t n_samples,
                                                         const int *__restrict__ F, double *__restrict__ part,
                                                         int overwrite, MultiState *__restrict__ state) {
    constexpr int T = ((ND + 1) * S + 15) / 16;
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < (uint64_t)S) {  // this call's whole-locus constants -> the float64 running constant of the score
        const int s = (int)i;
        MultiState &st = state[s];
        const double c = ldexp((double)st.const_hi * 4294967296.0 + (double)st.const_lo, -F[s]);
        st.const_sum = (overwrite ? 0.0 : st.const_sum) + (st.const_nan ? __longlong_as_double(0x7ff8000000000000ll) : c);
        st.const_lo = st.const_hi = 0;
        st.const_nan = 0;
        if (s == 0) st.m_low = 0;
    }
    if (i >= n_samples) return;
    const uint64_t g = i >> 5, si = i & 31;
    long long sum[16 * T];  // the sample's row of the partial sums: one entry per column
#pragma unroll
    for (int c = 0; c < 16 * T; ++c) sum[c] = 0;
    for (uint32_t c = 0; c < n_chunks; ++c) {
        const int4 *src = reinterpret_cast<const int4 *>(partial + (((uint64_t)c * n_groups + g) * 32 + si) * (T * 16));
#pragma unroll
        for (int q = 0; q < 4 * T; ++q) {
            if (4 * q >= (ND + 1) * S) break;  // (columns behind the flags are zero)
            const int4 v = src[q];
            sum[4 * q] += v.x, sum[4 * q + 1] += v.y, sum[4 * q + 2] += v.z, sum[4 * q + 3] += v.w;
        }
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
        double v = 0.0;
#pragma unroll
        for (int d = ND - 1; d >= 0; --d) v = v * 256.0 + (double)sum[multi_col(S, ND, s, d)];
        v = ldexp(v, -F[s] - 7);  // the digits are those of weight x 2^F x 128 (weight_digits)
        if (sum[multi_col(S, ND, s, ND)] != 0) v = __longlong_as_double(0x7ff8000000000000ll);
        double *dst = part + (uint64_t)s * n_samples + i;
        *dst = overwrite ? v : *dst + v;
    }
}

// nimpress.nim:643-649 per score: (sum + constants) / (2 nloci) + offset
__global__ __launch_bounds__(256) void multi_finish_kernel(const double *__restrict__ part, uint64_t n_samples, int S,
                                                           const MultiState *__restrict__ state,
                                                           const double *__restrict__ offsets, int have_sums,
                                                           int normalise, double *__restrict__ scores) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const int s = blockIdx.y;
    if (i >= n_samples) return;
    double v = have_sums ? part[(uint64_t)s * n_samples + i] : 0.0;
    v += state[s].const_sum;
    if (normalise) {  // (normalise = 0: the un-normalised sums of a row-sharded run, for the all-reduce)
        v /= (double)state[s].nloci * 2.0;
        v += offsets[s];
    }
    scores[(uint64_t)s * n_samples + i] = v;
}

// ---- host side ------------------------------------------------------------------------------
hipError_t launch_multi_params(hipStream_t st, const unsigned long long *d_tally, const nps_row_desc *d_desc,
                               uint64_t n_desc, int S, const MultiPlan &pl, uint64_t n_samples, DevParams p,
                               const int *d_F, void *d_table, void *d_state, int coarse_missing) {
    if (n_desc == 0 || S == 0) return hipSuccess;
    const uint32_t n_sb = (uint32_t)((n_desc + 127) / 128);
    (void)hipGetLastError();
    hipLaunchKernelGGL(multi_params_kernel, dim3((uint32_t)(((uint64_t)n_sb * 8 + 255) / 256), (uint32_t)(S + 1)),
                       dim3(256), 0, st, d_tally, d_desc, n_desc, S, pl.ND, pl.T, n_samples, p, d_F, (uint32_t *)d_table,
                       (MultiState *)d_state, coarse_missing, n_sb);
    return hipGetLastError();
}

MultiPlan multi_plan(uint64_t n_samples, uint64_t n_rows, int S, int ND, int coarse_missing /* leading digits kept, 0 = all */, int cus) {
    MultiPlan pl;
    pl.ND = ND;
    pl.T = ((ND + 1) * S + 15) / 16;
    pl.TD = (ND * S + 15) / 16;
    pl.TM = coarse_missing > 0 && coarse_missing < ND ? std::min(pl.TD, (coarse_missing * S + 15) / 16) : pl.TD;
    pl.TF0 = (ND * S) / 16;
    pl.GW = NPS_MULTI_GW;
    pl.n_groups = (n_samples + 31) / 32;
    pl.n_sb = (uint32_t)((n_rows + 127) / 128);
    pl.tiles = (uint32_t)((pl.n_groups + 31) / 32);  // 32 groups = 1 024 samples per workgroup
    // one workgroup per CU at a time: enough row chunks (12 rounds of workgroups or more) that the last round is nearly
    // full -- the smallest count whose last round wastes at most 1.5 % of the grid's slots (489 tiles of 1 024 samples:
    // 7 chunks left 0.63 of a round empty, 41.6 ms; 12 chunks fill 22.92 rounds, 39.9 ms), every chunk at least 16
    // superblocks long
    const uint32_t q_hi = std::min<uint32_t>(64, std::max<uint32_t>(1, pl.n_sb / 16));
    uint32_t q_lo = (uint32_t)std::max<uint64_t>(1, ((uint64_t)cus * 12 + pl.tiles - 1) / pl.tiles);
    q_lo = std::min(q_lo, q_hi);
    uint32_t q = q_lo;
    double best = 2.0;
    for (uint32_t c = q_lo; c <= std::min<uint32_t>(q_hi, 4 * q_lo + 8); ++c) {
        const uint64_t wgs = (uint64_t)pl.tiles * c, rounds = (wgs + cus - 1) / cus;
        const double waste = 1.0 - (double)wgs / (double)(rounds * cus);
        if (waste < best - 1e-12) {
            best = waste;
            q = c;
        }
        if (waste <= 0.015) break;
    }
    // int32 digit sums: the operand bytes of one byte of genotypes (four rows, both matrices) are at most
    // 3 + 15 + 63 + 128 and 1 + 5 + 21 + 85, a digit at most 128: 41 088 per four rows, so a chunk holds at most
    // 2^31 / 10 272 rows = 1 633 superblocks
    q = std::max<uint32_t>(q, (pl.n_sb + 1535) / 1536);
    pl.sb_per_chunk = (pl.n_sb + q - 1) / q;
    pl.n_chunks = pl.sb_per_chunk ? (pl.n_sb + pl.sb_per_chunk - 1) / pl.sb_per_chunk : 0;
    return pl;
}

template <int T, int TD, int TM, int TF0>
static void launch_mfma_t(hipStream_t st, const MultiPlan &pl, const void *d_units, uint64_t sb_first,
                          const void *d_table, int32_t *d_partial, const void *d_state, const uint32_t *d_sbflag) {
    constexpr int GW = NPS_MULTI_GW, WAVES = 32 / GW;
    hipLaunchKernelGGL((multi_mfma_kernel<T, TD, TM, TF0, GW, WAVES>), dim3(pl.tiles, pl.n_chunks), dim3(64 * WAVES), 0,
                       st, (const uint4 *)d_units, pl.n_groups, sb_first, pl.n_sb, pl.sb_per_chunk,
                       (const uint4 *)d_table, d_partial, (const MultiState *)d_state, d_sbflag);
}

hipError_t launch_multi_mfma(hipStream_t st, const MultiPlan &pl, const void *d_units, uint64_t sb_first,
                             const void *d_table, int32_t *d_partial, const void *d_state,
                             const unsigned long long *d_tally, uint64_t n_rows, uint32_t *d_sbflag) {
    if (pl.n_sb == 0 || pl.n_groups == 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(multi_sbflag_kernel, dim3((pl.n_sb + 3) / 4), dim3(256), 0, st, d_tally, n_rows, pl.n_sb, d_sbflag);
#define NPS_MFMA_CASE(T_, TD_, TM_, TF0_)                                                                   \
    case T_ * 1000 + TD_ * 100 + TM_ * 10 + TF0_:                                                           \
        launch_mfma_t<T_, TD_, TM_, TF0_>(st, pl, d_units, sb_first, d_table, d_partial, d_state, d_sbflag); \
        break;
    // every (score count 1..8, 6 or 7 digits, full or 32-bit is-missing weights) of multi_plan
    switch (pl.T * 1000 + pl.TD * 100 + pl.TM * 10 + pl.TF0) {
        NPS_MFMA_CASE(1, 1, 1, 0)
        NPS_MFMA_CASE(2, 2, 2, 1)
        NPS_MFMA_CASE(2, 2, 1, 1)
        NPS_MFMA_CASE(3, 3, 3, 2)
        NPS_MFMA_CASE(3, 3, 2, 2)
        NPS_MFMA_CASE(3, 2, 2, 1)
        NPS_MFMA_CASE(4, 4, 4, 3)
        NPS_MFMA_CASE(4, 4, 2, 3)
        NPS_MFMA_CASE(4, 4, 3, 3)
        NPS_MFMA_CASE(4, 3, 3, 2)
        NPS_MFMA_CASE(4, 3, 2, 2)
        NPS_MFMA_CASE(4, 3, 3, 3)
        NPS_MFMA_CASE(4, 3, 2, 3)
    default: return hipErrorInvalidValue;
    }
#undef NPS_MFMA_CASE
    return hipGetLastError();
}

template <int S>
static void launch_fold_s(hipStream_t st, const MultiPlan &pl, const int32_t *d_partial, uint64_t n_samples,
                          const int *d_F, double *d_part, int overwrite, void *d_state) {
    const dim3 grid((uint32_t)std::max<uint64_t>(1, (n_samples + 255) / 256));
    if (pl.ND == 6)
        hipLaunchKernelGGL((multi_fold_kernel<S, 6>), grid, dim3(256), 0, st, d_partial, pl.n_chunks, pl.n_groups,
                           n_samples, d_F, d_part, overwrite, (MultiState *)d_state);
    else
        hipLaunchKernelGGL((multi_fold_kernel<S, 7>), grid, dim3(256), 0, st, d_partial, pl.n_chunks, pl.n_groups,
                           n_samples, d_F, d_part, overwrite, (MultiState *)d_state);
}

hipError_t launch_multi_fold(hipStream_t st, const MultiPlan &pl, const int32_t *d_partial, uint64_t n_samples, int S,
                             const int *d_F, double *d_part, int overwrite, void *d_state) {
    if (pl.ND != 6 && pl.ND != 7) return hipErrorInvalidValue;
    (void)hipGetLastError();
    switch (S) {
    case 1: launch_fold_s<1>(st, pl, d_partial, n_samples, d_F, d_part, overwrite, d_state); break;
    case 2: launch_fold_s<2>(st, pl, d_partial, n_samples, d_F, d_part, overwrite, d_state); break;
    case 3: launch_fold_s<3>(st, pl, d_partial, n_samples, d_F, d_part, overwrite, d_state); break;
    case 4: launch_fold_s<4>(st, pl, d_partial, n_samples, d_F, d_part, overwrite, d_state); break;
    case 5: launch_fold_s<5>(st, pl, d_partial, n_samples, d_F, d_part, overwrite, d_state); break;
    case 6: launch_fold_s<6>(st, pl, d_partial, n_samples, d_F, d_part, overwrite, d_state); break;
    case 7: launch_fold_s<7>(st, pl, d_partial, n_samples, d_F, d_part, overwrite, d_state); break;
    case 8: launch_fold_s<8>(st, pl, d_partial, n_samples, d_F, d_part, overwrite, d_state); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_multi_finish(hipStream_t st, const double *d_part, uint64_t n_samples, int S, const void *d_state,
                               const double *d_offsets, int have_sums, double *d_scores, int normalise) {
    if (n_samples == 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(multi_finish_kernel, dim3((uint32_t)((n_samples + 255) / 256), (uint32_t)S), dim3(256), 0, st,
                       d_part, n_samples, S, (const MultiState *)d_state, d_offsets, have_sums, normalise, d_scores);
    return hipGetLastError();
}

size_t multi_state_bytes() { return sizeof(MultiState); }

}  // namespace nps
