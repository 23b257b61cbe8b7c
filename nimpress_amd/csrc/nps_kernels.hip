// nps_kernels.hip -- gfx950 (MI355X, wave64) kernels for the nimpress per-variant inner loop.
//
//   decode_gt_kernel      getRawDosages + tallyAlleles        nimpress.nim:367-391, 32-47
//   tally_packed_kernel   tallyAlleles on 2-bit rows          nimpress.nim:32-47
//   row_params_kernel     maxmis decision + imputation value  nimpress.nim:565-571, 417-481
//   accumulate_kernel     scores[i] += dosages[i]*beta        nimpress.nim:639-641
//   finish_kernel         /= 2*nloci ; += offset              nimpress.nim:643-649
//
// Arithmetic: integer popcounts for the tallies (bit-exact), float64 for everything that touches
// a score.  No MFMA: this is a streaming weighted reduction (0.25 B per genotype).
//
// The accumulate kernel never converts a genotype to a float.  Per group of 4 rows it builds a
// 256-entry float64 table in LDS, T[idx] = ((l0[c0]+l1[c1])+l2[c2])+l3[c3] (idx: low code bits of the
// four rows in the low nibble, high code bits in the high nibble) with l_r[c] = LUT of row r (0*b,
// 1*b, 2*b, imputed*b), transposes four 16-sample words into sixteen byte indices with two merge
// stages, and does one ds_read_b64 + one v_add_f64 per FOUR genotypes.
#include <algorithm>

#include "nps_kernels.h"

namespace nps {

// ------------------------------------------------------------------------------------------
// wave64 / block reductions
static __device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;  // valid in lane 0
}

// sums (a,b,c) over the first `nthreads` threads of a group of waves; result valid in thread 0
// of the group.  `slot` = LDS scratch of 3*4 uint32 per group.
template <int WAVES>
static __device__ __forceinline__ void group_sum3(uint32_t &a, uint32_t &b, uint32_t &c,
                                                  uint32_t *slot, int tig /* thread in group */) {
    a = wave_sum(a);
    b = wave_sum(b);
    c = wave_sum(c);
    if (WAVES > 1) {
        const int w = tig >> 6;
        if ((tig & 63) == 0) {
            slot[w * 3 + 0] = a;
            slot[w * 3 + 1] = b;
            slot[w * 3 + 2] = c;
        }
        __syncthreads();
        if (tig == 0) {
            a = b = c = 0;
#pragma unroll
            for (int k = 0; k < WAVES; ++k) {
                a += slot[k * 3 + 0];
                b += slot[k * 3 + 1];
                c += slot[k * 3 + 2];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// decode: bcf_get_genotypes int32 buffer -> 2-bit codes + tally.
// hts-nim value(): a < 0 (vector-end pad) is skipped; a in {0,1} is the missing allele (value -1);
// otherwise allele index (a>>1)-1.  Any missing allele makes the sample missing (NaN is sticky in
// nimpress.nim:385-390).
// T = int32 (bcf_get_genotypes layout) or the int8 / int16 vector as a BCF record stores it: the
// end-of-vector pad and the typed missing value are negative in every width (0x81 / 0x8001 /
// 0x80000001, 0x80 / 0x8000 / 0x80000000), so sign extension classifies them as the int32 path does.
template <int PLOIDY, typename T>
__global__ __launch_bounds__(256) void decode_gt_kernel(const T *__restrict__ gts, uint64_t n,
                                                        int eaidx, uint32_t *__restrict__ out_group,
                                                        int row_in_group,
                                                        unsigned long long *__restrict__ tally) {
    __shared__ uint32_t red[4 * 3];
    const uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t code = 0;
    bool miss = false;
    if (s < n) {
        int32_t a[PLOIDY];
        if (PLOIDY == 2 && sizeof(T) == 4) {
            const int2 v = reinterpret_cast<const int2 *>(gts)[s];
            a[0] = v.x;
            a[PLOIDY - 1] = v.y;
        } else if (PLOIDY == 2 && sizeof(T) == 2) {  // one load per sample (the buffer may be host memory)
            const uint32_t v = reinterpret_cast<const uint32_t *>(gts)[s];
            a[0] = (int32_t)(int16_t)(v & 0xFFFFu);
            a[PLOIDY - 1] = (int32_t)(int16_t)(v >> 16);
        } else if (PLOIDY == 2 && sizeof(T) == 1) {
            const uint32_t v = reinterpret_cast<const uint16_t *>(gts)[s];
            a[0] = (int32_t)(int8_t)(v & 0xFFu);
            a[PLOIDY - 1] = (int32_t)(int8_t)(v >> 8);
        } else {
#pragma unroll
            for (int k = 0; k < PLOIDY; ++k) a[k] = (int32_t)gts[s * PLOIDY + k];
        }
#pragma unroll
        for (int k = 0; k < PLOIDY; ++k) {
            if (a[k] >= 0) {
                if (a[k] < 2)
                    miss = true;
                else
                    code += (((a[k] >> 1) - 1) == eaidx) ? 1u : 0u;
            }
        }
        code = miss ? NPS_CODE_MISSING : (code == 2 ? NPS_CODE_DOSAGE2 : code);
    }
    uint32_t sh = ((code & 1u) | ((code >> 1) << 4)) << plane_bit(lane & 15);
    sh |= __shfl_xor(sh, 1, 64);
    sh |= __shfl_xor(sh, 2, 64);
    sh |= __shfl_xor(sh, 4, 64);
    sh |= __shfl_xor(sh, 8, 64);
    if ((lane & 15) == 0 && s < n) out_group[(s >> 4) * 4 + row_in_group] = sh;

    uint32_t m = miss ? 1u : 0u, e = miss ? 0u : (code == NPS_CODE_DOSAGE2 ? 2u : code), z = 0;
    group_sum3<4>(m, e, z, red, threadIdx.x);
    if (threadIdx.x == 0 && (m | e))
        atomicAdd(tally, ((unsigned long long)m << 32) | (unsigned long long)e);
}

template <typename T>
static hipError_t launch_decode_gt_t(hipStream_t st, const T *d_gts, uint64_t n, int ploidy, int eaidx,
                                     uint32_t *d_group, int row_in_group, unsigned long long *d_tally) {
    const uint64_t blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffull) return hipErrorInvalidValue;
    (void)hipGetLastError();  // drop any stale sticky error: report this launch only
    if (ploidy == 2)
        hipLaunchKernelGGL((decode_gt_kernel<2, T>), dim3((uint32_t)blocks), dim3(256), 0, st, d_gts, n,
                           eaidx, d_group, row_in_group, d_tally);
    else if (ploidy == 1)
        hipLaunchKernelGGL((decode_gt_kernel<1, T>), dim3((uint32_t)blocks), dim3(256), 0, st, d_gts, n,
                           eaidx, d_group, row_in_group, d_tally);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_decode_gt(hipStream_t st, const void *d_gts, int elem_bytes, uint64_t n, int ploidy,
                            int eaidx, uint32_t *d_group, int row_in_group,
                            unsigned long long *d_tally) {
    if (n == 0) return hipSuccess;
    switch (elem_bytes) {
    case 1: return launch_decode_gt_t(st, (const int8_t *)d_gts, n, ploidy, eaidx, d_group, row_in_group, d_tally);
    case 2: return launch_decode_gt_t(st, (const int16_t *)d_gts, n, ploidy, eaidx, d_group, row_in_group, d_tally);
    case 4: return launch_decode_gt_t(st, (const int32_t *)d_gts, n, ploidy, eaidx, d_group, row_in_group, d_tally);
    default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------
// tally of packed rows.  For a device word w of sixteen 2-bit codes (00 dosage 0, 01 dosage 1,
// 11 dosage 2, 10 missing; high code bits four above the low ones):
//   popc(w)                       = het + 2*hom + miss
//   popc(w>>4 & ~w & 0x0F0F0F0F)  = miss
//   neffect = het + 2*hom = popc(w) - miss
static __device__ __forceinline__ void tally_word(uint32_t w, uint32_t &cw, uint32_t &cm) {
    cw += __popc(w);
    cm += __popc((w >> 4) & ~w & 0x0F0F0F0Fu);
}

// grid = (row groups, column chunks of 1024 words); every block adds its part of the four row
// tallies with one 64-bit integer atomic per row (exact, order independent); `tally` must be zero
__global__ __launch_bounds__(256) void tally_packed_kernel(const uint32_t *__restrict__ codes,
                                                           uint64_t stride_words, uint32_t n_words,
                                                           uint64_t n_rows,
                                                           unsigned long long *__restrict__ tally, int parity) {
    __shared__ uint32_t red[4 * 3];
    const uint64_t grp = blockIdx.x;
    const uint32_t c0 = blockIdx.y * 1024u;
    uint32_t cw[4] = {0, 0, 0, 0}, cm[4] = {0, 0, 0, 0};
    const uint4 *p = reinterpret_cast<const uint4 *>(codes) + grp * stride_words;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const uint32_t c = c0 + u * 256 + threadIdx.x;
        if (c < n_words) {
            uint4 q = p[c];  // the same 16 samples of the group's four rows
            if (parity) q.x = parity_fix(q.x, q.y, q.z, q.w);
            tally_word(q.x, cw[0], cm[0]);
            tally_word(q.y, cw[1], cm[1]);
            tally_word(q.z, cw[2], cm[2]);
            tally_word(q.w, cw[3], cm[3]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        uint32_t a = cw[r], b = cm[r], z = 0;
        __syncthreads();  // `red` is reused per row
        group_sum3<4>(a, b, z, red, threadIdx.x);
        const uint64_t row = grp * 4 + r;
        if (threadIdx.x == 0 && row < n_rows && (a | b))
            atomicAdd(&tally[row], ((unsigned long long)b << 32) | (unsigned long long)(a - b));
    }
}

hipError_t launch_tally_packed(hipStream_t st, const uint32_t *d_codes, uint64_t stride_words,
                               uint64_t n_samples, uint64_t n_rows, unsigned long long *d_tally, int parity) {
    if (n_rows == 0) return hipSuccess;
    const uint64_t n_words = words_for(n_samples);
    const uint64_t n_groups = (n_rows + 3) / 4;
    const uint64_t chunks = std::max<uint64_t>(1, (n_words + 1023) / 1024);
    if (n_groups > 0x7fffffffull || chunks > 65535) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(d_tally, 0, sizeof(unsigned long long) * n_rows, st);
    if (e != hipSuccess) return e;
    (void)hipGetLastError();  // drop any stale sticky error: report this launch only
    hipLaunchKernelGGL(tally_packed_kernel, dim3((uint32_t)n_groups, (uint32_t)chunks), dim3(256), 0,
                       st, d_codes, stride_words, (uint32_t)n_words, n_rows, d_tally, parity);
    return hipGetLastError();
}

// one contiguous packed row (nps_push_packed / nps_push_bed staging, read where it lies: the pinned host
// slot): tally + scatter into the interleaved batch.  bed_mode: -1 = native codes, 0 / 1 = a PLINK .bed row
// whose effect allele is A2 / A1 (bed_recode below).  grid = chunks of 2048 words; every block adds its part
// with one 64-bit atomic (`tally` is zero before the first row of a batch slot is pushed, as for the decode
// kernel)
static __device__ __forceinline__ uint32_t bed_recode(uint32_t w, int map, uint32_t c,
                                                       uint32_t n_words, uint32_t tail_mask);

__global__ __launch_bounds__(256) void tally_scatter_row_kernel(const uint32_t *__restrict__ row,
                                                                uint32_t n_words, int bed_mode,
                                                                uint32_t tail_mask,
                                                                uint32_t *__restrict__ out_group,
                                                                int row_in_group,
                                                                unsigned long long *__restrict__ tally) {
    __shared__ uint32_t red[4 * 3];
    uint32_t cw = 0, cm = 0, cz = 0;
    const uint32_t c0 = blockIdx.x * 2048u;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const uint32_t c = c0 + u * 256 + threadIdx.x;
        if (c < n_words) {
            uint32_t x = row[c];  // the staging row is in the C-ABI's bit order
            if (bed_mode >= 0) x = bed_recode(x, bed_mode, c, n_words, tail_mask);
            const uint32_t w = word_to_planes(x);
            tally_word(w, cw, cm);
            out_group[(uint64_t)c * 4 + row_in_group] = w;
        }
    }
    group_sum3<4>(cw, cm, cz, red, threadIdx.x);
    if (threadIdx.x == 0 && (cw | cm))
        atomicAdd(tally, ((unsigned long long)cm << 32) | (unsigned long long)(cw - cm));
}

hipError_t launch_tally_scatter_row(hipStream_t st, const uint32_t *row, uint64_t n_samples, int bed_mode,
                                    uint32_t *d_group, int row_in_group,
                                    unsigned long long *d_tally) {
    const uint64_t n_words = words_for(n_samples);
    if (n_words == 0) return hipSuccess;
    const uint32_t rem = (uint32_t)(n_samples & 15);
    const uint32_t tail_mask = rem ? ((1u << (2 * rem)) - 1u) : 0xffffffffu;
    (void)hipGetLastError();
    hipLaunchKernelGGL(tally_scatter_row_kernel, dim3((uint32_t)((n_words + 2047) / 2048)), dim3(256), 0, st, row,
                       (uint32_t)n_words, bed_mode, tail_mask, d_group, row_in_group, d_tally);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// PLINK 1 .bed rows (variant-major, 4 samples per byte, sample 0 in the low bits; values 0 = hom A1,
// 1 = missing, 2 = het, 3 = hom A2) are already 2-bit and sample-minor: the only work is a bit
// permutation per code, chosen by which allele the score row counts:
//   effect = A1:  0->3 (dosage 2)  2->1  3->0  1->2 (missing)   = bitwise NOT
//   effect = A2:  0->0  2->1  3->3 (dosage 2)  1->2 (missing)   = swap the two bits of every code
// PLINK 2 .pgen fixed-width hard-call records have the same packing with the code = number of ALT alleles
// (0, 1, 2; 3 = missing):
//   effect = ALT: 0->0  1->1  2->3 (dosage 2)  3->2 (missing)   = flip the low bit where the high bit is set
//   effect = REF: 0->3 (dosage 2)  1->1  2->0  3->2 (missing)   = (xnor(h, l), not h)
// map: NPS_MAP_* (0 .bed A2, 1 .bed A1, 2 .pgen ALT, 3 .pgen REF).  Bits past the last sample are cleared (the
// files pad the last byte with zeros).
static __device__ __forceinline__ uint32_t bed_recode(uint32_t w, int map, uint32_t c,
                                                       uint32_t n_words, uint32_t tail_mask) {
    const uint32_t h = (w >> 1) & 0x55555555u, l = w & 0x55555555u;
    uint32_t x;
    switch (map) {
    case 1: x = ~w; break;
    case 2: x = w ^ h; break;
    case 3: x = ((~(h ^ l) & 0x55555555u) << 1) | (~h & 0x55555555u); break;
    default: x = h | (l << 1); break;
    }
    return c + 1 == n_words ? (x & tail_mask) : x;
}

// rows [0,k) of a plain row-major staging buffer (src_stride_words apart, C-ABI bit order) -> the
// group-interleaved cohort layout at dst (first group of the destination).  mode: nullptr = rows are native codes;
// else per row 0 = .bed row, effect allele A2; 1 = .bed row, effect allele A1.
__global__ __launch_bounds__(256) void interleave_rows_kernel(const uint32_t *__restrict__ src,
                                                              uint64_t src_stride_words, uint64_t k,
                                                              uint32_t n_words, uint32_t tail_mask,
                                                              const uint8_t *__restrict__ mode,
                                                              uint32_t *__restrict__ dst,
                                                              uint64_t stride_words) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    const uint64_t g = blockIdx.y;
    if (c >= n_words) return;
    uint32_t q[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint64_t row = g * 4 + r;
        uint32_t w = 0;
        if (row < k) {
            w = src[row * src_stride_words + c];
            w = mode ? bed_recode(w, (int)mode[row], c, n_words, tail_mask)
                     : (c + 1 == n_words ? (w & tail_mask) : w);
        }
        q[r] = word_to_planes(w);
    }
    reinterpret_cast<uint4 *>(dst)[g * stride_words + c] = make_uint4(q[0], q[1], q[2], q[3]);
}

hipError_t launch_interleave_rows(hipStream_t st, const uint32_t *d_src, uint64_t src_stride_words,
                                  uint64_t k, uint64_t n_samples, const uint8_t *d_mode, uint32_t *d_dst,
                                  uint64_t stride_words) {
    if (k == 0 || n_samples == 0) return hipSuccess;
    const uint32_t n_words = (uint32_t)words_for(n_samples);
    const uint32_t rem = (uint32_t)(n_samples & 15);
    const uint32_t tail_mask = rem ? ((1u << (2 * rem)) - 1u) : 0xffffffffu;
    const uint64_t groups = (k + 3) / 4;
    if (groups > 65535) return hipErrorInvalidValue;
    (void)hipGetLastError();
    hipLaunchKernelGGL(interleave_rows_kernel, dim3((n_words + 255) / 256, (uint32_t)groups), dim3(256), 0,
                       st, d_src, src_stride_words, k, n_words, tail_mask, d_mode, d_dst, stride_words);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// nps_cohort_optimize: the parity layout (nps_kernels.h).  In place; its own inverse.
__global__ __launch_bounds__(256) void cohort_parity_kernel(uint32_t *__restrict__ codes, uint64_t stride_words,
                                                            uint32_t n_words) {
    const uint32_t c = blockIdx.x * 256 + threadIdx.x;
    const uint64_t g = blockIdx.y;
    if (c >= n_words) return;
    uint4 *p = reinterpret_cast<uint4 *>(codes) + g * stride_words + c;
    uint4 q = *p;
    q.x = parity_fix(q.x, q.y, q.z, q.w);
    *p = q;
}

hipError_t launch_cohort_parity(hipStream_t st, uint32_t *d_codes, uint64_t stride_words, uint64_t n_samples,
                                uint64_t n_rows) {
    const uint64_t n_words = words_for(n_samples), n_groups = (n_rows + 3) / 4;
    if (n_groups == 0 || n_words == 0) return hipSuccess;
    (void)hipGetLastError();
    for (uint64_t g0 = 0; g0 < n_groups; g0 += 65535) {
        const uint64_t k = std::min<uint64_t>(65535, n_groups - g0);
        hipLaunchKernelGGL(cohort_parity_kernel, dim3((uint32_t)((n_words + 255) / 256), (uint32_t)k), dim3(256), 0,
                           st, d_codes + g0 * stride_words * 4, stride_words, (uint32_t)n_words);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// per-row decision and LUT.  Mirrors nimpress.nim:565-571 (maxmis, strict > on a double quotient),
// :417-447 (locus imputation constant) and :450-481 (sample imputation value).
__global__ __launch_bounds__(256) void row_params_kernel(
    const unsigned long long *__restrict__ tally, const nps_row_desc *__restrict__ desc,
    uint64_t n_rows, uint64_t n_rows_pad, uint64_t n_samples, DevParams p, double *__restrict__ lut,
    nps_locus_stat *__restrict__ stats, unsigned long long *__restrict__ nloci) {
    const uint64_t row = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    int used = 0;
    if (row < n_rows_pad) {
        double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
        if (row < n_rows) {
            const unsigned long long t = tally[row];
            const uint64_t nmiss = t >> 32;
            const uint64_t neff = t & 0xffffffffull;
            const uint64_t ngen = n_samples - nmiss;
            const double beta = desc[row].beta, eaf = desc[row].eaf;
            const bool rie = desc[row].ref_is_effect != 0;
            const double nan = __longlong_as_double(0x7ff8000000000000ll);
            int reason;
            const double missingrate = (double)nmiss / (double)n_samples;
            if (missingrate > p.max_missing_rate) {  // :565-571
                reason = NPS_REASON_MAXMIS;
                if (p.imp_locus == NPS_LOCUS_IGNORE) {
                    used = 0;
                } else {
                    const double c = p.imp_locus == NPS_LOCUS_PS       ? eaf * 2.0
                                     : p.imp_locus == NPS_LOCUS_HOMREF ? (rie ? 2.0 : 0.0)
                                                                       : nan;
                    used = 1;
                    v0 = v1 = v2 = v3 = c * beta;
                }
            } else {  // :582-585
                reason = NPS_REASON_GENOTYPED;
                used = 1;
                double imp;
                switch (p.imp_sample) {
                case NPS_SAMPLE_PS: imp = eaf * 2.0; break;
                case NPS_SAMPLE_HOMREF: imp = rie ? 2.0 : 0.0; break;
                case NPS_SAMPLE_FAIL: imp = nan; break;
                default:
                    if ((double)ngen >= p.min_cs)
                        imp = (double)neff / (double)ngen;
                    else
                        imp = p.imp_sample == NPS_SAMPLE_INT_PS ? eaf * 2.0 : nan;
                    break;
                }
                v0 = 0.0 * beta;  // LUT is indexed by CODE: 0, 1 = dosage ; 2 = missing ; 3 = dosage 2
                v1 = 1.0 * beta;
                v2 = imp * beta;
                v3 = 2.0 * beta;
            }
            if (stats) {
                nps_locus_stat s;
                s.ngenotyped = ngen;
                s.nmissing = nmiss;
                s.neffect = (double)neff;
                s.used = used;
                s.reason = reason;
                stats[row] = s;
            }
        }
        double4 *l4 = reinterpret_cast<double4 *>(lut + row * 4);
        *l4 = make_double4(v0, v1, v2, v3);
    }
    const int cnt = __syncthreads_count(used);
    if (threadIdx.x == 0 && cnt) atomicAdd(nloci, (unsigned long long)cnt);
}

hipError_t launch_row_params(hipStream_t st, const unsigned long long *d_tally,
                             const nps_row_desc *d_desc, uint64_t n_rows, uint64_t n_rows_pad,
                             uint64_t n_samples, DevParams p, double *d_lut, nps_locus_stat *d_stats,
                             unsigned long long *d_nloci) {
    if (n_rows_pad == 0) return hipSuccess;
    const uint64_t blocks = (n_rows_pad + 255) / 256;
    (void)hipGetLastError();  // drop any stale sticky error: report this launch only
    hipLaunchKernelGGL(row_params_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, d_tally, d_desc,
                       n_rows, n_rows_pad, n_samples, p, d_lut, d_stats, d_nloci);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// 4 rows x 16 samples of 2-bit codes (device words) -> 16 byte indices: bit r = low code bit of row
// r, bit 4+r = its high code bit.  x[q] byte k = index of sample 4k+q.  Two merge stages: rows
// (0,1),(2,3) at 1-bit, then (01,23) at 2-bit granularity.
static __device__ __forceinline__ void transpose_4x16(uint32_t w0, uint32_t w1, uint32_t w2,
                                                      uint32_t w3, uint32_t (&x)[4]) {
    const uint32_t m1 = 0x55555555u, m2 = 0x33333333u;
    const uint32_t a0 = (w0 & m1) | ((w1 << 1) & ~m1);  // samples 4k, 4k+2 of rows 0,1
    const uint32_t a1 = ((w0 >> 1) & m1) | (w1 & ~m1);  // samples 4k+1, 4k+3
    const uint32_t b0 = (w2 & m1) | ((w3 << 1) & ~m1);
    const uint32_t b1 = ((w2 >> 1) & m1) | (w3 & ~m1);
    x[0] = (a0 & m2) | ((b0 << 2) & ~m2);         // samples 0,4,8,12
    x[2] = ((a0 >> 2) & m2) | (b0 & ~m2);         // samples 2,6,10,14
    x[1] = (a1 & m2) | ((b1 << 2) & ~m2);         // samples 1,5,9,13
    x[3] = ((a1 >> 2) & m2) | (b1 & ~m2);         // samples 3,7,11,15
}

constexpr int kAccThreads = 256;
constexpr int kGps = 4;  // row groups per stage (16 rows)

__global__ __launch_bounds__(kAccThreads) void accumulate_kernel(
    const uint32_t *__restrict__ codes, uint64_t stride_words, uint64_t n_rows, uint32_t n_words,
    const double *__restrict__ lut, uint32_t n_groups, uint32_t groups_per_chunk,
    double *__restrict__ part, uint64_t part_chunk_stride, int parity) {
    __shared__ double T[2][kGps][256];  // 16 KiB
    const int tid = threadIdx.x;
    const uint32_t col = blockIdx.x * kAccThreads + tid;
    const bool active = col < n_words;
    const uint32_t g_begin = blockIdx.y * groups_per_chunk;
    if (g_begin >= n_groups) return;  // block-uniform
    const uint32_t g_end = min(n_groups, g_begin + groups_per_chunk);

    double acc[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = 0.0;

    int buf = 0;
    for (uint32_t g0 = g_begin; g0 < g_end; g0 += kGps, buf ^= 1) {
        const uint32_t ng = min((uint32_t)kGps, g_end - g0);
        uint32_t w[kGps][4];
#pragma unroll
        for (int gg = 0; gg < kGps; ++gg) {
            // one 16-byte load = the thread's word column of the group's four rows (rows past
            // n_rows inside the last group have a zero LUT, whatever the buffer holds there)
            uint4 q = make_uint4(0, 0, 0, 0);
            if (active && gg < (int)ng)
                q = reinterpret_cast<const uint4 *>(codes)[(uint64_t)(g0 + gg) * stride_words + col];
            w[gg][0] = q.x;
            w[gg][1] = q.y;
            w[gg][2] = q.z;
            w[gg][3] = q.w;
        }
        // table for entry `tid` of each group of this stage (two LDS buffers -> one barrier/stage)
#pragma unroll
        for (int gg = 0; gg < kGps; ++gg) {
            if (gg < (int)ng) {
                // entry tid: code of row r = bit r | bit 4+r << 1, summed in row order
                const double *l = lut + (uint64_t)(g0 + gg) * 16;
                const int c1 = ((tid >> 1) & 1) | ((tid >> 4) & 2);
                const int c2 = ((tid >> 2) & 1) | ((tid >> 5) & 2), c3 = ((tid >> 3) & 1) | ((tid >> 6) & 2);
                // bit 4 of the address: the high code bit of slot 0, or (parity layout) the XOR of all four
                const int h0 = parity ? ((tid >> 4) ^ (tid >> 5) ^ (tid >> 6) ^ (tid >> 7)) & 1 : (tid >> 4) & 1;
                const int c0 = (tid & 1) | (h0 << 1);
                T[buf][gg][tid] = ((l[c0] + l[4 + c1]) + l[8 + c2]) + l[12 + c3];
            }
        }
        __syncthreads();
#pragma unroll
        for (int gg = 0; gg < kGps; ++gg) {
            if (gg < (int)ng) {
                uint32_t x[4];
                transpose_4x16(w[gg][0], w[gg][1], w[gg][2], w[gg][3], x);
                const double *Tg = T[buf][gg];
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[4 * k + q] += Tg[(x[q] >> (8 * k)) & 0xFFu];
            }
        }
    }
    if (active) {
        double *dst = part + (uint64_t)blockIdx.y * part_chunk_stride + (uint64_t)col * 16;
#pragma unroll
        for (int s = 0; s < 16; s += 2) {
            double2 v = *reinterpret_cast<double2 *>(dst + s);
            v.x += acc[s];
            v.y += acc[s + 1];
            *reinterpret_cast<double2 *>(dst + s) = v;
        }
    }
}

hipError_t launch_accumulate(hipStream_t st, const uint32_t *d_codes, uint64_t stride_words,
                             uint64_t n_rows, const double *d_lut, const AccumGeom &g,
                             double *d_part, int parity) {
    if (n_rows == 0 || g.n_words == 0) return hipSuccess;
    const uint32_t n_groups = (uint32_t)((n_rows + 3) / 4);
    const uint32_t tiles = (g.n_words + kAccThreads - 1) / kAccThreads;
    if (g.n_chunks == 0 || g.n_chunks > 65535 || g.groups_per_chunk == 0) return hipErrorInvalidValue;
    if ((uint64_t)g.n_chunks * g.groups_per_chunk < n_groups) return hipErrorInvalidValue;
    (void)hipGetLastError();  // drop any stale sticky error: report this launch only
    hipLaunchKernelGGL(accumulate_kernel, dim3(tiles, g.n_chunks), dim3(kAccThreads), 0, st, d_codes,
                       stride_words, n_rows, g.n_words, d_lut, n_groups, g.groups_per_chunk, d_part,
                       g.part_chunk_stride, parity);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// (no __restrict__: nps_normalize_device runs it in place, every thread on its own element)
__global__ __launch_bounds__(256) void finish_kernel(const double *part, uint32_t n_chunks,
                                                     uint64_t part_chunk_stride, uint64_t n_samples,
                                                     double const_sum, const unsigned long long *d_nloci,
                                                     uint64_t host_nloci, int normalise, double offset,
                                                     double *scores) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_samples) return;
    double s = 0.0;
    for (uint32_t c = 0; c < n_chunks; ++c) s += part[(uint64_t)c * part_chunk_stride + i];
    s += const_sum;
    if (normalise) {
        const uint64_t nloci = host_nloci + (d_nloci ? (uint64_t)*d_nloci : 0ull);
        s /= (double)nloci * 2.0;  // nimpress.nim:645 (nloci = 0: 0/0 = NaN, as in the reference)
        s += offset;               // nimpress.nim:649
    }
    scores[i] = s;
}

hipError_t launch_finish(hipStream_t st, const double *d_part, uint32_t n_chunks,
                         uint64_t part_chunk_stride, uint64_t n_samples, double const_sum,
                         const unsigned long long *d_nloci, uint64_t host_nloci, int normalise,
                         double offset, double *d_scores) {
    if (n_samples == 0) return hipSuccess;
    const uint64_t blocks = (n_samples + 255) / 256;
    (void)hipGetLastError();  // drop any stale sticky error: report this launch only
    hipLaunchKernelGGL(finish_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, d_part, n_chunks,
                       part_chunk_stride, n_samples, const_sum, d_nloci, host_nloci, normalise, offset,
                       d_scores);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// synthetic cohort generator -- device copy of ref_synth_code (oracle/refcpu.c); the two must
// produce identical codes (tests/test_gpu_parity.py::test_synth_matches_oracle).
static __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void synth_gt_kernel(uint32_t *__restrict__ codes,
                                                       uint64_t stride_words, uint64_t n_samples,
                                                       uint32_t n_words, uint64_t row0, uint64_t gen_row0,
                                                       uint64_t n_rows, uint64_t seed,
                                                       const uint32_t *__restrict__ t_het,
                                                       const uint32_t *__restrict__ t_hom,
                                                       const uint32_t *__restrict__ t_miss) {
    const uint32_t word = blockIdx.x * 256 + threadIdx.x;
    const uint64_t g = blockIdx.y;  // group index relative to row0 (row0 is a multiple of 4)
    if (word >= n_words) return;
    uint32_t out[4] = {0, 0, 0, 0};
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const uint64_t r = g * 4 + rr;  // row relative to row0 = index into the threshold arrays
        if (r < n_rows) {
            const uint64_t key = mix64(seed ^ ((gen_row0 + r) * 0xD1B54A32D192ED03ull));
            const uint32_t th = t_het[r], tm = t_hom[r], tmi = t_miss[r];
            uint32_t w = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint64_t s = (uint64_t)word * 16 + k;
                if (s < n_samples) {
                    const uint64_t h = mix64(key + s);
                    const uint32_t gq = (uint32_t)h, ms = (uint32_t)(h >> 32);
                    const uint32_t c = ms < tmi ? NPS_CODE_MISSING
                                                : (gq < tm ? NPS_CODE_DOSAGE2 : (gq < th ? 1u : 0u));
                    w |= c << (2 * k);
                }
            }
            out[rr] = word_to_planes(w);
        }
    }
    reinterpret_cast<uint4 *>(codes)[((row0 >> 2) + g) * stride_words + word] =
        make_uint4(out[0], out[1], out[2], out[3]);
}

hipError_t launch_synth_gt(hipStream_t st, uint32_t *d_codes, uint64_t stride_words,
                           uint64_t n_samples, uint64_t row0, uint64_t gen_row0, uint64_t n_rows, uint64_t seed,
                           const uint32_t *d_t_het, const uint32_t *d_t_hom,
                           const uint32_t *d_t_miss) {
    const uint64_t n_words = words_for(n_samples);
    if (n_rows == 0 || n_words == 0) return hipSuccess;
    const uint64_t n_groups = (n_rows + 3) / 4;
    if ((row0 & 3) || n_groups > 65535) return hipErrorInvalidValue;  // caller splits row ranges
    (void)hipGetLastError();  // drop any stale sticky error: report this launch only
    hipLaunchKernelGGL(synth_gt_kernel, dim3((uint32_t)((n_words + 255) / 256), (uint32_t)n_groups),
                       dim3(256), 0, st, d_codes, stride_words, n_samples, (uint32_t)n_words, row0,
                       gen_row0, n_rows, seed, d_t_het, d_t_hom, d_t_miss);
    return hipGetLastError();
}

}  // namespace nps
