// nps_ds.hip -- FORMAT/DS (float32 dosage) path.  Build-defined extension: the reference decodes GT
// only (nimpress.nim:367-391); semantics follow the oracle's ref_raw_dosages_ds: one ALT dosage per
// sample, NaN = missing, effect allele == REF -> dosage = 2 - DS; then tallyAlleles (nim:32-47),
// the maxmis decision (:565-571), imputation (:417-481) and the accumulation (:639-641) unchanged.
//
// 4 bytes per genotype: purely HBM-bound, so the kernels are plain streaming kernels.  Per sample the
// accumulation does the reference's operations (float64 multiply, then add, rows in score-file
// order inside a row chunk; chunks are combined in fixed order); the row tallies use a fixed-shape
// tree.  Both are deterministic and differ from the reference's sequential sums in the last bits.
#include <algorithm>

#include "nps_kernels.h"

namespace nps {

static __device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
static __device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// one 256-thread block per row
__global__ __launch_bounds__(256) void ds_tally_kernel(const float *__restrict__ ds,
                                                       uint64_t stride_f, uint64_t n,
                                                       const nps_row_desc *__restrict__ desc,
                                                       uint64_t n_rows, DsTally *__restrict__ out) {
    __shared__ double s_sum[4];
    __shared__ uint32_t s_cnt[4];
    const uint64_t row = blockIdx.x;
    if (row >= n_rows) return;
    const bool rie = desc[row].ref_is_effect == 1;  // bit 1 set: the row already counts the effect allele
    const float4 *p = reinterpret_cast<const float4 *>(ds + row * stride_f);
    const uint64_t n4 = (n + 3) / 4;  // rows are zero padded to a multiple of 64 floats
    uint32_t cnt = 0;
    double sum = 0.0;
    auto take = [&](const float4 q, uint64_t v) {
        const float e[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (v * 4 + k < n) {
                if (isnan(e[k]))
                    cnt += 1;
                else
                    sum += rie ? 2.0 - (double)e[k] : (double)e[k];
            }
        }
    };
    // 8 independent 16-byte loads in flight per thread (the per-thread sum order stays v-ascending)
    uint64_t v = threadIdx.x;
    for (; v + 7 * 256 < n4; v += 8 * 256) {
        float4 q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = p[v + u * 256];
#pragma unroll
        for (int u = 0; u < 8; ++u) take(q[u], v + u * 256);
    }
    for (; v < n4; v += 256) take(p[v], v);
    sum = wave_sum_f64(sum);
    cnt = wave_sum_u32(cnt);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        s_sum[w] = sum;
        s_cnt[w] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        DsTally t;
        t.nmiss = (unsigned long long)s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        t.neff = ((s_sum[0] + s_sum[1]) + s_sum[2]) + s_sum[3];
        out[row] = t;
    }
}

__global__ __launch_bounds__(256) void ds_params_kernel(const DsTally *__restrict__ tally,
                                                        const nps_row_desc *__restrict__ desc,
                                                        uint64_t n_rows, uint64_t n_samples,
                                                        DevParams p, DsRowP *__restrict__ rowp,
                                                        nps_locus_stat *__restrict__ stats,
                                                        unsigned long long *__restrict__ nloci) {
    const uint64_t row = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    int used = 0;
    if (row < n_rows) {
        const uint64_t nmiss = tally[row].nmiss;
        const double neff = tally[row].neff;
        const uint64_t ngen = n_samples - nmiss;
        const double beta = desc[row].beta, eaf = desc[row].eaf;
        const bool rie = (desc[row].ref_is_effect & 1) != 0;       // homref imputation value
        const bool flip = desc[row].ref_is_effect == 1;            // dosage = 2 - DS
        const double nan = __longlong_as_double(0x7ff8000000000000ll);
        DsRowP r;
        r.beta = beta;
        r.imp = 0.0;
        r.cst = 0.0;
        r.mode = 0;
        r.rie = flip ? 1 : 0;
        int reason;
        const double missingrate = (double)nmiss / (double)n_samples;
        if (missingrate > p.max_missing_rate) {  // nim:565-571
            reason = NPS_REASON_MAXMIS;
            if (p.imp_locus != NPS_LOCUS_IGNORE) {
                r.cst = p.imp_locus == NPS_LOCUS_PS       ? eaf * 2.0
                        : p.imp_locus == NPS_LOCUS_HOMREF ? (rie ? 2.0 : 0.0)
                                                          : nan;
                r.mode = 2;
                used = 1;
            }
        } else {  // nim:450-481
            reason = NPS_REASON_GENOTYPED;
            used = 1;
            r.mode = 1;
            switch (p.imp_sample) {
            case NPS_SAMPLE_PS: r.imp = eaf * 2.0; break;
            case NPS_SAMPLE_HOMREF: r.imp = rie ? 2.0 : 0.0; break;
            case NPS_SAMPLE_FAIL: r.imp = nan; break;
            default:
                if ((double)ngen >= p.min_cs)
                    r.imp = neff / (double)ngen;
                else
                    r.imp = p.imp_sample == NPS_SAMPLE_INT_PS ? eaf * 2.0 : nan;
                break;
            }
        }
        rowp[row] = r;
        if (stats) {
            nps_locus_stat s;
            s.ngenotyped = ngen;
            s.nmissing = nmiss;
            s.neffect = neff;
            s.used = used;
            s.reason = reason;
            stats[row] = s;
        }
    }
    const int cnt = __syncthreads_count(used);
    if (threadIdx.x == 0 && cnt) atomicAdd(nloci, (unsigned long long)cnt);
}

// one thread = FOUR neighbouring samples x one chunk of rows, rows in order: score += dosage * beta (nim:639-641) -- a
// 16-byte non-temporal load per row and lane (the rows are read once here and are zero padded to 64 floats, so the load of a
// row's last lanes stays inside it), eight rows in flight.  grid.y row chunks add into separate partial-score planes (combined
// in fixed order by finish_kernel), so that a 200 000-sample cohort still fills the chip.
// (Until round 5: one sample and a 4-byte load per lane -- 2.6 TB/s over the two passes)
typedef float ds_v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void ds_accumulate_kernel(const float *__restrict__ ds,
                                                            uint64_t stride_f, uint64_t n,
                                                            const DsRowP *__restrict__ rowp_all,
                                                            uint64_t n_rows_all,
                                                            uint64_t rows_per_chunk,
                                                            double *__restrict__ part,
                                                            uint64_t part_chunk_stride) {
    const uint64_t i = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const uint64_t r_begin = (uint64_t)blockIdx.y * rows_per_chunk;
    if (i >= n || r_begin >= n_rows_all) return;
    const uint64_t n_rows = min(rows_per_chunk, n_rows_all - r_begin);
    const DsRowP *rowp = rowp_all + r_begin;
    double *part0 = part + (uint64_t)blockIdx.y * part_chunk_stride + i;
    const int live = (int)min((uint64_t)4, n - i);
    double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < live) s[k] = part0[k];
    const float *p = ds + r_begin * stride_f + i;
    auto apply = [&](const DsRowP &r, const ds_v4f e) {
        if (r.mode == 0) return;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            double d;
            if (r.mode == 2)
                d = r.cst;
            else if (isnan(e[k]))
                d = r.imp;
            else
                d = r.rie ? 2.0 - (double)e[k] : (double)e[k];
            s[k] += d * r.beta;
        }
    };
    auto load = [&](uint64_t row) { return __builtin_nontemporal_load(reinterpret_cast<const ds_v4f *>(p + row * stride_f)); };
    uint64_t row = 0;
    for (; row + 8 <= n_rows; row += 8) {
        ds_v4f q[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) q[u] = load(row + u);
#pragma unroll
        for (int u = 0; u < 8; ++u) apply(rowp[row + u], q[u]);
    }
    for (; row < n_rows; ++row) apply(rowp[row], load(row));
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < live) part0[k] = s[k];
}

// FORMAT/GT with ploidy > 2 (dosage can exceed 2, nimpress.nim:385-390): decoded to a float dosage
// row on the device and scored through the DS path.  Same allele rules as decode_gt_kernel.
template <typename T>
__global__ __launch_bounds__(256) void decode_gt_to_ds_kernel(const T *__restrict__ gts, uint64_t n,
                                                              int ploidy, int eaidx,
                                                              float *__restrict__ out) {
    const uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    int cnt = 0;
    bool miss = false;
    for (int k = 0; k < ploidy; ++k) {
        const int32_t a = (int32_t)gts[s * (uint64_t)ploidy + k];
        if (a >= 0) {
            if (a < 2)
                miss = true;
            else
                cnt += ((a >> 1) - 1) == eaidx;
        }
    }
    out[s] = miss ? __int_as_float(0x7fc00000) : (float)cnt;
}

hipError_t launch_decode_gt_to_ds(hipStream_t st, const void *d_gts, int elem_bytes, uint64_t n,
                                  int ploidy, int eaidx, float *d_out) {
    if (n == 0) return hipSuccess;
    (void)hipGetLastError();
    const dim3 grid((uint32_t)((n + 255) / 256)), block(256);
    switch (elem_bytes) {
    case 1:
        hipLaunchKernelGGL(decode_gt_to_ds_kernel<int8_t>, grid, block, 0, st, (const int8_t *)d_gts, n,
                           ploidy, eaidx, d_out);
        break;
    case 2:
        hipLaunchKernelGGL(decode_gt_to_ds_kernel<int16_t>, grid, block, 0, st, (const int16_t *)d_gts,
                           n, ploidy, eaidx, d_out);
        break;
    case 4:
        hipLaunchKernelGGL(decode_gt_to_ds_kernel<int32_t>, grid, block, 0, st, (const int32_t *)d_gts,
                           n, ploidy, eaidx, d_out);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// device copy of ref_synth_ds (oracle/refcpu.c)
static __device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__global__ __launch_bounds__(256) void synth_ds_kernel(float *__restrict__ ds, uint64_t stride_f,
                                                       uint64_t n, uint64_t row0, uint64_t gen_row0,
                                                       uint64_t seed,
                                                       const uint32_t *__restrict__ t_het,
                                                       const uint32_t *__restrict__ t_hom,
                                                       const uint32_t *__restrict__ t_miss) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t r = blockIdx.y;
    if (i >= n) return;
    const uint64_t h = mix64(mix64(seed ^ ((gen_row0 + r) * 0xD1B54A32D192ED03ull)) + i);
    const uint32_t g = (uint32_t)h, ms = (uint32_t)(h >> 32);
    float d;
    if (ms < t_miss[r]) {
        d = __int_as_float(0x7fc00000);
    } else {
        const int c = g < t_hom[r] ? 2 : (g < t_het[r] ? 1 : 0);
        const int noise = (int)((ms >> 8) & 255u) - 128;
        d = (float)c + (float)noise * (1.0f / 1024.0f);
        d = d < 0.0f ? 0.0f : (d > 2.0f ? 2.0f : d);
    }
    ds[(row0 + r) * stride_f + i] = d;
}

// ---- NPS_FMT_DS16 ----------------------------------------------------------------------------
// k <-> float32: the value of code k is float32(k / 10^4), correctly rounded -- what strtof gives for the decimal text -- for
// every k in 0 .. 20 000; 0xFFFF is a missing dosage.
// (ds16_value: nps_kernels.h)

// device copy of ref_synth_ds16 (oracle/refcpu.c): the genotype's dosage plus a decimal "imputation noise" of up to
// +-0.508 in steps of 0.004, clipped to [0, 2] -- values with at most three decimals, as an imputation tool prints them
__global__ __launch_bounds__(256) void synth_ds16_kernel(uint16_t *__restrict__ ds, uint64_t stride_e, uint64_t n, uint64_t row0,
                                                         uint64_t gen_row0, uint64_t seed, const uint32_t *__restrict__ t_het,
                                                         const uint32_t *__restrict__ t_hom,
                                                         const uint32_t *__restrict__ t_miss) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t r = blockIdx.y;
    if (i >= n) return;
    const uint64_t h = mix64(mix64(seed ^ ((gen_row0 + r) * 0xD1B54A32D192ED03ull)) + i);
    const uint32_t g = (uint32_t)h, ms = (uint32_t)(h >> 32);
    uint32_t k;
    if (ms < t_miss[r]) {
        k = 0xffffu;
    } else {
        const int c = g < t_hom[r] ? 2 : (g < t_het[r] ? 1 : 0);
        const int noise = (int)((ms >> 8) & 255u) - 128;
        const int v = c * 10000 + noise * 40;
        k = (uint32_t)(v < 0 ? 0 : (v > 20000 ? 20000 : v));
    }
    ds[(row0 + r) * stride_e + i] = (uint16_t)k;
}

// one workgroup per row: float32 -> k; a value that is neither NaN nor the value of some k marks the row
__global__ __launch_bounds__(256) void ds16_pack_kernel(const float *__restrict__ src, uint64_t src_stride_f, uint64_t n,
                                                        uint16_t *__restrict__ dst, uint64_t dst_stride_e,
                                                        unsigned char *__restrict__ bad) {
    const float *in = src + (uint64_t)blockIdx.x * src_stride_f;
    uint16_t *out = dst + (uint64_t)blockIdx.x * dst_stride_e;
    bool b = false;
    for (uint64_t i = threadIdx.x; i < n; i += 256) {
        const float v = in[i];
        uint32_t k = 0xffffu;
        if (v == v) {
            const double x = (double)v * 1e4;
            k = x >= 0.0 && x <= 20000.5 ? (uint32_t)__double2ll_rn(x) : 0xfffeu;
            if (k > 20000u || __float_as_uint(ds16_value(k)) != __float_as_uint(v)) {  // (-0.0 is refused too: its bits differ)
                b = true;
                k = 0u;
            }
        }
        out[i] = (uint16_t)k;
    }
    const int any = __syncthreads_or(b ? 1 : 0);
    if (threadIdx.x == 0) bad[blockIdx.x] = any ? 1 : 0;
}

__global__ __launch_bounds__(256) void ds16_unpack_kernel(const uint16_t *__restrict__ src, uint64_t src_stride_e, uint64_t n,
                                                          float *__restrict__ dst, uint64_t dst_stride_f) {
    const uint16_t *in = src + (uint64_t)blockIdx.x * src_stride_e;
    float *out = dst + (uint64_t)blockIdx.x * dst_stride_f;
    for (uint64_t i = threadIdx.x; i < n; i += 256) {
        const uint32_t k = in[i];
        out[i] = k == 0xffffu ? __int_as_float(0x7fc00000) : ds16_value(k);
    }
}

hipError_t launch_synth_ds16(hipStream_t st, uint16_t *d_ds, uint64_t stride_e, uint64_t n, uint64_t row0, uint64_t gen_row0,
                             uint64_t n_rows, uint64_t seed, const uint32_t *d_t_het, const uint32_t *d_t_hom,
                             const uint32_t *d_t_miss) {
    if (n_rows == 0 || n == 0) return hipSuccess;
    if (n_rows > 65535) return hipErrorInvalidValue;
    (void)hipGetLastError();
    hipLaunchKernelGGL(synth_ds16_kernel, dim3((uint32_t)((n + 255) / 256), (uint32_t)n_rows), dim3(256), 0, st, d_ds,
                       stride_e, n, row0, gen_row0, seed, d_t_het, d_t_hom, d_t_miss);
    return hipGetLastError();
}

hipError_t launch_ds16_pack(hipStream_t st, const float *d_src, uint64_t src_stride_f, uint64_t n, uint64_t n_rows,
                            uint16_t *d_dst, uint64_t dst_stride_e, unsigned char *d_bad) {
    if (n_rows == 0) return hipSuccess;
    if (n_rows > 0x7fffffffull) return hipErrorInvalidValue;
    (void)hipGetLastError();
    hipLaunchKernelGGL(ds16_pack_kernel, dim3((uint32_t)n_rows), dim3(256), 0, st, d_src, src_stride_f, n, d_dst, dst_stride_e,
                       d_bad);
    return hipGetLastError();
}

hipError_t launch_ds16_unpack(hipStream_t st, const uint16_t *d_src, uint64_t src_stride_e, uint64_t n, uint64_t n_rows,
                              float *d_dst, uint64_t dst_stride_f) {
    if (n_rows == 0) return hipSuccess;
    if (n_rows > 0x7fffffffull) return hipErrorInvalidValue;
    (void)hipGetLastError();
    hipLaunchKernelGGL(ds16_unpack_kernel, dim3((uint32_t)n_rows), dim3(256), 0, st, d_src, src_stride_e, n, d_dst, dst_stride_f);
    return hipGetLastError();
}

// ---- launchers --------------------------------------------------------------------------------
hipError_t launch_ds_tally(hipStream_t st, const float *d_ds, uint64_t stride_f, uint64_t n,
                           const nps_row_desc *d_desc, uint64_t n_rows, DsTally *d_tally) {
    if (n_rows == 0) return hipSuccess;
    if (n_rows > 0x7fffffffull) return hipErrorInvalidValue;
    (void)hipGetLastError();
    hipLaunchKernelGGL(ds_tally_kernel, dim3((uint32_t)n_rows), dim3(256), 0, st, d_ds, stride_f, n,
                       d_desc, n_rows, d_tally);
    return hipGetLastError();
}

hipError_t launch_ds_params(hipStream_t st, const DsTally *d_tally, const nps_row_desc *d_desc,
                            uint64_t n_rows, uint64_t n_samples, DevParams p, DsRowP *d_rowp,
                            nps_locus_stat *d_stats, unsigned long long *d_nloci) {
    if (n_rows == 0) return hipSuccess;
    (void)hipGetLastError();
    hipLaunchKernelGGL(ds_params_kernel, dim3((uint32_t)((n_rows + 255) / 256)), dim3(256), 0, st,
                       d_tally, d_desc, n_rows, n_samples, p, d_rowp, d_stats, d_nloci);
    return hipGetLastError();
}

hipError_t launch_ds_accumulate(hipStream_t st, const float *d_ds, uint64_t stride_f, uint64_t n,
                                const DsRowP *d_rowp, uint64_t n_rows, double *d_part,
                                uint32_t n_chunks, uint64_t part_chunk_stride) {
    if (n_rows == 0 || n == 0) return hipSuccess;
    if (n_chunks == 0 || n_chunks > 65535 || part_chunk_stride < n) return hipErrorInvalidValue;
    const uint64_t rows_per_chunk = std::max<uint64_t>(16, (n_rows + n_chunks - 1) / n_chunks);
    (void)hipGetLastError();
    hipLaunchKernelGGL(ds_accumulate_kernel, dim3((uint32_t)((n + 1023) / 1024), n_chunks), dim3(256), 0,
                       st, d_ds, stride_f, n, d_rowp, n_rows, rows_per_chunk, d_part,
                       part_chunk_stride);
    return hipGetLastError();
}

// one workgroup per row: does the row hold a value that is neither missing (NaN) nor a dosage (0 <= DS <= 2)?
__global__ __launch_bounds__(256) void ds_range_kernel(const float *__restrict__ ds, uint64_t stride_f, uint64_t n,
                                                       unsigned char *__restrict__ bad) {
    const float *row = ds + (uint64_t)blockIdx.x * stride_f;
    bool b = false;
    for (uint64_t i = threadIdx.x; i < n; i += 256) {
        const float v = row[i];
        b |= v == v && !(v >= 0.0f && v <= 2.0f);
    }
    const int any = __syncthreads_or(b ? 1 : 0);
    if (threadIdx.x == 0) bad[blockIdx.x] = any ? 1 : 0;
}

hipError_t launch_ds_range_check(hipStream_t st, const float *d_ds, uint64_t stride_f, uint64_t n, uint64_t n_rows,
                                 unsigned char *d_bad) {
    if (n_rows == 0) return hipSuccess;
    (void)hipGetLastError();
    for (uint64_t r0 = 0; r0 < n_rows; r0 += 1u << 30) {
        const uint64_t k = std::min<uint64_t>(1u << 30, n_rows - r0);
        hipLaunchKernelGGL(ds_range_kernel, dim3((uint32_t)k), dim3(256), 0, st, d_ds + r0 * stride_f, stride_f, n, d_bad + r0);
    }
    return hipGetLastError();
}

hipError_t launch_synth_ds(hipStream_t st, float *d_ds, uint64_t stride_f, uint64_t n, uint64_t row0,
                           uint64_t gen_row0, uint64_t n_rows, uint64_t seed, const uint32_t *d_t_het,
                           const uint32_t *d_t_hom, const uint32_t *d_t_miss) {
    if (n_rows == 0 || n == 0) return hipSuccess;
    if (n_rows > 65535) return hipErrorInvalidValue;
    (void)hipGetLastError();
    hipLaunchKernelGGL(synth_ds_kernel, dim3((uint32_t)((n + 255) / 256), (uint32_t)n_rows), dim3(256),
                       0, st, d_ds, stride_f, n, row0, gen_row0, seed, d_t_het, d_t_hom, d_t_miss);
    return hipGetLastError();
}

}  // namespace nps
