// nps_fused.hip -- the fused, single-read kernel for a resident 2-bit cohort (gfx950, wave64).
//
// Problem: tallyAlleles (nimpress.nim:32-47) needs a WHOLE ROW (all samples) before any sample of
// that row can be accumulated (the maxmis decision :565-571 and the internal imputation value
// :470-477 depend on it), while the accumulation (nimpress.nim:639-641) wants every sample to
// keep its float64 partial score on chip across ALL rows.  A two-kernel design reads the matrix
// twice.  This kernel reads it once:
//
//   * grid = Q teams x P workgroups, all co-resident (cooperative launch, <= 1 per CU).
//     Workgroup (q,p) owns the sample slice [p*T*16, (p+1)*T*16) -- one 16-sample word column per
//     thread, 16 float64 accumulators per thread -- and the row batches b = q, q+Q, q+2Q, ...
//     (16 rows = 4 groups per batch).
//   * per batch: 16 coalesced dword loads per thread into a 3-deep REGISTER ring; partial tallies
//     by popcount + DPP reduce-scatter + LDS add; one 64-bit agent-scope atomic add per row
//     publishes (arrivals<<56 | nmiss<<28 | neffect); the batch is consumed one ring step later:
//     one wave polls the 16 tally words until all P slices have arrived, 16 lanes derive the rows'
//     LUTs, the workgroup builds four 256-entry float64 tables in LDS, and every thread does one
//     ds_read_b64 + one v_add_f64 per FOUR genotypes straight from its ring registers.
//   * inter-workgroup traffic is 8-byte agent-scope atomics on both sides (an `sc1` form measured
//     valid on gfx950, MI355X_MICROARCH.md "Valid forms"); every spin is bounded and sets a
//     timeout word instead of hanging.
//
// HBM traffic = the matrix once (+ 8 B of atomics per row per slice, + row descriptors).
#include "nps_kernels.h"

namespace nps {

constexpr int kRowsPerBatch = 16;
constexpr uint32_t kSpinLimit = 1u << 20;  // ~1 s of polling before a wait gives up

struct FusedArgs {
    const uint32_t *codes;
    uint64_t stride_words;
    uint64_t n_rows;
    uint64_t n_samples;
    uint32_t n_words;
    uint32_t n_batches;
    uint32_t P, Q;
    const nps_row_desc *desc;
    DevParams prm;
    unsigned long long *tally;  // [n_batches*16], zeroed before the launch
    nps_locus_stat *stats;      // [n_rows] or nullptr
    unsigned long long *nloci;
    double *part;               // [Q][part_team_stride]
    uint64_t part_team_stride;
    unsigned int *timeout;      // zeroed before the launch
};

// ---- DPP helpers --------------------------------------------------------------------------
template <int CTRL>
static __device__ __forceinline__ uint32_t dpp(uint32_t v) {
    // out-of-range source lanes read 0 (bound_ctrl)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
constexpr int kQuadSwap1 = 0xB1;  // quad_perm:[1,0,3,2]  lane ^ 1
constexpr int kQuadSwap2 = 0x4E;  // quad_perm:[2,3,0,1]  lane ^ 2
constexpr int kRowShr4 = 0x114;
constexpr int kRowShr8 = 0x118;

// popcount triple of one word, packed (popc(w)+popc(w&0xAAAAAAAA)) << 16 | missing
static __device__ __forceinline__ uint32_t tally_pack(uint32_t w) {
    const uint32_t t = __popc(w) + __popc(w & 0xAAAAAAAAu);
    const uint32_t m = __popc(w & (w >> 1) & 0x55555555u);
    return (t << 16) | m;
}

// 4 rows x 16 samples of 2-bit codes -> 16 byte indices, then the bank-spreading fold
// e -> e ^ (e >> 5) (tables are stored at the folded position).  x[q] byte k = sample 4k+q.
static __device__ __forceinline__ void transpose_fold_4x16(uint32_t w0, uint32_t w1, uint32_t w2,
                                                           uint32_t w3, uint32_t (&x)[4]) {
    const uint32_t m2 = 0x33333333u, m4 = 0x0F0F0F0Fu;
    const uint32_t e01 = (w0 & m2) | ((w1 << 2) & ~m2);
    const uint32_t o01 = ((w0 >> 2) & m2) | (w1 & ~m2);
    const uint32_t e23 = (w2 & m2) | ((w3 << 2) & ~m2);
    const uint32_t o23 = ((w2 >> 2) & m2) | (w3 & ~m2);
    x[0] = (e01 & m4) | ((e23 << 4) & ~m4);
    x[1] = (o01 & m4) | ((o23 << 4) & ~m4);
    x[2] = ((e01 >> 4) & m4) | (e23 & ~m4);
    x[3] = ((o01 >> 4) & m4) | (o23 & ~m4);
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] ^= (x[q] >> 5) & 0x07070707u;
}

// LDS layout (bytes)
//   [0, 8192)        float64 tables, 4 groups x 256 entries (folded index)
//   [8192, 8704)     row LUTs of the batch being consumed, 16 rows x 4 float64
//   [8704, 8832)     partial tallies, 2 parities x 16 rows x uint32
struct __attribute__((aligned(16))) FusedLds {
    double table[4][256];
    double lut[kRowsPerBatch][4];
    uint32_t tally[2][kRowsPerBatch];
    uint32_t pad[4];
};

template <int T>
__global__ __launch_bounds__(T) void fused_kernel(const FusedArgs a) {
    static_assert(T == 1024 || T == 512, "workgroup size");
    __shared__ FusedLds lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t team = blockIdx.y;   // grid = (P slices, Q teams)
    const uint32_t slice = blockIdx.x;
    const uint32_t col = slice * T + tid;
    const bool active = col < a.n_words;
    const uint32_t voff = col * 4u;     // byte offset inside a row; rows are < 4 GB
    const uint32_t row_bytes = a.n_words * 4u;
    // local batch k of this team is global batch team + k*Q
    const uint32_t n_local = a.n_batches > team ? (a.n_batches - team + a.Q - 1) / a.Q : 0;

    if (tid < 2 * kRowsPerBatch) (&lds.tally[0][0])[tid] = 0;
    __syncthreads();

    double acc[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = 0.0;
    uint32_t ring[3][kRowsPerBatch];
    uint32_t nloci_local = 0;  // meaningful in wave 0 of slice 0
    bool timed_out = false;    // wave 0 only: stop waiting once any bounded wait has expired

    // One buffer descriptor per row (wave-uniform, lives in SGPRs): the hardware range check
    // returns 0 for columns past the end of the row and for rows past the end of the matrix, so
    // the loads need no per-lane predication and no 64-bit per-lane addresses.
    auto load_batch = [&](uint32_t k, uint32_t(&dst)[kRowsPerBatch]) {
        const uint64_t row0 = (uint64_t)(team + (uint64_t)k * a.Q) * kRowsPerBatch;
        const bool batch_ok = k < n_local;
#pragma unroll
        for (int r = 0; r < kRowsPerBatch; ++r) {
            const uint64_t row = row0 + r;
            const bool ok = batch_ok && row < a.n_rows;
            const uint32_t *base = a.codes + (ok ? row : 0) * a.stride_words;
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<uint32_t *>(base), 0, ok ? row_bytes : 0u, 0x00020000);
            dst[r] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, 0, 0);
        }
    };

    // partial tally of this workgroup's slice for local batch k (data in `src`) -> global atomics
    auto tally_publish = [&](uint32_t k, const uint32_t(&src)[kRowsPerBatch]) {
        // batches past the end (k >= n_local, only in the last ring turn) run through unchanged:
        // their loads returned zeros and all their rows are >= n_rows, so nothing is published
        const int par = k & 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint32_t a0 = tally_pack(src[4 * g + 0]), a1 = tally_pack(src[4 * g + 1]);
            uint32_t a2 = tally_pack(src[4 * g + 2]), a3 = tally_pack(src[4 * g + 3]);
            // reduce-scatter over lane^1: even lanes keep rows 0,1 ; odd lanes rows 2,3
            const bool odd = lane & 1;
            const uint32_t x0 = odd ? a2 : a0, y0 = odd ? a0 : a2;
            const uint32_t x1 = odd ? a3 : a1, y1 = odd ? a1 : a3;
            const uint32_t b0 = x0 + dpp<kQuadSwap1>(y0);
            const uint32_t b1 = x1 + dpp<kQuadSwap1>(y1);
            // lane^2: (lane&2)==0 keeps b0, else b1
            const bool hi = lane & 2;
            const uint32_t x = hi ? b1 : b0, y = hi ? b0 : b1;
            uint32_t c = x + dpp<kQuadSwap2>(y);
            // sum over lanes congruent mod 4 within each 16-lane DPP row
            c += dpp<kRowShr4>(c);
            c += dpp<kRowShr8>(c);
            // lanes 12..15 of each DPP row hold the row totals; lane&3 -> row: 0,2,1,3
            if ((lane & 12) == 12) {
                const int r = ((lane & 1) << 1) | ((lane >> 1) & 1);
                atomicAdd(&lds.tally[par][4 * g + r], c);
            }
        }
        __syncthreads();
        if (tid < kRowsPerBatch) {
            const uint32_t v = lds.tally[par][tid];
            lds.tally[par][tid] = 0;  // reused two batches later, many barriers away
            const uint64_t row = (uint64_t)(team + (uint64_t)k * a.Q) * kRowsPerBatch + tid;
            if (row < a.n_rows) {
                const uint64_t t = v >> 16, m = v & 0xFFFFu;
                const uint64_t neff = t - 3 * m;
                const unsigned long long add = (1ull << 56) | (m << 28) | neff;
                __hip_atomic_fetch_add(&a.tally[row], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };

    // wait for the complete tallies of local batch k, derive the row LUTs, build the tables and
    // accumulate from `src`.  When a bounded wait expires the timeout word is set (the host then
    // reports NPS_E_TIMEOUT and discards the scores) and every later wait in every workgroup
    // falls through at once, so the grid always drains.
    auto consume = [&](uint32_t k, const uint32_t(&src)[kRowsPerBatch]) {
        // (batches past the end: every row invalid -> zero LUTs -> +0.0 to every accumulator)
        const uint64_t row0 = (uint64_t)(team + (uint64_t)k * a.Q) * kRowsPerBatch;
        if (tid < 64) {  // wave 0: lanes 0..15 own one row each
            const uint64_t row = row0 + lane;
            const bool valid = lane < kRowsPerBatch && row < a.n_rows;
            unsigned long long x = 0;
            bool ok = !valid;
            uint32_t spins = 0;
            while (true) {
                if (!ok) {
                    x = __hip_atomic_load(&a.tally[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = (uint32_t)(x >> 56) == a.P;
                }
                if (__all(ok)) break;
                if (timed_out) break;
                __builtin_amdgcn_s_sleep(4);
                ++spins;
                if ((spins & 255u) == 0) {
                    const unsigned int t =
                        __hip_atomic_load(a.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (t != 0 || spins >= kSpinLimit) {
                        if (lane == 0)
                            __hip_atomic_store(a.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        timed_out = true;
                        break;
                    }
                }
            }
            int used = 0;
            if (valid && ok) {
                const uint64_t nmiss = (x >> 28) & 0xFFFFFFFull, neff = x & 0xFFFFFFFull;
                const uint64_t ngen = a.n_samples - nmiss;
                const double beta = a.desc[row].beta, eaf = a.desc[row].eaf;
                const bool rie = a.desc[row].ref_is_effect != 0;
                const double nan = __longlong_as_double(0x7ff8000000000000ll);
                double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
                int reason;
                const double missingrate = (double)nmiss / (double)a.n_samples;
                if (missingrate > a.prm.max_missing_rate) {  // nimpress.nim:565-571
                    reason = NPS_REASON_MAXMIS;
                    if (a.prm.imp_locus != NPS_LOCUS_IGNORE) {
                        const double c = a.prm.imp_locus == NPS_LOCUS_PS       ? eaf * 2.0
                                         : a.prm.imp_locus == NPS_LOCUS_HOMREF ? (rie ? 2.0 : 0.0)
                                                                               : nan;
                        used = 1;
                        v0 = v1 = v2 = v3 = c * beta;
                    }
                } else {  // nimpress.nim:450-481
                    reason = NPS_REASON_GENOTYPED;
                    used = 1;
                    double imp;
                    switch (a.prm.imp_sample) {
                    case NPS_SAMPLE_PS: imp = eaf * 2.0; break;
                    case NPS_SAMPLE_HOMREF: imp = rie ? 2.0 : 0.0; break;
                    case NPS_SAMPLE_FAIL: imp = nan; break;
                    default:
                        if ((double)ngen >= a.prm.min_cs)
                            imp = (double)neff / (double)ngen;
                        else
                            imp = a.prm.imp_sample == NPS_SAMPLE_INT_PS ? eaf * 2.0 : nan;
                        break;
                    }
                    v0 = 0.0 * beta;
                    v1 = 1.0 * beta;
                    v2 = 2.0 * beta;
                    v3 = imp * beta;
                }
                lds.lut[lane][0] = v0;
                lds.lut[lane][1] = v1;
                lds.lut[lane][2] = v2;
                lds.lut[lane][3] = v3;
                if (slice == 0 && a.stats) {
                    nps_locus_stat s;
                    s.ngenotyped = ngen;
                    s.nmissing = nmiss;
                    s.neffect = (double)neff;
                    s.used = used;
                    s.reason = reason;
                    a.stats[row] = s;
                }
            } else if (lane < kRowsPerBatch) {  // rows past the end of the matrix: zero LUT
                lds.lut[lane][0] = lds.lut[lane][1] = lds.lut[lane][2] = lds.lut[lane][3] = 0.0;
            }
            nloci_local += (uint32_t)__popcll(__ballot(used != 0));
        }
        __syncthreads();
        // tables: entry e of group g at folded position e ^ (e >> 5)
#pragma unroll
        for (int i = tid; i < 4 * 256; i += T) {
            const int g = i >> 8, e = i & 255;
            const double v = ((lds.lut[4 * g][e & 3] + lds.lut[4 * g + 1][(e >> 2) & 3]) +
                              lds.lut[4 * g + 2][(e >> 4) & 3]) +
                             lds.lut[4 * g + 3][e >> 6];
            lds.table[g][e ^ (e >> 5)] = v;
        }
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint32_t x[4];
            transpose_fold_4x16(src[4 * g], src[4 * g + 1], src[4 * g + 2], src[4 * g + 3], x);
            const double *Tg = lds.table[g];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[4 * kk + q] += Tg[(x[q] >> (8 * kk)) & 0xFFu];
                // Pin the four adds here.  Without the opaque use hipcc (ROCm 7.2) sinks the
                // v_add_f64 of a whole batch below the next batch's barrier and spills the looked-up
                // values to scratch; with it, four lookups are in flight per wave at a time, which
                // also bounds the live registers (ring + accumulators already hold 80 of the 128 a
                // 16-wave workgroup may use).
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(acc[4 * kk + q]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // software pipeline: loads run 2 batches ahead, tallies 1 batch ahead of the accumulation
    load_batch(0, ring[0]);
    load_batch(1, ring[1]);
    tally_publish(0, ring[0]);
    for (uint32_t k = 0; k < n_local; k += 3) {
        load_batch(k + 2, ring[2]);
        tally_publish(k + 1, ring[1]);
        consume(k, ring[0]);
        load_batch(k + 3, ring[0]);
        tally_publish(k + 2, ring[2]);
        consume(k + 1, ring[1]);
        load_batch(k + 4, ring[1]);
        tally_publish(k + 3, ring[0]);
        consume(k + 2, ring[2]);
    }

    if (active) {
        double *dst = a.part + (uint64_t)team * a.part_team_stride + (uint64_t)col * 16;
#pragma unroll
        for (int s = 0; s < 16; s += 2)
            *reinterpret_cast<double2 *>(dst + s) = make_double2(acc[s], acc[s + 1]);
    }
    if (slice == 0 && tid == 0 && nloci_local) atomicAdd(a.nloci, (unsigned long long)nloci_local);
}

// part0[i] += sum_q part[q][i]  (fold the teams' partial scores into chunk 0 of the context)
__global__ __launch_bounds__(256) void fold_kernel(const double *__restrict__ part, uint32_t Q,
                                                   uint64_t team_stride, uint64_t n,
                                                   double *__restrict__ part0) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (uint32_t q = 0; q < Q; ++q) s += part[(uint64_t)q * team_stride + i];
    part0[i] += s;
}

// ---- host side ------------------------------------------------------------------------------
hipError_t fused_plan(int device, uint64_t n_samples, uint64_t n_rows, FusedPlan *plan) {
    plan->ok = false;
    if (n_samples == 0 || n_rows == 0) return hipSuccess;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return e;
    const int cus = prop.multiProcessorCount;
    int per_cu = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fused_kernel<1024>, 1024, 0);
    if (e != hipSuccess) return e;
    if (per_cu < 1 || cus < 1) return hipSuccess;
    const uint32_t capacity = (uint32_t)cus;  // one workgroup per CU, never more
    const uint64_t n_words = words_for(n_samples);
    const uint64_t P = (n_words + 1023) / 1024;
    if (P > capacity || P > 255) return hipSuccess;  // too many samples for one team
    const uint64_t n_batches = (n_rows + kRowsPerBatch - 1) / kRowsPerBatch;
    uint64_t Q = capacity / P;
    if (Q > n_batches) Q = n_batches;
    if (Q < 1) return hipSuccess;
    plan->threads = 1024;
    plan->P = (uint32_t)P;
    plan->Q = (uint32_t)Q;
    plan->n_batches = (uint32_t)n_batches;
    plan->part_team_stride = P * 1024 * 16;
    plan->ok = n_batches <= 0xffffffffull;
    return hipSuccess;
}

hipError_t launch_fused(hipStream_t st, const FusedPlan &plan, const uint32_t *d_codes,
                        uint64_t stride_words, uint64_t n_samples, uint64_t n_rows,
                        const nps_row_desc *d_desc, DevParams prm, unsigned long long *d_tally,
                        nps_locus_stat *d_stats, unsigned long long *d_nloci, double *d_part,
                        unsigned int *d_timeout) {
    FusedArgs a;
    a.codes = d_codes;
    a.stride_words = stride_words;
    a.n_rows = n_rows;
    a.n_samples = n_samples;
    a.n_words = (uint32_t)words_for(n_samples);
    a.n_batches = plan.n_batches;
    a.P = plan.P;
    a.Q = plan.Q;
    a.desc = d_desc;
    a.prm = prm;
    a.tally = d_tally;
    a.stats = d_stats;
    a.nloci = d_nloci;
    a.part = d_part;
    a.part_team_stride = plan.part_team_stride;
    a.timeout = d_timeout;
    void *args[] = {&a};
    // cooperative launch: the runtime rejects a grid that cannot be fully resident
    return hipLaunchCooperativeKernel((const void *)fused_kernel<1024>, dim3(plan.P, plan.Q),
                                      dim3(1024), args, 0, st);
}

hipError_t launch_fold(hipStream_t st, const double *d_part, uint32_t Q, uint64_t team_stride,
                       uint64_t n_samples, double *d_part0) {
    if (n_samples == 0 || Q == 0) return hipSuccess;
    (void)hipGetLastError();  // drop any stale sticky error: report this launch only
    hipLaunchKernelGGL(fold_kernel, dim3((uint32_t)((n_samples + 255) / 256)), dim3(256), 0, st,
                       d_part, Q, team_stride, n_samples, d_part0);
    return hipGetLastError();
}

}  // namespace nps
