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
#include <algorithm>

#include "nps_kernels.h"

namespace nps {

constexpr int kRowsPerBatch = 16;
constexpr int kFusedDefaultThreads = 1024;
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

// Row LUT from a complete tally word: the maxmis decision (nimpress.nim:565-571), the locus constant
// (:417-447) or the sample imputation value (:450-481).  One lane per row.
static __device__ __forceinline__ void row_lut(const FusedArgs &a, unsigned long long x, uint64_t row,
                                               bool write_stats, double (&v)[4], int &used) {
    const uint64_t nmiss = (x >> 28) & 0xFFFFFFFull, neff = x & 0xFFFFFFFull;
    const uint64_t ngen = a.n_samples - nmiss;
    const double beta = a.desc[row].beta, eaf = a.desc[row].eaf;
    const bool rie = a.desc[row].ref_is_effect != 0;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    int reason;
    used = 0;
    v[0] = v[1] = v[2] = v[3] = 0.0;
    const double missingrate = (double)nmiss / (double)a.n_samples;
    if (missingrate > a.prm.max_missing_rate) {
        reason = NPS_REASON_MAXMIS;
        if (a.prm.imp_locus != NPS_LOCUS_IGNORE) {
            const double c = a.prm.imp_locus == NPS_LOCUS_PS       ? eaf * 2.0
                             : a.prm.imp_locus == NPS_LOCUS_HOMREF ? (rie ? 2.0 : 0.0)
                                                                   : nan;
            used = 1;
            v[0] = v[1] = v[2] = v[3] = c * beta;
        }
    } else {
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
        v[0] = 0.0 * beta;
        v[1] = 1.0 * beta;
        v[2] = 2.0 * beta;
        v[3] = imp * beta;
    }
    if (write_stats) {
        nps_locus_stat s;
        s.ngenotyped = ngen;
        s.nmissing = nmiss;
        s.neffect = (double)neff;
        s.used = used;
        s.reason = reason;
        a.stats[row] = s;
    }
}

// Per workgroup and batch k (16 rows), two barriers:
//   S1  issue the 16 row loads of batch k+2 (register ring) ; wave 0 issues the poll of batch k
//   S2  partial tallies of batch k+1 (popcounts, DPP reduce-scatter, LDS adds)
//       wave 0: poll result of batch k (its latency hid under S2) -> 16 row LUTs -> LDS
//   --- barrier X
//   S4  16 lanes publish batch k+1 (one 64-bit agent-scope atomic per row);
//       every thread builds its share of the four 256-entry tables of batch k
//   --- barrier Y
//   S6  accumulate batch k from the ring registers: ds_read_b64 + v_add_f64 per 4 genotypes
template <int T>
__global__ __launch_bounds__(T, 4) void fused_kernel(const FusedArgs a) {
    static_assert(T % 64 == 0 && T >= 64 && T <= 1024, "workgroup size");
    __shared__ FusedLds lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t team = blockIdx.y;   // grid = (P slices, Q teams)
    const uint32_t slice = blockIdx.x;
    const uint32_t col = slice * T + tid;
    const bool active = col < a.n_words;
    const uint32_t voff = col * 4u;     // byte offset inside a row; rows are < 4 GB
    const uint32_t row_bytes = a.n_words * 4u;
    const uint64_t stride_bytes = a.stride_words * 4u;
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
    unsigned long long polled = 0;  // wave 0, lanes 0..15: tally word of "my" row of batch k

    // first row of local batch k (batches past the end get a row index >= n_rows)
    auto batch_row0 = [&](uint32_t k) -> uint64_t {
        return (uint64_t)(team + (uint64_t)k * a.Q) * kRowsPerBatch;
    };

    // One buffer descriptor per row (wave-uniform, in SGPRs): the hardware range check returns 0
    // for columns past the end of a row and for rows past the end of the matrix, so the loads need
    // no per-lane predication and no 64-bit per-lane addresses.
    auto load_batch = [&](uint32_t k, uint32_t(&dst)[kRowsPerBatch]) {
        const uint64_t row0 = batch_row0(k);
        const bool in = k < n_local && row0 < a.n_rows;
        const uint32_t nvalid = in ? (uint32_t)min((uint64_t)kRowsPerBatch, a.n_rows - row0) : 0u;
        const char *p = reinterpret_cast<const char *>(a.codes) + (in ? row0 : 0) * stride_bytes;
#pragma unroll
        for (int r = 0; r < kRowsPerBatch; ++r) {
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char *>(p), 0, (uint32_t)r < nvalid ? row_bytes : 0u, 0x00020000);
            dst[r] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, 0, 0);
            p += stride_bytes;
        }
    };

    // S2: partial tally of this workgroup's slice (batches past the end hold zeros)
    auto tally_local = [&](uint32_t k, const uint32_t(&src)[kRowsPerBatch]) {
        const int par = k & 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t a0 = tally_pack(src[4 * g + 0]), a1 = tally_pack(src[4 * g + 1]);
            const uint32_t a2 = tally_pack(src[4 * g + 2]), a3 = tally_pack(src[4 * g + 3]);
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
    };

    // S4 (after barrier X): lanes 0..15 publish the workgroup's partial tallies of batch k
    auto publish = [&](uint32_t k) {
        if (tid < kRowsPerBatch) {
            const int par = k & 1;
            const uint32_t v = lds.tally[par][tid];
            lds.tally[par][tid] = 0;  // next used two batches later, several barriers away
            const uint64_t row = batch_row0(k) + tid;
            if (k < n_local && row < a.n_rows) {
                const uint64_t t = v >> 16, m = v & 0xFFFFu;
                const uint64_t neff = t - 3 * m;
                const unsigned long long add = (1ull << 56) | (m << 28) | neff;
                __hip_atomic_fetch_add(&a.tally[row], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };

    // S1 (wave 0): start reading the tally words of batch k; the result is looked at after S2
    auto poll_issue = [&](uint32_t k) {
        if (tid < kRowsPerBatch) {
            const uint64_t row = batch_row0(k) + tid;
            polled = (k < n_local && row < a.n_rows)
                         ? __hip_atomic_load(&a.tally[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                         : 0ull;
        }
    };

    // end of S2 (wave 0): wait until all P slices of every row of batch k have arrived, then derive
    // the 16 row LUTs.  A bounded wait that expires sets the timeout word (the host reports
    // NPS_E_TIMEOUT and discards the scores); later waits in every workgroup then fall through at
    // once, so the grid always drains.
    auto poll_finish_lut = [&](uint32_t k) {
        if (tid < 64) {
            const uint64_t row = batch_row0(k) + lane;
            const bool valid = lane < kRowsPerBatch && k < n_local && row < a.n_rows;
            bool ok = !valid || (uint32_t)(polled >> 56) == a.P;
            uint32_t spins = 0;
            while (!__all(ok) && !timed_out) {
                __builtin_amdgcn_s_sleep(2);
                if (!ok) {
                    polled = __hip_atomic_load(&a.tally[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = (uint32_t)(polled >> 56) == a.P;
                }
                if ((++spins & 255u) == 0) {
                    const unsigned int t =
                        __hip_atomic_load(a.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (t != 0 || spins >= kSpinLimit) {
                        if (lane == 0)
                            __hip_atomic_store(a.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        timed_out = true;
                    }
                }
            }
            int used = 0;
            if (lane < kRowsPerBatch) {
                double v[4] = {0.0, 0.0, 0.0, 0.0};
                if (valid && ok) row_lut(a, polled, row, slice == 0 && a.stats != nullptr, v, used);
                *reinterpret_cast<double2 *>(&lds.lut[lane][0]) = make_double2(v[0], v[1]);
                *reinterpret_cast<double2 *>(&lds.lut[lane][2]) = make_double2(v[2], v[3]);
            }
            nloci_local += (uint32_t)__popcll(__ballot(used != 0));
        }
    };

    // S4: tables of the batch whose LUTs are in LDS; entry e of group g at e ^ (e >> 5)
    auto build_tables = [&]() {
#pragma unroll
        for (int i = tid; i < 4 * 256; i += T) {
            const int g = i >> 8, e = i & 255;
            const double v = ((lds.lut[4 * g][e & 3] + lds.lut[4 * g + 1][(e >> 2) & 3]) +
                              lds.lut[4 * g + 2][(e >> 4) & 3]) +
                             lds.lut[4 * g + 3][e >> 6];
            lds.table[g][e ^ (e >> 5)] = v;
        }
    };

    // S6
    auto accumulate = [&](const uint32_t(&src)[kRowsPerBatch]) {
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
                // 16-wave-per-CU kernel may use).
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(acc[4 * kk + q]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    auto step = [&](uint32_t k, uint32_t(&r_next2)[kRowsPerBatch], const uint32_t(&r_next1)[kRowsPerBatch],
                    const uint32_t(&r_cur)[kRowsPerBatch]) {
        load_batch(k + 2, r_next2);   // S1
        poll_issue(k);
        tally_local(k + 1, r_next1);  // S2
        poll_finish_lut(k);
        __syncthreads();              // X
        publish(k + 1);               // S4
        build_tables();
        __syncthreads();              // Y
        accumulate(r_cur);            // S6
    };

    // prologue: batch 0 loaded + tallied + published, batch 1 loaded
    load_batch(0, ring[0]);
    load_batch(1, ring[1]);
    tally_local(0, ring[0]);
    __syncthreads();
    publish(0);
    // batches past the end (k >= n_local, only in the last ring turn) run through unchanged: their
    // loads return zeros, nothing is published or waited for, LUTs are zero: +0.0 to every score
    for (uint32_t k = 0; k < n_local; k += 3) {
        step(k, ring[2], ring[1], ring[0]);
        step(k + 1, ring[0], ring[2], ring[1]);
        step(k + 2, ring[1], ring[0], ring[2]);
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
template <int T>
static hipError_t plan_for(int cus, uint64_t n_words, uint64_t n_batches, FusedPlan *plan) {
    int per_cu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fused_kernel<T>, T, 0);
    if (e != hipSuccess) return e;
    // the kernel needs 128 VGPRs: 16 waves per CU; never ask for more than that many workgroups
    per_cu = std::min(per_cu, 1024 / T);
    if (per_cu < 1) return hipSuccess;
    const uint64_t capacity = (uint64_t)cus * per_cu;
    const uint64_t P = (n_words + T - 1) / T;
    if (P > capacity || P > 255) return hipSuccess;  // 8-bit arrival count in the tally word
    uint64_t Q = std::min<uint64_t>(capacity / P, n_batches);
    if (Q < 1 || Q > 65535) return hipSuccess;
    plan->threads = T;
    plan->P = (uint32_t)P;
    plan->Q = (uint32_t)Q;
    plan->n_batches = (uint32_t)n_batches;
    plan->part_team_stride = P * T * 16;
    plan->ok = true;
    return hipSuccess;
}

hipError_t fused_plan(int device, uint64_t n_samples, uint64_t n_rows, int want_threads,
                      FusedPlan *plan) {
    *plan = FusedPlan{};
    if (n_samples == 0 || n_rows == 0 || n_samples >= (1ull << 27)) return hipSuccess;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return e;
    const int cus = prop.multiProcessorCount;
    if (cus < 1) return hipSuccess;
    const uint64_t n_words = words_for(n_samples);
    const uint64_t n_batches = (n_rows + kRowsPerBatch - 1) / kRowsPerBatch;
    if (n_batches > 0xfffffff0ull) return hipSuccess;
    // default: the smallest workgroup whose team still fits the 8-bit arrival count and the grid
    const int order_default[3] = {kFusedDefaultThreads, 512, 1024};
    const int order_want[3] = {want_threads, want_threads, want_threads};
    const int *order = want_threads ? order_want : order_default;
    for (int i = 0; i < 3 && !plan->ok; ++i) {
        switch (order[i]) {
        case 256: e = plan_for<256>(cus, n_words, n_batches, plan); break;
        case 512: e = plan_for<512>(cus, n_words, n_batches, plan); break;
        case 1024: e = plan_for<1024>(cus, n_words, n_batches, plan); break;
        default: return hipErrorInvalidValue;
        }
        if (e != hipSuccess) return e;
    }
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
    const void *fn = plan.threads == 256   ? (const void *)fused_kernel<256>
                     : plan.threads == 512 ? (const void *)fused_kernel<512>
                                           : (const void *)fused_kernel<1024>;
    // cooperative launch: the runtime rejects a grid that cannot be fully resident
    return hipLaunchCooperativeKernel(fn, dim3(plan.P, plan.Q), dim3(plan.threads), args, 0, st);
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
