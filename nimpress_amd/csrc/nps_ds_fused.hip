// nps_ds_fused.hip -- single-read kernel for a resident FORMAT/DS (float32 dosage) cohort.
//
// Same problem and same structure as nps_fused.hip (the 2-bit GT kernel): tallyAlleles
// (nimpress.nim:32-47) needs the whole row before any of its samples can be accumulated
// (:565-571, :470-477), the accumulation (:639-641) wants every sample's float64 partial on chip
// across all rows.  The two-pass DS kernels (nps_ds.hip) read the 4 B/genotype matrix twice; this
// kernel reads it once:
//
//   * cooperative grid = Q teams x P workgroups (one per CU).  Workgroup (q,p) owns 7 680 samples
//     (8 per data thread: two float4 columns 256 samples apart, so every wave instruction reads
//     1 KiB contiguously) and the row batches q, q+Q, ... (2 rows per batch).
//   * 15 data waves: 6-deep register ring of batches; per phase they accumulate batch k in the
//     reference's own order and operations (row after row, float64 `dosage * beta` then `+=`,
//     not fused), tally batch k+5 (NaN count by wave ballot, dosage sum by a fixed-shape tree) and
//     refill the ring.
//   * control wave: combines the waves' partial tallies in fixed order and hands the slice's tally over with
//     two 64-bit agent-scope atomic adds on the row's pair of words, each carrying its own arrival count:
//     word 0 = arrivals<<56 | high part of the dosage sum<<28 | nmissing, word 1 = arrivals<<56 | low part.
//     The slice's dosage sum travels as a FIXED-POINT integer (2^F, F from the cohort size): integer adds
//     commute, so the total does not depend on the order of arrival (bit-reproducible), nobody stores, drains
//     or re-reads per-slice partial sums (rounds 1-3: an sc1 store, a drained queue, then the arrival, and P
//     loads + a tree sum on the other side: 11 % of the pass).  Two phases later the control wave looks at the
//     pair until both counts say P and derives the row's parameters.  One workgroup barrier per batch.
//     Range: FORMAT/DS values are dosages, 0 <= DS <= 2: the bound is what keeps the fields of the words apart.  It is
//     checked where a cohort is filled (nps_cohort_upload: ds_range_kernel; the generator clips), never in this loop
//     (a check on the data waves cost 15 % of the pass): a cohort that holds a value outside is scored by the
//     two-pass kernels.
//
// Semantics are the build-defined DS extension of the oracle (ref_raw_dosages_ds): NaN = missing,
// effect allele == REF -> dosage = 2 - DS.  The row's dosage sum is formed as sum(DS) and turned into
// 2*ngenotyped - sum(DS) for such rows (the two-pass kernel adds 2 - DS per sample); both differ
// from the oracle's sequential sum in the last bits only (test bar 1e-9 relative).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "nps_kernels.h"

namespace nps {

constexpr int kDsRows = 2;     // rows per batch (float32 rows)
constexpr int kDsRows16 = 2;   // ... of 16-bit rows (four would halve the hand-overs per row; the ring of 6 x 4 rows does not fit 128 VGPRs: 292 bytes spilled)
constexpr int kDsRowsMax = 4;
constexpr int kDsPerThread = 8;
constexpr int kDsRing = 6;     // batches held per data thread: one accumulated, four tallied and waiting for the hand-over, one on its way
// workgroup size T: wave 0 is the control wave, T/64 - 1 data waves of 512 samples each
static constexpr uint32_t ds_slice_samples(int threads) { return (uint32_t)(threads - 64) * kDsPerThread; }
constexpr uint32_t kDsSpinLimit = 1u << 20;

struct DsFusedArgs {
    const void *ds;         // float32 rows, or (H) uint16 rows of NPS_FMT_DS16
    uint64_t stride_bytes;  // bytes from one row to the next
    uint64_t n_rows;
    uint64_t n_samples;
    uint32_t n_batches;
    uint32_t P, Q;
    const nps_row_desc *desc;
    DevParams prm;
    int64_t t_maxmis;  // the largest nmissing for which nmissing / N > --maxmis is false (-1: none), found with that division
    unsigned long long *tally;  // [n_rows][2], zeroed: arrivals << 56 | sum_hi << 28 | nmissing ; arrivals << 56 | sum_lo
    double scale, inv_scale;    // 2^F, 2^-F: the slices' dosage sums travel as round(sum * 2^F)
    nps_locus_stat *stats;
    unsigned long long *nloci;
    double *part;  // [Q][part_team_stride]
    uint64_t part_team_stride;
    unsigned int *timeout;
};

struct DsRowLds {
    double beta, imp;  // imp: value of a missing sample (mode 1) or of every sample (mode 2)
    int32_t mode;      // 0 dropped (beta = imp = 0), 1 genotyped, 2 locus constant
    int32_t flip;      // dosage = 2 - DS
};
struct __attribute__((aligned(16))) DsFusedLds {
    double wsum[2][kDsRowsMax][16];
    uint32_t wcnt[2][kDsRowsMax][16];
    DsRowLds rowp[2][kDsRowsMax];
};

// wave sum by DPP (row_shr 1,2,4,8, then row_bcast15 / row_bcast31): fixed order, total in lane 63.
// Two v_mov_b32_dpp + one v_add_f64 per stage, ~100 cycles of dependent latency per row instead of
// the ~1 500 of a ds_bpermute tree.
template <int CTRL, int ROW_MASK>
static __device__ __forceinline__ double dpp_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
static __device__ __forceinline__ double wave_dpp_sum(double v) {
    v += dpp_f64<0x111, 0xF>(v);  // row_shr:1 (lanes without a source add +0.0)
    v += dpp_f64<0x112, 0xF>(v);  // row_shr:2
    v += dpp_f64<0x114, 0xF>(v);  // row_shr:4
    v += dpp_f64<0x118, 0xF>(v);  // row_shr:8   -> lane 15 of every row holds the row's sum
    v += dpp_f64<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
    v += dpp_f64<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
    return v;                     // lane 63
}
static __device__ __forceinline__ double uniform_f64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readfirstlane((int)b);
    const int hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// diagnostics builds (-DNPS_DS_TIMERS): cycles per phase of the control wave and of data wave 1 of one workgroup
#ifdef NPS_DS_TIMERS
__device__ unsigned long long g_ds_timers[2][8];
#define DST(i) do { if (timing) { const unsigned long long now_ = __builtin_readcyclecounter(); tph[i] += now_ - tlast; tlast = now_; } } while (0)
#else
#define DST(i) do { } while (0)
#endif
// H: a NPS_FMT_DS16 cohort -- 2 bytes per genotype: k = dosage x 10^4 (0 .. 20 000), 0xFFFF = missing.  A thread holds
// eight NEIGHBOURING samples (one 16-byte load per row) and turns k back into the float32 a decimal parser gives for the
// text -- ds16_value(k), nps_kernels.h -- wherever the float32
// kernel reads a value: the arithmetic after that is the float32 kernel's, operation for operation.
template <int T, bool H>
__global__ __launch_bounds__(T, 4) void ds_fused_kernel(const DsFusedArgs a) {
    constexpr uint32_t kDsSliceSamples = ds_slice_samples(T);
    constexpr int R = H ? kDsRows16 : kDsRows, D = kDsRing;
    __shared__ DsFusedLds lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const uint32_t team = blockIdx.y, slice = blockIdx.x;
    const uint32_t n_local = a.n_batches > team ? (a.n_batches - team + a.Q - 1) / a.Q : 0;
    const uint32_t n_steps = (n_local + D - 1) / D * D;

#ifdef NPS_DS_TIMERS
    const bool timing = slice == a.P / 2 && team == 0;
    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    auto batch_row0 = [&](uint32_t k) -> uint64_t { return (uint64_t)(team + (uint64_t)k * a.Q) * R; };

    // Barrier #j closes the phase in which the data waves tallied batch j.
    //   data waves, phase k (between #(k+4) and #(k+5)): accumulate batch k with rowp[k&1], tally
    //       batch k+5, refill the ring slot of batch k with batch k+6
    //   control wave, same phase: publish batch k+4; look at batch k+2 (published by every slice two
    //       phases ago: a look that finds a word incomplete costs the phase a second round trip, and
    //       with one phase of distance that happened often enough to cost 17 % of the pass);
    //       parameters of batch k+1 -> rowp[(k+1)&1]
    if (wave == 0) {
        // ------------------------------------------------------------------ control wave
        uint32_t nloci_local = 0;
        bool timed_out = false;

        // One phase of the control wave costs ONE memory round trip: the loads of the partial sums
        // of batch k+1 (whose arrival word matched a phase ago), the poll of batch k+2 and the
        // publication of batch k+3 are issued back to back and drained by one s_waitcnt.
        struct Polled {  // lanes < R: one row each
            unsigned long long x, y;  // the row's pair of words
            double beta, eaf;
            int rflags;
            bool valid, ok;
        };

        auto poll_issue = [&](uint32_t kt) -> Polled {
            Polled q;
            const uint64_t row = batch_row0(kt) + lane;
            q.valid = lane < R && kt < n_local && row < a.n_rows;
            q.x = q.y = 0;
            q.beta = q.eaf = 0.0;
            q.rflags = 0;
            if (q.valid) {
                q.x = __hip_atomic_load(&a.tally[2 * row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                q.y = __hip_atomic_load(&a.tally[2 * row + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                q.beta = a.desc[row].beta;
                q.eaf = a.desc[row].eaf;
                q.rflags = a.desc[row].ref_is_effect;
            }
            q.ok = false;
            return q;
        };
        auto poll_finish = [&](uint32_t kt, Polled &q) {  // spins until all P slices have arrived
            const uint64_t row = batch_row0(kt) + lane;
            q.ok = !q.valid || ((uint32_t)(q.x >> 56) == a.P && (uint32_t)(q.y >> 56) == a.P);
            uint32_t spins = 0;
            while (!__all(q.ok) && !timed_out) {
                __builtin_amdgcn_s_sleep(1);
                if (!q.ok) {
                    q.x = __hip_atomic_load(&a.tally[2 * row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    q.y = __hip_atomic_load(&a.tally[2 * row + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    q.ok = (uint32_t)(q.x >> 56) == a.P && (uint32_t)(q.y >> 56) == a.P;
                }
                if ((++spins & 255u) == 0) {
                    const unsigned int t =
                        __hip_atomic_load(a.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (t != 0 || spins >= kDsSpinLimit) {
                        if (lane == 0)
                            __hip_atomic_store(a.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        timed_out = true;
                    }
                }
            }
        };

        // partial tallies of batch k (its LDS sums are complete): two atomic adds, nothing to drain, nothing stored
        auto publish = [&](uint32_t k) {
            double s = 0.0;
            uint32_t cnt = 0;
            const uint64_t row = batch_row0(k) + lane;
            const bool valid = lane < R && k < n_local && row < a.n_rows;
            if (lane < R) {
                const int par = k & 1;
#pragma unroll
                for (int w = 1; w < T / 64; ++w) {  // fixed order
                    s += lds.wsum[par][lane][w];
                    cnt += lds.wcnt[par][lane][w];
                }
            }
            if (valid) {
                const double x = s * a.scale;
                // (a slice holds < 2^14 of dosage, F <= 46; cohorts with values outside [0, 2] never get here)
                const unsigned long long S = x >= 0.0 && x < 9.0e18 ? (unsigned long long)__double2ll_rn(x) : 0ull;
                __hip_atomic_fetch_add(&a.tally[2 * row], (1ull << 56) | ((S >> 32) << 28) | (unsigned long long)cnt,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(&a.tally[2 * row + 1], (1ull << 56) | (S & 0xffffffffull), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
            }
        };

        // row parameters of batch kt from its complete pair of words
        auto params = [&](uint32_t kt, const Polled &q) {
            const unsigned long long tot = (((q.x >> 28) & 0xfffffffull) << 32) + (q.y & ((1ull << 56) - 1));
            const double mysum = (double)tot * a.inv_scale;
            int used = 0;
            if (lane < R) {
                const uint64_t row = batch_row0(kt) + lane;
                DsRowLds rp;
                rp.beta = 0.0;
                rp.imp = 0.0;
                rp.mode = 0;
                rp.flip = q.rflags == 1;  // bit 1 set: the row already counts the effect allele
                if (q.valid && q.ok) {
                    const double beta = q.beta, eaf = q.eaf;
                    const bool rie = (q.rflags & 1) != 0;  // homref imputation value
                    const uint64_t nmiss = q.x & 0xfffffffull;
                    const uint64_t ngen = a.n_samples - nmiss;
                    const double neff = rp.flip ? 2.0 * (double)ngen - mysum : mysum;
                    const double nan = __longlong_as_double(0x7ff8000000000000ll);
                    int reason;
                    // nim:565-571: nmissing / N > --maxmis, as an integer comparison (the host found the threshold with the
                    // division itself; a float64 division here is ~400 cycles of the control wave's phase)
                    if ((int64_t)nmiss > a.t_maxmis) {
                        reason = NPS_REASON_MAXMIS;
                        if (a.prm.imp_locus != NPS_LOCUS_IGNORE) {
                            rp.imp = a.prm.imp_locus == NPS_LOCUS_PS       ? eaf * 2.0
                                     : a.prm.imp_locus == NPS_LOCUS_HOMREF ? (rie ? 2.0 : 0.0)
                                                                           : nan;
                            rp.mode = 2;  // every sample gets the locus constant
                            used = 1;
                        }
                    } else {  // nim:450-481
                        reason = NPS_REASON_GENOTYPED;
                        used = 1;
                        rp.mode = 1;
                        switch (a.prm.imp_sample) {
                        case NPS_SAMPLE_PS: rp.imp = eaf * 2.0; break;
                        case NPS_SAMPLE_HOMREF: rp.imp = rie ? 2.0 : 0.0; break;
                        case NPS_SAMPLE_FAIL: rp.imp = nan; break;
                        default:
                            if ((double)ngen >= a.prm.min_cs)
                                rp.imp = neff / (double)ngen;
                            else
                                rp.imp = a.prm.imp_sample == NPS_SAMPLE_INT_PS ? eaf * 2.0 : nan;
                            break;
                        }
                    }
                    if (rp.mode != 0) rp.beta = beta;  // a dropped row adds imp * beta = 0 * 0
                    if (slice == 0 && a.stats != nullptr) {
                        nps_locus_stat s;
                        s.ngenotyped = ngen;
                        s.nmissing = nmiss;
                        s.neffect = neff;
                        s.used = used;
                        s.reason = reason;
                        a.stats[row] = s;
                    }
                }
                lds.rowp[kt & 1][lane] = rp;
            }
            nloci_local += (uint32_t)__popcll(__ballot(used != 0));
        };

        __syncthreads();  // #0
        publish(0);
        __syncthreads();  // #1
        publish(1);
        __syncthreads();  // #2
        publish(2);
        __syncthreads();  // #3
        publish(3);
        Polled cur = poll_issue(0);
        poll_finish(0, cur);
        params(0, cur);
        cur = poll_issue(1);
        poll_finish(1, cur);
        __syncthreads();  // #4
        for (uint32_t k = 0; k < n_steps; ++k) {
            DST(7);
            Polled nxt = poll_issue(k + 2);   // batch k+2: published by every slice TWO phases ago: complete unless a slice lags
            publish(k + 4);
            DST(0);
            poll_finish(k + 2, nxt);
            DST(1);
            params(k + 1, cur);
            DST(2);
            cur = nxt;
            __syncthreads();  // #(k+5)
            DST(3);
        }
#ifdef NPS_DS_TIMERS
        if (timing && lane == 0)
            for (int i = 0; i < 8; ++i) g_ds_timers[0][i] = tph[i];
#endif
        if (slice == 0 && lane == 0 && nloci_local) atomicAdd(a.nloci, (unsigned long long)nloci_local);
        return;
    }

    // ---------------------------------------------------------------------- data waves
    // samples of this thread: s0 + 0..3 and s0 + 256 + 0..3 (H: s0 + 0..7)
    const uint32_t s0 = slice * kDsSliceSamples + (uint32_t)(wave - 1) * 512u + (uint32_t)lane * (H ? 8u : 4u);
    const uint32_t voff = s0 * (H ? 2u : 4u);
    // clamp so that the descriptor range (bytes of one row) stays below 4 GiB
    // (the range is checked dword by dword: an odd number of 16-bit codes is rounded up -- the row is zero padded)
    const uint32_t row_bytes = (uint32_t)min(H ? (a.n_samples * 2ull + 3ull) & ~3ull : a.n_samples * 4ull, 0xfffffff0ull);
    const uint64_t stride_bytes = a.stride_bytes;

    double acc[kDsPerThread];
#pragma unroll
    for (int s = 0; s < kDsPerThread; ++s) acc[s] = 0.0;
    constexpr int kRW = H ? 4 : 8;  // ring words per row and thread
    // (the float32 kernel sits at exactly 128 VGPRs without a spill: its ring stays an array of floats and its expressions
    //  stay what they were before the 16-bit variant existed -- kept as uint32 with casts at the uses it spilled 80 bytes
    //  per lane and took 61 ms instead of 38)
    typedef typename std::conditional<H, uint32_t, float>::type RingE;
    typedef RingE RingT[R * kRW];
    RingT ring[D];

    // row r of batch k -> dst[kRW r ..]; rows past the end of the matrix read as zeros (range 0: dosage 0, not missing)
    auto load_row = [&](uint32_t k, int r, RingT &dst) {
        const uint64_t row = batch_row0(k) + r;
        const bool in = k < n_local && row < a.n_rows;
        const char *p = reinterpret_cast<const char *>(a.ds) + (in ? row : 0) * stride_bytes;
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(p), 0, in ? row_bytes : 0u, 0x00020000);
        const auto qa = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 2);
        if constexpr (H) {
#pragma unroll
            for (int s = 0; s < 4; ++s) dst[r * kRW + s] = qa[s];
        } else {
            const auto qb = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff + 1024u, 0, 2);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                dst[r * 8 + s] = __uint_as_float(qa[s]);
                dst[r * 8 + 4 + s] = __uint_as_float(qb[s]);
            }
        }
    };
    auto load_batch = [&](uint32_t k, RingT &dst) {
#pragma unroll
        for (int r = 0; r < R; ++r) load_row(k, r, dst);
    };
    // (H) elements 2p, 2p + 1 of row r of a ring slot -- the two halves of one word: whether they are missing, and their
    // float32 values (meaningless if they are): ds16_value of nps_kernels.h on a pair (v_pk_mul_f32 / v_pk_fma_f32)
    typedef float f2 __attribute__((ext_vector_type(2)));
    auto pair = [&](const RingT &src, int r, int p, f2 &v, bool &nan0, bool &nan1) {
        const uint32_t w = (uint32_t)src[r * kRW + p];
        const uint32_t k0 = w & 0xffffu, k1 = w >> 16;
        nan0 = k0 == 0xffffu;
        nan1 = k1 == 0xffffu;
        const f2 a = {(float)k0, (float)k1};
        const f2 c1 = {9.999999747378752e-05f, 9.999999747378752e-05f}, c2 = {2.5262125290942405e-12f, 2.5262125290942405e-12f};
        v = __builtin_elementwise_fma(a, c1, a * c2);
    };

    // partial tally of a batch: NaN count of the wave by ballot (SALU), dosage sum by a fixed tree
    auto tally = [&](uint32_t k, const RingT &src) {
        const int par = k & 1;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            uint32_t cnt = 0;
            double s = 0.0;
            if constexpr (H) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    f2 v;
                    bool n0, n1;
                    pair(src, r, p, v, n0, n1);
                    cnt += (uint32_t)__popcll(__ballot(n0)) + (uint32_t)__popcll(__ballot(n1));
                    s += (double)(n0 ? 0.0f : v[0]);
                    s += (double)(n1 ? 0.0f : v[1]);
                    if (p & 1) {
                        asm volatile("" : "+v"(s));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < kDsPerThread; ++e) {
                    const float v = src[r * 8 + e];
                    const bool nan = v != v;
                    cnt += (uint32_t)__popcll(__ballot(nan));
                    s += (double)(nan ? 0.0f : v);
                    if ((e & 3) == 3) {
                        asm volatile("" : "+v"(s));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            s = wave_dpp_sum(s);
            if (lane == 63) {
                lds.wsum[par][r][wave] = s;
                lds.wcnt[par][r][wave] = cnt;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // score += dosage * beta, row after row (nimpress.nim:639-641).  Branch-free: dosage =
    // c0 + sgn * DS with (c0, sgn) = (0, 1) or (2, -1) -- one rounding, equal to DS and 2 - DS -- and
    // the row's `all` flag (locus constant, or dropped row with beta = imp = 0) ORed into the NaN mask.
    auto accumulate = [&](uint32_t k, const RingT &cur) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const DsRowLds &rp = lds.rowp[k & 1][r];
            const double beta = uniform_f64(rp.beta), imp = uniform_f64(rp.imp);
            const bool all = __builtin_amdgcn_readfirstlane(rp.mode) != 1;
            const bool flip = __builtin_amdgcn_readfirstlane(rp.flip) != 0;
            const double c0 = flip ? 2.0 : 0.0, sgn = flip ? -1.0 : 1.0;
            if constexpr (H) {
                // (a wave-uniform branch per row instead of the fma -- seven rows in ten need no dosage instruction -- and the
                //  decode on pairs changed nothing: 45 ms either way; see DESIGN.md 4.4 for what bounds this variant)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    f2 v;
                    bool n0, n1;
                    pair(cur, r, p, v, n0, n1);
                    const double d0 = n0 || all ? imp : __fma_rn((double)v[0], sgn, c0);
                    const double d1 = n1 || all ? imp : __fma_rn((double)v[1], sgn, c0);
                    acc[2 * p] += d0 * beta;
                    acc[2 * p + 1] += d1 * beta;
                    asm volatile("" : "+v"(acc[2 * p]), "+v"(acc[2 * p + 1]));
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int e = 0; e < kDsPerThread; ++e) {
                    const float v = cur[r * 8 + e];
                    const double d = (v != v) || all ? imp : __fma_rn((double)v, sgn, c0);
                    acc[e] += d * beta;
                    if (e & 1) {  // two elements at a time: keeps the converted values out of the ring's way
                        asm volatile("" : "+v"(acc[e - 1]), "+v"(acc[e]));
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    };

    auto step = [&](uint32_t k, RingT &r_cur, const RingT &r_tal) {
        DST(7);
        accumulate(k, r_cur);
        DST(0);
        load_batch(k + 6, r_cur);  // in flight during the tally below and the next accumulation
        __builtin_amdgcn_sched_barrier(0);
        DST(1);
        tally(k + 5, r_tal);
        __builtin_amdgcn_sched_barrier(0);
        DST(2);
        __syncthreads();  // #(k+5)
        DST(3);
    };

    load_batch(0, ring[0]);
    load_batch(1, ring[1]);
    load_batch(2, ring[2]);
    load_batch(3, ring[3]);
    load_batch(4, ring[4]);
    load_batch(5, ring[5]);
    tally(0, ring[0]);
    __syncthreads();  // #0
    tally(1, ring[1]);
    __syncthreads();  // #1
    tally(2, ring[2]);
    __syncthreads();  // #2
    tally(3, ring[3]);
    __syncthreads();  // #3
    tally(4, ring[4]);
    __syncthreads();  // #4
    for (uint32_t k = 0; k < n_steps; k += D) {
        step(k, ring[0], ring[5]);
        step(k + 1, ring[1], ring[0]);
        step(k + 2, ring[2], ring[1]);
        step(k + 3, ring[3], ring[2]);
        step(k + 4, ring[4], ring[3]);
        step(k + 5, ring[5], ring[4]);
    }
#ifdef NPS_DS_TIMERS
    if (timing && tid == 64)
        for (int i = 0; i < 8; ++i) g_ds_timers[1][i] = tph[i];
#endif
    double *dst = a.part + (uint64_t)team * a.part_team_stride;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint64_t s = (uint64_t)s0 + (H ? 4u : 256u) * h;
        if (s < a.n_samples) {  // part_team_stride is padded: whole groups of 4 can be written
            *reinterpret_cast<double2 *>(dst + s) = make_double2(acc[4 * h], acc[4 * h + 1]);
            *reinterpret_cast<double2 *>(dst + s + 2) = make_double2(acc[4 * h + 2], acc[4 * h + 3]);
        }
    }
}

// ---- host side ------------------------------------------------------------------------------
template <int T>
static hipError_t ds_plan_for(int cus, uint64_t n_samples, uint64_t n_batches, int max_q, FusedPlan *plan) {
    *plan = FusedPlan{};
    int per_cu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ds_fused_kernel<T, false>, T, 0);
    if (e != hipSuccess) return e;
    if (per_cu < 1) return hipSuccess;
    const uint64_t slice = ds_slice_samples(T);
    const uint64_t P = (n_samples + slice - 1) / slice;
    if (P > (uint64_t)cus || P > 255) return hipSuccess;
    uint64_t Q = std::min<uint64_t>((uint64_t)cus / P, n_batches);
    if (max_q > 0) Q = std::min<uint64_t>(Q, (uint64_t)max_q);  // diagnostics: fewer CUs in use
    if (Q < 1 || Q > 65535) return hipSuccess;
    plan->threads = T;
    plan->P = (uint32_t)P;
    plan->Q = (uint32_t)Q;
    plan->n_batches = (uint32_t)n_batches;
    plan->part_team_stride = P * slice;
    plan->ok = true;
    return hipSuccess;
}

hipError_t ds_fused_plan(int device, uint64_t n_samples, uint64_t n_rows, int want, int max_q,
                         FusedPlan *plan, int elem_bytes) {
    *plan = FusedPlan{};
    if (n_samples == 0 || n_rows == 0 || n_samples >= (1ull << 27)) return hipSuccess;  // 28-bit tally fields
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return e;
    const int cus = prop.multiProcessorCount;
    if (cus < 1) return hipSuccess;
    const uint64_t rows_per_batch = elem_bytes == 2 ? kDsRows16 : kDsRows;
    const uint64_t n_batches = (n_rows + rows_per_batch - 1) / rows_per_batch;
    if (n_batches > 0xfffffff0ull) return hipSuccess;
    // as for the GT kernel: the most teams first, then the smallest workgroup that still gives that many
    const int candidates[3] = {1024, 960, 896};
    for (int t : candidates) {
        if (want && want != t) continue;
        FusedPlan p;
        e = t == 1024 ? ds_plan_for<1024>(cus, n_samples, n_batches, max_q, &p)
            : t == 960 ? ds_plan_for<960>(cus, n_samples, n_batches, max_q, &p)
                       : ds_plan_for<896>(cus, n_samples, n_batches, max_q, &p);
        if (e != hipSuccess) return e;
        if (p.ok && (!plan->ok || p.Q > plan->Q || (p.Q == plan->Q && p.threads < plan->threads))) *plan = p;
    }
    return hipSuccess;
}

hipError_t launch_ds_fused(hipStream_t st, const FusedPlan &plan, const void *d_ds, uint64_t stride_bytes, int elem_bytes,
                           uint64_t n_samples, uint64_t n_rows, const nps_row_desc *d_desc,
                           DevParams prm, int64_t t_maxmis, unsigned long long *d_tally,
                           nps_locus_stat *d_stats, unsigned long long *d_nloci, double *d_part,
                           unsigned int *d_timeout) {
    if (elem_bytes != 4 && elem_bytes != 2) return hipErrorInvalidValue;
    const bool half = elem_bytes == 2;
    DsFusedArgs a;
    a.ds = d_ds;
    a.stride_bytes = stride_bytes;
    a.n_rows = n_rows;
    a.n_samples = n_samples;
    a.n_batches = plan.n_batches;
    a.P = plan.P;
    a.Q = plan.Q;
    a.desc = d_desc;
    a.prm = prm;
    a.t_maxmis = t_maxmis;
    a.tally = d_tally;
    // total < 2 N 2^F must leave its high part (>> 32) inside 28 bits, a slice's sum (< 2^14) inside 63
    int lg = 0;
    while ((1ull << lg) < n_samples) ++lg;
    const int F = std::min(46, 58 - lg);
    a.scale = std::ldexp(1.0, F);
    a.inv_scale = std::ldexp(1.0, -F);
    a.stats = d_stats;
    a.nloci = d_nloci;
    a.part = d_part;
    a.part_team_stride = plan.part_team_stride;
    a.timeout = d_timeout;
    void *args[] = {&a};
    // cooperative launch: the runtime rejects a grid that cannot be fully resident
    const void *fn = half ? (plan.threads == 1024  ? (const void *)ds_fused_kernel<1024, true>
                             : plan.threads == 960 ? (const void *)ds_fused_kernel<960, true>
                                                   : (const void *)ds_fused_kernel<896, true>)
                          : (plan.threads == 1024  ? (const void *)ds_fused_kernel<1024, false>
                             : plan.threads == 960 ? (const void *)ds_fused_kernel<960, false>
                                                   : (const void *)ds_fused_kernel<896, false>);
#ifdef NPS_DS_TIMERS
    {
        hipError_t e = hipLaunchCooperativeKernel(fn, dim3(plan.P, plan.Q), dim3(plan.threads), args, 0, st);
        if (e != hipSuccess) return e;
        (void)hipStreamSynchronize(st);
        unsigned long long h[2][8];
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ds_timers), sizeof(h));
        const double steps = (double)((plan.n_batches + plan.Q - 1) / plan.Q);
        fprintf(stderr, "ds timers control wave (cycles per phase): issue+publish %.0f  poll-finish %.0f  params %.0f  barrier %.0f  loop %.0f\n",
                h[0][0] / steps, h[0][1] / steps, h[0][2] / steps, h[0][3] / steps, h[0][7] / steps);
        fprintf(stderr, "ds timers data wave 1 (cycles per phase): accumulate %.0f  load-issue %.0f  tally %.0f  barrier %.0f  loop %.0f\n",
                h[1][0] / steps, h[1][1] / steps, h[1][2] / steps, h[1][3] / steps, h[1][7] / steps);
        return hipSuccess;
    }
#endif
    return hipLaunchCooperativeKernel(fn, dim3(plan.P, plan.Q), dim3(plan.threads), args, 0, st);
}

}  // namespace nps
