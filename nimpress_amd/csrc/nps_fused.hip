// nps_fused.hip -- the fused, single-read kernel for a resident 2-bit cohort (gfx950, wave64).
//
// Problem: tallyAlleles (nimpress.nim:32-47) needs a WHOLE ROW (all samples) before any sample of
// that row can be accumulated (the maxmis decision :565-571 and the internal imputation value
// :470-477 depend on it), while the accumulation (nimpress.nim:639-641) wants every sample to
// keep its float64 partial score on chip across ALL rows.  A two-kernel design reads the matrix
// twice.  This kernel reads it once:
//
//   * grid = Q teams x P workgroups, all co-resident (cooperative launch, one workgroup per CU).
//     Workgroup (q,p) owns the sample slice [p*960*16, (p+1)*960*16) -- one 16-sample word column
//     and 16 float64 accumulators per data thread -- and the row batches b = q, q+Q, q+2Q, ...
//     (16 rows = 4 groups per batch).
//   * data waves (15 of 16), per batch: 16 coalesced row loads through per-row buffer descriptors
//     into a 4-deep REGISTER ring; partial tallies of the batch three ahead (popcount + DPP
//     reduce-scatter + LDS add) interleaved with the accumulation of the current batch: one
//     ds_read_b64 + one v_add_f64 per FOUR genotypes from 256-entry float64 tables in LDS.
//   * control wave (wave 0, carries no samples), concurrently: publishes the workgroup's partial
//     tallies with one 64-bit agent-scope atomic per row (arrivals<<56 | nmiss<<28 | neffect), polls
//     the tally words of the batch published one phase earlier until all P slices have arrived,
//     derives the 16 row LUTs and writes the next batch's four tables into the other LDS buffer.
//   * one workgroup barrier per batch.  Inter-workgroup traffic is 8-byte agent-scope atomics on
//     both sides (an `sc1` form measured valid on gfx950, MI355X_MICROARCH.md "Valid forms"); every
//     spin is bounded and sets a timeout word instead of hanging.
//
// HBM traffic = the matrix once (+ 8 B of atomics per row per slice, + row descriptors).
// The kernel sits at the knee of three limits (PMC: VALU busy 74 %, LDS busy 68 %, HBM 4.96 TB/s in
// a tally-only build), and at the register limit of 4 waves/SIMD: keep `ScratchSize` at 0 -- a
// 32-byte spill costs 28 % (diagnostics are therefore template parameters, not run-time flags).
#include <algorithm>
#include <cstdlib>

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
    unsigned long long *telemetry;  // diagnostics builds only: 8 counters of the control wave
};

// control-wave telemetry (cycles spent polling / in the chain / at the barrier) exists in diagnostics
// builds only (-DNPS_DIAGNOSTICS, tools/mkexp.sh); the release kernel carries none of it
#ifdef NPS_DIAGNOSTICS
#define NPS_TEL(x) x
#else
#define NPS_TEL(x)
#endif

// popcounts of one device word, packed popc(w) << 16 | missing.  With the codes 00/01/11 = dosage
// 0/1/2 and 10 = missing, popc(w) = effect alleles + missing samples: five VALU ops per row word.
static __device__ __forceinline__ uint32_t tally_pack(uint32_t w) {
    const uint32_t t = __popc(w);
    const uint32_t m = __popc((w >> 4) & ~w & 0x0F0F0F0Fu);
    return (t << 16) | m;
}

// (table_index() -- which code bit lands in which LDS bank-select bit -- lives in nps_kernels.h: the four
// LOW code bits in the low nibble, the high code bits, or their parity, above them.)

// (a & m) | (b & ~m) as ONE instruction.  Written as asm because hipcc (ROCm 7.2) otherwise breaks
// the three merge stages below into separate v_and / v_bitop3 ops (~70 instead of 24 per group), and
// this kernel is bound by VALU issue.  m is wave-uniform (SGPR).
static __device__ __forceinline__ uint32_t bfi(uint32_t m, uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(d) : "s"(m), "v"(a), "v"(b));
    return d;
}

// 4 rows x 16 samples of 2-bit codes -> 16 table indices in 16 VALU ops (2 stages x 4 merges, each a
// shift + v_bfi): x[q] byte k = table_index of sample 4k+q.  The device words keep the low and the
// high code bits of four samples in separate nibbles of a byte (nps_kernels.h), so exchanging the two
// sample-in-byte bits with the two row bits IS the whole transposition.
static __device__ __forceinline__ void transpose_fold_4x16(uint32_t w0, uint32_t w1, uint32_t w2,
                                                           uint32_t w3, uint32_t (&x)[4]) {
    const uint32_t m1 = 0x55555555u, m2 = 0x33333333u;
    // stage 1: samples 4k, 4k+2 (a0, b0) and 4k+1, 4k+3 (a1, b1) of the row pairs (0,1) and (2,3)
    const uint32_t a0 = bfi(m1, w0, w1 << 1);
    const uint32_t a1 = bfi(m1, w0 >> 1, w1);
    const uint32_t b0 = bfi(m1, w2, w3 << 1);
    const uint32_t b1 = bfi(m1, w2 >> 1, w3);
    // stage 2: nibbles = four rows, one code bit, one sample
    x[0] = bfi(m2, a0, b0 << 2);   // samples 0,4,8,12
    x[2] = bfi(m2, a0 >> 2, b0);   // samples 2,6,10,14
    x[1] = bfi(m2, a1, b1 << 2);   // samples 1,5,9,13
    x[3] = bfi(m2, a1 >> 2, b1);   // samples 3,7,11,15
}

// Row LUT from a complete tally word: the maxmis decision (nimpress.nim:565-571), the locus constant
// (:417-447) or the sample imputation value (:450-481).  One lane per row.
static __device__ __forceinline__ void row_lut(const FusedArgs &a, unsigned long long x, uint64_t row,
                                               double beta, double eaf, bool rie, bool write_stats,
                                               double (&v)[4], int &used) {
    const uint64_t nmiss = (x >> 28) & 0xFFFFFFFull, neff = x & 0xFFFFFFFull;
    const uint64_t ngen = a.n_samples - nmiss;
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
        v[0] = 0.0 * beta;  // indexed by CODE: 0, 1 = dosage ; 2 = missing ; 3 = dosage 2
        v[1] = 1.0 * beta;
        v[2] = imp * beta;
        v[3] = 2.0 * beta;
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

// ---------------------------------------------------------------------------------------------
// The kernel.  Wave 0 of every workgroup (the control wave) carries no samples: it does
// everything that is serial per batch -- publishing the workgroup's partial tallies, waiting for the
// other slices, the 16 row LUTs (two float64 divisions each) and the four 256-entry tables -- while
// the other T/64-1 waves accumulate the previous batch.  One barrier per batch, tables double
// buffered.  Slices are (T-64)*16 samples wide.
template <int WAVES>  // data waves
struct __attribute__((aligned(16))) FusedCwLds {
    double table[2][4][256];
    double lut[kRowsPerBatch][4];
    // partial tallies of a batch: [parity][row group][data wave][lane]; lane = 16-lane DPP row * 16 +
    // row in group * 4 + lane in quad: every data lane stores its partial of ONE row (plain stores, no
    // LDS atomics); the control wave sums the 4 lanes x 4 DPP rows x WAVES slots of a row
    uint32_t tslot[2][4][WAVES][64];
};

template <int T, int DBG, int PAR>  // DBG: diagnostics build (bit 0 re-read rows 0..15, bit 1 skip accumulation);
                                    // PAR: the cohort is in the parity layout (nps_cohort_optimize)
__global__ __launch_bounds__(T, 4) void fused_cw_kernel(const FusedArgs a) {
    static_assert(T % 64 == 0 && T >= 128 && T <= 1024, "workgroup size");
    constexpr int TD = T - 64;  // data threads
    constexpr int kRing = 4;    // batches in flight per data thread: k (accumulating) .. k+3 (tallying)
    __shared__ FusedCwLds<TD / 64> lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const uint32_t team = blockIdx.y;   // grid = (P slices, Q teams)
    const uint32_t slice = blockIdx.x;
    const uint32_t n_local = a.n_batches > team ? (a.n_batches - team + a.Q - 1) / a.Q : 0;
    const uint32_t n_steps = (n_local + kRing - 1) / kRing * kRing;  // data loop unrolled by the ring

    auto batch_row0 = [&](uint32_t k) -> uint64_t {
        return (uint64_t)(team + (uint64_t)k * a.Q) * kRowsPerBatch;
    };


    // Barrier #j (j = 0,1,2,...) closes the phase in which the data waves tallied batch j.
    //   data waves, phase k (between #(k+2) and #(k+3)): accumulate batch k with tables[k&1],
    //       tally batch k+3, refill the ring slot of batch k with batch k+4
    //   control wave, same phase: publish batch k+2 (tallied in the previous phase); the tally words
    //       of batch k+1 were published by every slice one whole phase ago, so one poll normally
    //       suffices: LUTs and tables of batch k+1 -> tables[(k+1)&1]
    if (tid < 64) {
        // ------------------------------------------------------------------ control wave
        uint32_t nloci_local = 0;
        bool timed_out = false;
        NPS_TEL(unsigned long long tel_spins = 0; unsigned long long tel_wait = 0;
                unsigned long long tel_chain = 0; unsigned long long tel_bar = 0;)

        auto publish = [&](uint32_t k) {  // the tally slots of batch k are complete (barrier passed)
            // lane = (row of the batch, DPP row of the data waves): sum the four quad lanes of that slot
            // over the data waves, then over the four lanes of the quad
            const int prow = lane >> 2;
            const uint4 *slot = reinterpret_cast<const uint4 *>(
                &lds.tslot[k & 1][prow >> 2][0][(lane & 3) * 16 + (prow & 3) * 4]);
            uint32_t v = 0;
#pragma unroll
            for (int w = 0; w < TD / 64; ++w) {
                const uint4 q = slot[w * 16];
                v += (q.x + q.y) + (q.z + q.w);
            }
            v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);  // lane ^ 1
            v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);  // lane ^ 2
            if ((lane & 3) == 0) {
                const uint64_t row = batch_row0(k) + prow;
                if (k < n_local && row < a.n_rows) {
                    const uint64_t t = v >> 16, m = v & 0xFFFFu;
                    const uint64_t neff = t - m;
                    const unsigned long long add = (1ull << 56) | (m << 28) | neff;
                    __hip_atomic_fetch_add(&a.tally[row], add, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        };

        // phase work: publish batch kp, then tables of batch kt (its tally words are polled first,
        // so the publish and the row-descriptor loads hide under the poll's round trip)
        auto phase = [&](uint32_t kp, uint32_t kt) {
            const uint64_t row = batch_row0(kt) + lane;
            const bool valid = lane < kRowsPerBatch && kt < n_local && row < a.n_rows;
            unsigned long long x = 0;
            double beta = 0.0, eaf = 0.0;
            bool rie = false;
            NPS_TEL(const unsigned long long t_w0 = __builtin_amdgcn_s_memtime();)
            if (valid) {
                x = __hip_atomic_load(&a.tally[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                beta = a.desc[row].beta;
                eaf = a.desc[row].eaf;
                rie = a.desc[row].ref_is_effect != 0;
            }
            publish(kp);
            bool ok = !valid || (uint32_t)(x >> 56) == a.P;
            uint32_t spins = 0;
            while (!__all(ok) && !timed_out) {
                NPS_TEL(++tel_spins;)
                __builtin_amdgcn_s_sleep(1);
                if (!ok) {
                    x = __hip_atomic_load(&a.tally[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = (uint32_t)(x >> 56) == a.P;
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
            NPS_TEL(tel_wait += __builtin_amdgcn_s_memtime() - t_w0;)
            int used = 0;
            if (lane < kRowsPerBatch) {
                double v[4] = {0.0, 0.0, 0.0, 0.0};
                if (valid && ok)
                    row_lut(a, x, row, beta, eaf, rie, slice == 0 && a.stats != nullptr, v, used);
                *reinterpret_cast<double2 *>(&lds.lut[lane][0]) = make_double2(v[0], v[1]);
                *reinterpret_cast<double2 *>(&lds.lut[lane][2]) = make_double2(v[2], v[3]);
            }
            nloci_local += (uint32_t)__popcll(__ballot(used != 0));
            // tables: lane = (group g, c2, c3); 16 (c0,c1) entries each, summed in row order
            const int g = lane >> 4, c2 = lane & 3, c3 = (lane >> 2) & 3;
            const double l2 = lds.lut[4 * g + 2][c2], l3 = lds.lut[4 * g + 3][c3];
            double l0[4], l1[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                l0[c] = lds.lut[4 * g][c];
                l1[c] = lds.lut[4 * g + 1][c];
            }
            double *tab = lds.table[kt & 1][g];
#pragma unroll
            for (int c1 = 0; c1 < 4; ++c1)
#pragma unroll
                for (int c0 = 0; c0 < 4; ++c0) {
                    tab[table_index(c0, c1, c2, c3, PAR)] = ((l0[c0] + l1[c1]) + l2) + l3;
                }
        };

        __syncthreads();  // #0
        publish(0);
        __syncthreads();  // #1
        phase(1, 0);
        __syncthreads();  // #2
        for (uint32_t k = 0; k < n_steps; ++k) {
            NPS_TEL(const unsigned long long t0 = __builtin_amdgcn_s_memtime();)
            phase(k + 2, k + 1);
            NPS_TEL(const unsigned long long t1 = __builtin_amdgcn_s_memtime();)
            __syncthreads();  // #(k+3)
            NPS_TEL(tel_chain += t1 - t0; tel_bar += __builtin_amdgcn_s_memtime() - t1;)
        }
        if (slice == 0 && lane == 0 && nloci_local)
            atomicAdd(a.nloci, (unsigned long long)nloci_local);
        NPS_TEL(if (lane == 0 && a.telemetry) {  // summed over workgroups; read by the host
            atomicAdd(&a.telemetry[0], tel_spins);
            atomicAdd(&a.telemetry[1], tel_wait);
            atomicAdd(&a.telemetry[2], tel_chain);
            atomicAdd(&a.telemetry[3], tel_bar);
            atomicAdd(&a.telemetry[4], (unsigned long long)n_steps);
        })
        return;
    }

    // ---------------------------------------------------------------------- data waves
    const int dt = tid - 64;
    const uint32_t col = slice * TD + dt;
    const bool active = col < a.n_words;
    const uint32_t voff = col * 16u;            // byte offset of the thread's column inside a group
    const uint32_t row_bytes = a.n_words * 4u;
    const uint64_t group_bytes = a.stride_words * 16u;

    double acc[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) acc[s] = 0.0;
    uint32_t ring[kRing][kRowsPerBatch];

    // One buffer descriptor per row GROUP (wave-uniform, in SGPRs) and one 16-byte load per thread
    // and group: the thread's word column of the group's four rows.  The hardware range check
    // returns 0 for columns past the end of a row and for groups past the end of the matrix, so the
    // loads need no per-lane predication and no 64-bit per-lane addresses.
    auto load_batch = [&](uint32_t k, uint32_t(&dst)[kRowsPerBatch]) {
        const uint64_t row0 = batch_row0(k);
        const bool in = k < n_local && row0 < a.n_rows;
        // groups of this batch that hold at least one row of the matrix
        const uint32_t ngroups = in ? (uint32_t)((min((uint64_t)kRowsPerBatch, a.n_rows - row0) + 3) / 4) : 0u;
        const char *p = reinterpret_cast<const char *>(a.codes) +
                        (in && !(DBG & 1) ? (row0 >> 2) : 0) * group_bytes;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char *>(p), 0, (uint32_t)g < ngroups ? row_bytes * 4u : 0u, 0x00020000);
            const auto q = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 2);
            dst[4 * g + 0] = q[0];
            dst[4 * g + 1] = q[1];
            dst[4 * g + 2] = q[2];
            dst[4 * g + 3] = q[3];
            p += group_bytes;
        }
    };

    // reduce-scatter of four packed row tallies over the quads of each 16-lane DPP row
    auto tally_reduce4 = [&](int par, int g, const uint32_t(&tp)[4]) {
        // A DPP bank mask enables whole quads (bank b = lanes 4b..4b+3 of a 16-lane row), so the scatter
        // goes over the quads -- distance 8, then distance 4, every add writing only the quads that keep
        // its row: lane j of quad b of every 16-lane row ends with the sum of tally row b over lanes
        // j, j+4, j+8, j+12 of that row.  6 adds, no selects; the control wave adds the four lanes of
        // the quad.  s_nop: a DPP operand written by one of the two preceding VALU instructions needs
        // wait states the compiler cannot see inside the asm.
        uint32_t b0, b1, c;
        asm volatile(
            "s_nop 1\n\t"
            "v_add_u32_dpp %0, %3, %3 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_u32_dpp %1, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_u32_dpp %0, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "v_add_u32_dpp %1, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
            "s_nop 0\n\t"
            "v_add_u32_dpp %2, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_u32_dpp %2, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
            : "=&v"(b0), "=&v"(b1), "=&v"(c)
            : "v"(tp[0]), "v"(tp[1]), "v"(tp[2]), "v"(tp[3]));
        lds.tslot[par][g][dt >> 6][lane] = c;
    };

    auto tally_local = [&](uint32_t k, const uint32_t(&src)[kRowsPerBatch]) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint32_t tp[4];
            // (parity layout: the stored high-bit plane of slot 0 is the XOR of the four rows' planes)
            tp[0] = tally_pack(PAR ? parity_fix(src[4 * g], src[4 * g + 1], src[4 * g + 2], src[4 * g + 3]) : src[4 * g]);
#pragma unroll
            for (int r = 1; r < 4; ++r) tp[r] = tally_pack(src[4 * g + r]);
            tally_reduce4(k & 1, g, tp);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // accumulate batch k (tables[k&1]) and tally batch k+3 in ONE instruction stream: a row of tally
    // popcounts behind every four table lookups, so that all data waves of the CU issue a uniform
    // LDS/VALU mix instead of queueing on the LDS in one phase and on the VALU in the next.
    auto accumulate_and_tally = [&](uint32_t k, const uint32_t(&cur)[kRowsPerBatch],
                                    const uint32_t(&tal)[kRowsPerBatch]) {
        const int par = (k + 3) & 1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint32_t x[4];
            transpose_fold_4x16(cur[4 * g], cur[4 * g + 1], cur[4 * g + 2], cur[4 * g + 3], x);
            const double *Tg = lds.table[k & 1][g];
            uint32_t tp[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                double v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = Tg[(x[q] >> (8 * kk)) & 0xFFu];
                // VALU work while the lookups are in flight
                tp[kk] = tally_pack(PAR && kk == 0 ? parity_fix(tal[4 * g], tal[4 * g + 1], tal[4 * g + 2], tal[4 * g + 3])
                                                   : tal[4 * g + kk]);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[4 * kk + q] += v[q];
                // Pin the four adds here.  Without the opaque use hipcc (ROCm 7.2) sinks the
                // v_add_f64 of a whole batch below the next barrier and spills the looked-up values
                // to scratch; with it, four lookups are in flight per wave at a time.
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(acc[4 * kk + q]));
                __builtin_amdgcn_sched_barrier(0);
            }
            tally_reduce4(par, g, tp);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    auto step = [&](uint32_t k, uint32_t(&r_cur)[kRowsPerBatch], const uint32_t(&r_tal)[kRowsPerBatch]) {
        if (DBG & 2)  // diagnostics only: tally without the accumulation
            tally_local(k + 3, r_tal);
        else
            accumulate_and_tally(k, r_cur, r_tal);
        load_batch(k + 4, r_cur);
        __syncthreads();  // #(k+3)
    };

    load_batch(0, ring[0]);
    load_batch(1, ring[1]);
    load_batch(2, ring[2]);
    load_batch(3, ring[3]);
    tally_local(0, ring[0]);
    __syncthreads();  // #0
    tally_local(1, ring[1]);
    __syncthreads();  // #1
    tally_local(2, ring[2]);
    __syncthreads();  // #2
    for (uint32_t k = 0; k < n_steps; k += kRing) {
        step(k, ring[0], ring[3]);
        step(k + 1, ring[1], ring[0]);
        step(k + 2, ring[2], ring[1]);
        step(k + 3, ring[3], ring[2]);
    }
    if (active) {
        double *dst = a.part + (uint64_t)team * a.part_team_stride + (uint64_t)col * 16;
#pragma unroll
        for (int s = 0; s < 16; s += 2)
            *reinterpret_cast<double2 *>(dst + s) = make_double2(acc[s], acc[s + 1]);
    }
}

// Epilogue of a fused pass: part0[i] (+)= sum_q part[q][i] (the teams' partial scores, fixed order, into
// chunk 0 of the context); the tally words go back to zero for the next pass (every slice has read
// them); a raised bounded-wait word is recorded in the context's sticky status word and cleared.
__global__ __launch_bounds__(256) void fold_kernel(const double *__restrict__ part, uint32_t Q,
                                                   uint64_t team_stride, uint64_t n,
                                                   double *__restrict__ part0, int overwrite,
                                                   unsigned long long *__restrict__ tally, uint64_t n_tally,
                                                   unsigned int *__restrict__ timeout,
                                                   unsigned long long *__restrict__ status) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * 256;
    for (uint64_t j = i; j < n_tally; j += nthreads) tally[j] = 0ull;
    if (i == 0 && timeout) {
        if (*timeout) atomicOr(status, 1ull);
        *timeout = 0u;
    }
    if (i >= n) return;
    double s = 0.0;
    for (uint32_t q = 0; q < Q; ++q) s += part[(uint64_t)q * team_stride + i];
    part0[i] = overwrite ? s : part0[i] + s;
}

// ---- host side ------------------------------------------------------------------------------
template <int T>
static hipError_t plan_for(int cus, uint64_t n_words, uint64_t n_batches, int max_q, FusedPlan *plan) {
    int per_cu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fused_cw_kernel<T, 0, 0>, T, 0);
    if (e != hipSuccess) return e;
    // the kernel needs 128 VGPRs: 16 waves per CU; never ask for more than that many workgroups
    per_cu = std::min(per_cu, T >= 768 ? 1 : 1024 / T);
    if (per_cu < 1) return hipSuccess;
    const uint64_t capacity = (uint64_t)cus * per_cu;
    const uint64_t cols = T - 64;  // word columns per workgroup (wave 0 is the control wave)
    const uint64_t P = (n_words + cols - 1) / cols;
    if (P > capacity || P > 255) return hipSuccess;  // 8-bit arrival count in the tally word
    uint64_t Q = std::min<uint64_t>(capacity / P, n_batches);
    if (max_q > 0) Q = std::min<uint64_t>(Q, (uint64_t)max_q);  // diagnostics: fewer CUs in use
    if (Q < 1 || Q > 65535) return hipSuccess;
    plan->threads = T;
    plan->P = (uint32_t)P;
    plan->Q = (uint32_t)Q;
    plan->n_batches = (uint32_t)n_batches;
    plan->part_team_stride = P * cols * 16;
    plan->ok = true;
    return hipSuccess;
}

hipError_t fused_plan(int device, uint64_t n_samples, uint64_t n_rows, int want_threads, int max_q,
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
    auto try_plan = [&](int threads, FusedPlan *out) -> hipError_t {
        *out = FusedPlan{};
        switch (threads) {
        case 256: return plan_for<256>(cus, n_words, n_batches, max_q, out);
        case 512: return plan_for<512>(cus, n_words, n_batches, max_q, out);
        case 768: return plan_for<768>(cus, n_words, n_batches, max_q, out);
        case 832: return plan_for<832>(cus, n_words, n_batches, max_q, out);
        case 896: return plan_for<896>(cus, n_words, n_batches, max_q, out);
        case 960: return plan_for<960>(cus, n_words, n_batches, max_q, out);
        case 1024: return plan_for<1024>(cus, n_words, n_batches, max_q, out);
        default: return hipErrorInvalidValue;
        }
    };
    if (want_threads) return try_plan(want_threads, plan);
    // A pass takes (row batches / teams) batch phases, and a phase is a little shorter with fewer data
    // waves per workgroup (measured: 14 waves 0.8 % faster than 15 at the same number of teams): the
    // most teams first, then the smallest workgroup that still gives that many.
    const int candidates[5] = {1024, 960, 896, 832, 512};
    for (int t : candidates) {
        FusedPlan p;
        e = try_plan(t, &p);
        if (e != hipSuccess) return e;
        if (p.ok && (!plan->ok || p.Q > plan->Q || (p.Q == plan->Q && t >= 832 && p.threads < plan->threads)))
            *plan = p;
    }
    return hipSuccess;
}

template <int PAR>
static const void *fused_entry(int threads) {
    return threads == 256   ? (const void *)fused_cw_kernel<256, 0, PAR>
           : threads == 512 ? (const void *)fused_cw_kernel<512, 0, PAR>
           : threads == 768 ? (const void *)fused_cw_kernel<768, 0, PAR>
           : threads == 832 ? (const void *)fused_cw_kernel<832, 0, PAR>
           : threads == 896 ? (const void *)fused_cw_kernel<896, 0, PAR>
           : threads == 960 ? (const void *)fused_cw_kernel<960, 0, PAR>
                            : (const void *)fused_cw_kernel<1024, 0, PAR>;
}

hipError_t launch_fused(hipStream_t st, const FusedPlan &plan, const uint32_t *d_codes,
                        uint64_t stride_words, uint64_t n_samples, uint64_t n_rows,
                        const nps_row_desc *d_desc, DevParams prm, unsigned long long *d_tally,
                        nps_locus_stat *d_stats, unsigned long long *d_nloci, double *d_part,
                        unsigned int *d_timeout, int parity) {
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
    a.telemetry = reinterpret_cast<unsigned long long *>(d_timeout) + 2;  // same 256-byte block
    void *args[] = {&a};
    const void *fn = parity ? fused_entry<1>(plan.threads) : fused_entry<0>(plan.threads);
#ifdef NPS_DIAGNOSTICS
    // diagnostics builds only (tools/mkexp.sh -DNPS_DIAGNOSTICS): NPS_DEBUG_FLAGS bit 0 = every batch
    // re-reads rows 0..15 (L2 hits), bit 1 = skip the accumulation; instantiated for T = 1024, plain layout
    const int dbg = getenv("NPS_DEBUG_FLAGS") ? atoi(getenv("NPS_DEBUG_FLAGS")) & 3 : 0;
    if (dbg && plan.threads == 1024 && !parity)
        fn = dbg == 1   ? (const void *)fused_cw_kernel<1024, 1, 0>
             : dbg == 2 ? (const void *)fused_cw_kernel<1024, 2, 0>
                        : (const void *)fused_cw_kernel<1024, 3, 0>;
#endif
    // cooperative launch: the runtime rejects a grid that cannot be fully resident
    return hipLaunchCooperativeKernel(fn, dim3(plan.P, plan.Q), dim3(plan.threads), args, 0, st);
}

hipError_t launch_fold(hipStream_t st, const double *d_part, uint32_t Q, uint64_t team_stride,
                       uint64_t n_samples, double *d_part0, int overwrite, unsigned long long *d_tally,
                       uint64_t n_tally, unsigned int *d_timeout, unsigned long long *d_status) {
    if (Q == 0) return hipSuccess;
    (void)hipGetLastError();  // drop any stale sticky error: report this launch only
    const uint64_t blocks = std::max<uint64_t>(1, (n_samples + 255) / 256);
    hipLaunchKernelGGL(fold_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, d_part, Q, team_stride,
                       n_samples, d_part0, overwrite, d_tally, n_tally, d_timeout, d_status);
    return hipGetLastError();
}

}  // namespace nps
