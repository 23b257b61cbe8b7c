// nps_kernels.h -- host-callable launchers for the gfx950 kernels of libnps.
// Internal header (not part of the C-ABI; the public boundary is include/nps.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nps.h"

namespace nps {

// Device layout of the packed GT matrix (DESIGN.md "Data layout"): 2-bit codes, 16 samples per
// uint32 word (word column c = samples 16c..16c+15), the two bits of a code in separate nibbles:
// byte k of a word holds samples 16c+4k..16c+4k+3, their LOW code bits in bits 0..3 and their HIGH
// code bits in bits 4..7 (word_to_planes / word_from_planes convert from / to the C-ABI's order,
// sample i in bits 2i, 2i+1).  With the planes apart, four row words become sixteen table indices in
// two merge stages instead of three.  Rows are stored in GROUPS OF FOUR, interleaved
// at word granularity: word (row, c) lives at uint32 index ((row/4)*stride_words + c)*4 + row%4, so
// the four rows a thread needs for one table lookup group are ONE 16-byte load and a wave reads
// 1 KiB contiguously.  stride_words = word columns per group, padded to 64 (a group is a multiple
// of 1 KiB); padding words and the rows that pad the last group are zero.
// Tally word per row (two-pass path): (nmissing << 32) | neffect.
constexpr uint32_t kStrideAlignWords = 64;
static __host__ __device__ inline uint64_t g4_word_index(uint64_t row, uint64_t col, uint64_t stride_words) {
    return ((row >> 2) * stride_words + col) * 4 + (row & 3);
}

// C-ABI word (sample i in bits 2i, 2i+1) -> device word (byte k: low bits of samples 4k..4k+3 in the
// low nibble, high bits in the high nibble), and back.  Two delta swaps per direction.
static __host__ __device__ inline uint32_t word_to_planes(uint32_t x) {
    uint32_t t = (x ^ (x >> 1)) & 0x22222222u;  // bits 1 <-> 2 of every nibble
    x ^= t ^ (t << 1);
    t = (x ^ (x >> 2)) & 0x0C0C0C0Cu;  // bit pairs (3:2) <-> (5:4) of every byte
    x ^= t ^ (t << 2);
    return x;
}
static __host__ __device__ inline uint32_t word_from_planes(uint32_t x) {
    uint32_t t = (x ^ (x >> 2)) & 0x0C0C0C0Cu;
    x ^= t ^ (t << 2);
    t = (x ^ (x >> 1)) & 0x22222222u;
    x ^= t ^ (t << 1);
    return x;
}
// bit of the LOW code bit of sample j (0..15) in a device word; the HIGH code bit is 4 above it
static __host__ __device__ inline int plane_bit(int j) { return 8 * (j >> 2) + (j & 3); }

static inline uint64_t words_for(uint64_t n_samples) { return (n_samples + 15) / 16; }
static inline uint64_t stride_words_for(uint64_t n_samples) {
    uint64_t w = words_for(n_samples);
    if (w == 0) w = 1;
    return (w + kStrideAlignWords - 1) / kStrideAlignWords * kStrideAlignWords;
}

struct DevParams {
    int32_t imp_locus, imp_missing, imp_sample;
    double max_missing_rate;
    double min_cs;  // compared in double, nimpress.nim:471
};

// raw GT buffer (device copy; elem_bytes 4 = bcf_get_genotypes int32, 1 / 2 = the int8 / int16 vector
// of a BCF record) -> row `row_in_group` of the group at d_group (group interleaved layout) + tally
// (atomic add into *tally)
hipError_t launch_decode_gt(hipStream_t st, const void *d_gts, int elem_bytes, uint64_t n, int ploidy,
                            int eaidx, uint32_t *d_group, int row_in_group,
                            unsigned long long *d_tally);

// tally of rows [0,n_rows) of a group-interleaved matrix (d_codes = first group):
// tally[row] = (nmiss<<32)|neff   (direct store)
hipError_t launch_tally_packed(hipStream_t st, const uint32_t *d_codes, uint64_t stride_words,
                               uint64_t n_samples, uint64_t n_rows, unsigned long long *d_tally,
                               int parity = 0);

// a plain (contiguous) packed row, device or pinned host memory: tally it and scatter it into row
// `row_in_group` of a group; bed_mode -1 = native codes, 0 / 1 = PLINK .bed row, effect allele A2 / A1
hipError_t launch_tally_scatter_row(hipStream_t st, const uint32_t *row, uint64_t n_samples, int bed_mode,
                                    uint32_t *d_group, int row_in_group,
                                    unsigned long long *d_tally);

// plain row-major rows (native codes, or PLINK .bed rows with a per-row effect-allele flag) -> the
// group-interleaved cohort layout, on the device; k <= 4*65535 rows per launch
hipError_t launch_interleave_rows(hipStream_t st, const uint32_t *d_src, uint64_t src_stride_words,
                                  uint64_t k, uint64_t n_samples, const uint8_t *d_mode, uint32_t *d_dst,
                                  uint64_t stride_words);

// nps_cohort_optimize (nps_kernels.hip): the "parity layout".  The table index of the accumulation kernels
// puts the four LOW code bits and index bit 4 into the five LDS bank-select bits; the other three bits
// choose among the eight entries of a bank, and a lookup costs one LDS cycle per different entry of the
// busiest bank.  In the plain layout bit 4 is the HIGH code bit of slot 0, so every lane whose slot 1..3
// carries a dosage-2 / missing code shares its bank with the (many) lanes that have the same low bits and
// no high bit at all.  In the parity layout the high-bit plane of slot 0 holds the XOR of the four rows'
// high bits instead: lanes with exactly one high bit set -- the common case after "none" -- go to the
// other sixteen banks.  Bank model on HWE genotypes: 2.06 LDS cycles per 32-lane lookup against 2.69 plain
// (2.32 for round 1's "most frequent row in slot 0").  The transform is its own inverse and independent of
// the data; the kernels undo it for the tally (two VALU ops per group of 4 words) and build their tables
// for the index as it comes.
static __host__ __device__ inline uint32_t parity_fix(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
    return w0 ^ ((w1 ^ w2 ^ w3) & 0xF0F0F0F0u);
}
// table address of a sample whose four rows have codes c0..c3 (bit 0 low, bit 1 high code bit)
static __host__ __device__ inline int table_index(int c0, int c1, int c2, int c3, int parity) {
    const int h0 = parity ? ((c0 ^ c1 ^ c2 ^ c3) >> 1) & 1 : (c0 >> 1);
    return (c0 & 1) | ((c1 & 1) << 1) | ((c2 & 1) << 2) | ((c3 & 1) << 3) | (h0 << 4) | ((c1 >> 1) << 5) |
           ((c2 >> 1) << 6) | ((c3 >> 1) << 7);
}
// in place, all groups of a resident 2-bit cohort; applied again it restores the plain layout
hipError_t launch_cohort_parity(hipStream_t st, uint32_t *d_codes, uint64_t stride_words, uint64_t n_samples,
                                uint64_t n_rows);

// per-row decision + LUT {0b,1b,2b,imp*b} (or the locus constant); rows [n_rows, n_rows_pad) get a
// zero LUT.  Adds the number of used rows to *d_nloci.
hipError_t launch_row_params(hipStream_t st, const unsigned long long *d_tally,
                             const nps_row_desc *d_desc, uint64_t n_rows, uint64_t n_rows_pad,
                             uint64_t n_samples, DevParams p, double *d_lut, nps_locus_stat *d_stats,
                             unsigned long long *d_nloci);

// scores partials: part[chunk][sample] += sum over the chunk's rows of LUT[row][code]
struct AccumGeom {
    uint32_t n_words;          // ceil(N/16)
    uint32_t n_chunks;         // grid.y
    uint32_t groups_per_chunk; // row groups (4 rows) per chunk
    uint64_t part_chunk_stride; // doubles between chunks in `part` (>= n_words*16)
};
hipError_t launch_accumulate(hipStream_t st, const uint32_t *d_codes, uint64_t stride_words,
                             uint64_t n_rows, const double *d_lut, const AccumGeom &g,
                             double *d_part, int parity = 0);

// scores[i] = (sum_chunks part[c][i] + const_sum) / (2 * nloci) + offset      nimpress.nim:643-649
// nloci = host_nloci + *d_nloci (d_nloci may be null), read on the device so that no host round
// trip sits between the accumulation and this kernel; normalise = 0 hands out the plain sums.
hipError_t launch_finish(hipStream_t st, const double *d_part, uint32_t n_chunks,
                         uint64_t part_chunk_stride, uint64_t n_samples, double const_sum,
                         const unsigned long long *d_nloci, uint64_t host_nloci, int normalise,
                         double offset, double *d_scores);

// synthetic cohort rows (counter-based generator shared with oracle/refcpu.c): rows [row0, row0+n_rows)
// of the buffer are filled with the generator's rows gen_row0, gen_row0+1, ...
hipError_t launch_synth_gt(hipStream_t st, uint32_t *d_codes, uint64_t stride_words,
                           uint64_t n_samples, uint64_t row0, uint64_t gen_row0, uint64_t n_rows, uint64_t seed,
                           const uint32_t *d_t_het, const uint32_t *d_t_hom,
                           const uint32_t *d_t_miss);

// ---- fused single-read kernel (nps_fused.hip) ---------------------------------------------
struct FusedPlan {
    bool ok = false;       // shape fits the persistent grid
    uint32_t threads = 0;  // workgroup size
    uint32_t P = 0, Q = 0; // slices per team, teams
    uint32_t n_batches = 0;
    uint64_t part_team_stride = 0;  // doubles per team in the partial-score buffer
};
// want_threads: 0 = default, else a workgroup size the kernel is instantiated for; max_q: 0 = no limit
// (both non-zero only in diagnostics builds)
hipError_t fused_plan(int device, uint64_t n_samples, uint64_t n_rows, int want_threads, int max_q,
                      FusedPlan *plan);
// d_tally: [plan.n_batches*16] zeroed; d_part: [Q*part_team_stride]; d_timeout: zeroed word
hipError_t launch_fused(hipStream_t st, const FusedPlan &plan, const uint32_t *d_codes,
                        uint64_t stride_words, uint64_t n_samples, uint64_t n_rows,
                        const nps_row_desc *d_desc, DevParams prm, unsigned long long *d_tally,
                        nps_locus_stat *d_stats, unsigned long long *d_nloci, double *d_part,
                        unsigned int *d_timeout, int parity = 0);
// Epilogue of a fused pass, one launch: part0[i] (+)= sum_q part[q][i] (overwrite != 0: part0 holds
// nothing yet and is written, not read); the n_tally tally words are zeroed again for the next pass;
// a raised bounded-wait word is ORed into *d_status and cleared.
hipError_t launch_fold(hipStream_t st, const double *d_part, uint32_t Q, uint64_t team_stride,
                       uint64_t n_samples, double *d_part0, int overwrite, unsigned long long *d_tally,
                       uint64_t n_tally, unsigned int *d_timeout, unsigned long long *d_status);

// ---- FORMAT/DS float32 path (nps_ds.hip) ---------------------------------------------------------
struct DsTally {
    unsigned long long nmiss;
    double neff;
};
struct DsRowP {
    double beta, imp, cst;
    int32_t mode;  // 0 dropped, 1 genotyped (imp = value for missing samples), 2 locus constant cst
    int32_t rie;   // effect allele is REF: dosage = 2 - DS
};
static inline uint64_t ds_stride_floats(uint64_t n_samples) {
    uint64_t w = n_samples ? n_samples : 1;
    return (w + 63) / 64 * 64;
}
hipError_t launch_ds_tally(hipStream_t st, const float *d_ds, uint64_t stride_f, uint64_t n,
                           const nps_row_desc *d_desc, uint64_t n_rows, DsTally *d_tally);
hipError_t launch_ds_params(hipStream_t st, const DsTally *d_tally, const nps_row_desc *d_desc,
                            uint64_t n_rows, uint64_t n_samples, DevParams p, DsRowP *d_rowp,
                            nps_locus_stat *d_stats, unsigned long long *d_nloci);
hipError_t launch_ds_accumulate(hipStream_t st, const float *d_ds, uint64_t stride_f, uint64_t n,
                                const DsRowP *d_rowp, uint64_t n_rows, double *d_part,
                                uint32_t n_chunks, uint64_t part_chunk_stride);
// single-read kernel for a resident DS cohort (nps_ds_fused.hip); plan.threads/P/Q as for the GT kernel.
// d_tally: [n_rows] zeroed; d_psum: [n_rows * plan.P]; d_part: [Q*part_team_stride]
hipError_t ds_fused_plan(int device, uint64_t n_samples, uint64_t n_rows, int want_threads, int max_q,
                         FusedPlan *plan, int elem_bytes = 4 /* 2: NPS_FMT_DS16 (four rows per batch) */);
hipError_t launch_ds_fused(hipStream_t st, const FusedPlan &plan, const void *d_ds, uint64_t stride_bytes,
                           int elem_bytes /* 4: float32 rows; 2: NPS_FMT_DS16 rows */, uint64_t n_samples, uint64_t n_rows, const nps_row_desc *d_desc,
                           DevParams prm, int64_t t_maxmis /* largest nmissing not over --maxmis, -1: none */,
                           unsigned long long *d_tally /* [n_rows][2], zero */,
                           nps_locus_stat *d_stats, unsigned long long *d_nloci, double *d_part,
                           unsigned int *d_timeout);
// per row of a float32 dosage matrix: d_bad[r] = 1 when a value that is not NaN lies outside [0, 2]
hipError_t launch_ds_range_check(hipStream_t st, const float *d_ds, uint64_t stride_f, uint64_t n, uint64_t n_rows,
                                 unsigned char *d_bad);
hipError_t launch_decode_gt_to_ds(hipStream_t st, const void *d_gts, int elem_bytes, uint64_t n,
                                  int ploidy, int eaidx, float *d_out);
hipError_t launch_synth_ds(hipStream_t st, float *d_ds, uint64_t stride_f, uint64_t n, uint64_t row0,
                           uint64_t gen_row0, uint64_t n_rows, uint64_t seed, const uint32_t *d_t_het,
                           const uint32_t *d_t_hom, const uint32_t *d_t_miss);
// The float32 value of code k of a NPS_FMT_DS16 cohort, 0 <= k <= 20 000: float32(k / 10^4) correctly rounded -- the float32
// a decimal parser gives for the text -- in three float32 operations: with c1 = float32(1e-4) and c2 = float32(1e-4 - c1),
// fma(k, c1, float32(k c2)) equals the correctly rounded quotient for EVERY such k (checked exhaustively with exact
// rationals when this was written, and on the device by tests/test_gpu_parity.py::test_ds16_every_code_round_trips; the
// oracle computes float32(double(k) * 1e-4), which tests/test_host_logic.py checks against strtof for every k).  The
// obvious float64 form costs three quarter-rate conversions and a float64 multiply per genotype: 57 ms instead of 38 for a
// pass the float32 kernel does in 38.
#ifdef __HIPCC__
static __device__ __forceinline__ float ds16_value(uint32_t k) {
    const float a = (float)k;
    return __fmaf_rn(a, 9.999999747378752e-05f, a * 2.5262125290942405e-12f);
}
#endif
// NPS_FMT_DS16 (2 bytes per genotype: k = dosage x 10^4, 0xFFFF = missing): the generator (device copy of ref_synth_ds16),
// float32 rows -> k with a per-row flag for rows that hold a value no k stands for, and k -> float32 rows
hipError_t launch_synth_ds16(hipStream_t st, uint16_t *d_ds, uint64_t stride_e, uint64_t n, uint64_t row0, uint64_t gen_row0,
                             uint64_t n_rows, uint64_t seed, const uint32_t *d_t_het, const uint32_t *d_t_hom,
                             const uint32_t *d_t_miss);
hipError_t launch_ds16_pack(hipStream_t st, const float *d_src, uint64_t src_stride_f, uint64_t n, uint64_t n_rows,
                            uint16_t *d_dst, uint64_t dst_stride_e, unsigned char *d_bad);
hipError_t launch_ds16_unpack(hipStream_t st, const uint16_t *d_src, uint64_t src_stride_e, uint64_t n, uint64_t n_rows,
                              float *d_dst, uint64_t dst_stride_f);

// ---- several scores in one pass on the matrix cores (nps_multi.hip) -----------------------------
// NPS_FMT_GT2M cohort: ceil(rows/128) superblocks x ceil(samples/32) groups x 1 KiB units
static inline uint64_t gt2m_groups(uint64_t n_samples) { return (n_samples + 31) / 32; }
static inline uint64_t gt2m_superblocks(uint64_t n_rows) { return (n_rows + 127) / 128; }
static inline uint64_t gt2m_bytes(uint64_t n_samples, uint64_t n_rows) {
    return gt2m_superblocks(n_rows) * gt2m_groups(n_samples) * 1024;
}
// rows [row0, row0+n_rows) (row0 a multiple of 128) from the synthetic generator's rows gen_row0.., and
// their whole-row tallies (nmissing << 32 | neffect) into d_tally[row0 ..]
hipError_t launch_synth_gt2m(hipStream_t st, void *d_units, uint64_t n_samples, uint64_t row0, uint64_t gen_row0,
                             uint64_t n_rows, uint64_t seed, const uint32_t *d_t_het, const uint32_t *d_t_hom,
                             const uint32_t *d_t_miss, unsigned long long *d_tally);
// a whole 2-bit row-major cohort (plain order) -> units and whole-row tallies (nmissing << 32 | neffect)
hipError_t launch_convert_gt2m(hipStream_t st, const uint32_t *d_src, uint64_t src_stride_words,
                               uint64_t n_samples, uint64_t n_rows, void *d_units, unsigned long long *d_tally);
struct MultiPlan {
    int ND = 0;                  // base-256 digits per weight (6 or 7)
    int T = 0;                   // tiles of 16 columns: ND digit columns and one flag column per score
    int TD = 0, TM = 0, TF0 = 0; // tiles of the dosage / the is-missing matrix (flags apart); first tile with flags
    int GW = 0;                  // sample groups per wave
    uint64_t n_groups = 0;
    uint32_t n_sb = 0, tiles = 0, sb_per_chunk = 0, n_chunks = 0;
    uint64_t table_bytes() const { return (uint64_t)n_sb * 2 * T * 2 * 64 * 16; }
    uint64_t flag_bytes() const { return (uint64_t)n_sb * 4; }  // one word per superblock, behind the tables
    uint64_t partial_elems() const { return (uint64_t)n_chunks * n_groups * 32 * T * 16; }
};
MultiPlan multi_plan(uint64_t n_samples, uint64_t n_rows, int S, int ND, int coarse_missing /* leading base-256 digits of the is-missing weights kept: 4, 5; 0 = all */, int cus);
size_t multi_state_bytes();
hipError_t launch_multi_params(hipStream_t st, const unsigned long long *d_tally, const nps_row_desc *d_desc,
                               uint64_t n_desc, int S, const MultiPlan &pl, uint64_t n_samples, DevParams p,
                               const int *d_F, void *d_table, void *d_state, int coarse_missing);
hipError_t launch_multi_mfma(hipStream_t st, const MultiPlan &pl, const void *d_units, uint64_t sb_first,
                             const void *d_table, int32_t *d_partial, const void *d_state,
                             const unsigned long long *d_tally, uint64_t n_rows, uint32_t *d_sbflag);
hipError_t launch_multi_fold(hipStream_t st, const MultiPlan &pl, const int32_t *d_partial, uint64_t n_samples, int S,
                             const int *d_F, double *d_part, int overwrite, void *d_state);
hipError_t launch_multi_finish(hipStream_t st, const double *d_part, uint64_t n_samples, int S, const void *d_state,
                               const double *d_offsets, int have_sums, double *d_scores, int normalise = 1);

// ---- single-read kernel on the matrix cores for the strip layout NPS_FMT_GT2X (nps_mx.hip) -----------------
// cohort = [strip of 2048 samples][superblock of 128 rows][unit of 32 samples][row][8 bytes]; codes 0, 1, 2 =
// dosage, 3 = missing; sample s of a unit in bits 2s, 2s+1 of the row's 8 bytes.  Every strip but the last has
// 64 units; the last has what is left.
struct MxGeom {
    uint64_t n_units = 0, n_sb = 0;
    uint32_t P = 0, nu_last = 0;
};
static inline MxGeom mx_geom(uint64_t n_samples, uint64_t n_rows) {
    MxGeom g;
    g.n_units = (n_samples + 31) / 32;
    g.n_sb = (n_rows + 127) / 128;
    g.P = (uint32_t)((g.n_units + 63) / 64);
    g.nu_last = g.P ? (uint32_t)(g.n_units - 64ull * (g.P - 1)) : 0u;
    return g;
}
static inline uint64_t gt2x_superblocks(uint64_t n_rows) { return (n_rows + 127) / 128; }
static inline uint64_t gt2x_bytes(uint64_t n_samples, uint64_t n_rows) {
    const MxGeom g = mx_geom(n_samples, n_rows);
    return g.n_units * g.n_sb * 1024;
}
// KiB index of unit `unit` (= sample / 32) of superblock sb
static __host__ __device__ inline uint64_t gt2x_unit_index(uint64_t unit, uint64_t sb, uint64_t n_units, uint64_t n_sb) {
    const uint64_t p = unit >> 6, P = (n_units + 63) >> 6;
    const uint64_t nu = p == P - 1 ? n_units - 64 * (P - 1) : 64;
    return p * 64 * n_sb + sb * nu + (unit & 63);
}
struct MxPlan {
    bool ok = false;
    bool given = false;  // the shape does not fit one cooperative grid (more strips than compute units): the row
                         // tallies come from launch_mx_tally, the accumulation runs as an ordinary grid (two reads)
    uint32_t P = 0, Q = 0, nu_last = 0, n_sb = 0, n_flush = 0;  // strips, row teams per strip (superblock k belongs to team k % Q)
    uint64_t cpart_floats = 0;  // digit sums handed to mx_fold_kernel
    // the first form (launch_fused_mx) may cut the unit sequence into strips of U = 62 units instead of the layout's 64
    // where that puts more compute units to work (500 000 samples: 253 strips instead of 245): Pv strips, the last of
    // nu_last_v units.  U = 64: Pv = P.  Every other kernel works on the layout's strips.
    uint32_t U = 64, Pv = 0, nu_last_v = 0;
};
// two_pass: plan the tally + accumulate pair whatever the shape (NPS_MODE_TWOPASS)
hipError_t mx_plan(int device, uint64_t n_samples, uint64_t n_rows, bool two_pass, MxPlan *plan);
// d_tally: [n_sb*128] zeroed; d_tally1: [ceil(P/16)][n_sb*128] zeroed (both are zero again after launch_mx_fold); d_cpart: [plan.cpart_floats]; d_const_sum: 2 * plan.Q doubles, zero on entry (the caller zeroes them again after launch_mx_fold); d_pre: 32 bytes
// per row (scratch, written by the pass's first launch);
// t_maxmis: largest nmissing with !((double)nmissing / (double)N > --maxmis); F: fixed-point scale 2^F with
// |beta| (4 + max(2, 2 |eaf|)) 2^F < 2^56 for every row
hipError_t launch_fused_mx(hipStream_t st, const MxPlan &plan, const void *d_units, uint64_t n_sb_cohort, uint64_t sb0,
                           uint64_t n_samples, uint64_t n_rows, const nps_row_desc *d_desc, DevParams prm,
                           int64_t t_maxmis, int F, void *d_pre, unsigned long long *d_tally,
                           unsigned long long *d_tally1, nps_locus_stat *d_stats,
                           unsigned long long *d_nloci, double *d_const_sum, float *d_cpart, unsigned int *d_timeout);
// plan.given only: the whole-row tallies of the run's rows into d_tally (zero on entry), one read of the matrix
hipError_t launch_mx_tally(hipStream_t st, const MxPlan &plan, const void *d_units, uint64_t n_sb_cohort, uint64_t sb0,
                           uint64_t n_samples, unsigned long long *d_tally);
// keep (or nullptr): the run's complete tally words (nmissing << 28 | neffect, arrival count stripped) are copied there
// before they are zeroed -- the cohort's kept tallies as a by-product of a single-read pass
hipError_t launch_mx_fold(hipStream_t st, const MxPlan &plan, const float *d_cpart, uint64_t n_samples, int F,
                          const double *d_const_sum, double *d_part0, int overwrite, unsigned long long *d_tally,
                          uint64_t n_tally, unsigned long long *d_tally1, uint64_t n_tally1, unsigned int *d_timeout,
                          unsigned long long *d_status,
                          bool vstrips = false /* the digit sums come from launch_fused_mx with plan.U < 64 */,
                          unsigned long long *d_keep = nullptr, uint64_t n_keep = 0);
hipError_t launch_mx_prep(hipStream_t st, const nps_row_desc *d_desc, uint64_t n_rows, DevParams prm, int F, void *d_pre);
// nps_mxg.hip: the run with its row tallies GIVEN (plan.given: kept with the cohort, or from launch_mx_tally): per-row
// decisions + operands, then an ordinary grid of P x Q workgroups.  d_ops: 48 bytes per row padded to 128 rows;
// d_const_part: one double per superblock; d_done: one zeroed word (zero again afterwards); d_const_sum as for launch_fused_mx
hipError_t launch_mx_given(hipStream_t st, const MxPlan &plan, const void *d_units, uint64_t n_sb_cohort, uint64_t sb0,
                           uint64_t n_samples, uint64_t n_rows, const nps_row_desc *d_desc, DevParams prm,
                           int64_t t_maxmis, int F, const unsigned long long *d_tally, nps_locus_stat *d_stats,
                           unsigned long long *d_nloci, double *d_const_sum, float *d_cpart, void *d_ops,
                           double *d_const_part, unsigned int *d_done,
                           unsigned int *d_timeout /* raised when a wave's wait for its operand tables expires */);
// rows [row0, row0+n_rows) (row0 a multiple of 128) of a cohort of n_rows_cohort rows; rows past the end inside
// the last superblock written become zero
hipError_t launch_synth_gt2x(hipStream_t st, void *d_units, uint64_t n_samples, uint64_t n_rows_cohort, uint64_t row0,
                             uint64_t gen_row0, uint64_t n_rows, uint64_t seed, const uint32_t *d_t_het,
                             const uint32_t *d_t_hom, const uint32_t *d_t_miss);
hipError_t launch_rows_to_gt2x(hipStream_t st, const uint32_t *d_src, uint64_t src_stride_words, uint64_t n_samples,
                               uint64_t n_rows_cohort, uint64_t row0, uint64_t n_rows, void *d_units);
hipError_t launch_gt2x_to_rows(hipStream_t st, const void *d_units, uint64_t n_samples, uint64_t n_rows_cohort,
                               uint64_t row0, uint64_t n_rows, uint32_t *d_dst, uint64_t dst_stride_words);
hipError_t launch_gt2_to_gt2x(hipStream_t st, const uint32_t *d_src, uint64_t stride_words, uint64_t n_samples,
                              uint64_t n_rows, void *d_units);

}  // namespace nps
