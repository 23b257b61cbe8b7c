/*
 * nps.h -- C-ABI of libnps.so, the MI355X (gfx950) polygenic-score engine.
 *
 * This is the drop-in boundary for the per-variant inner loop of nimpress
 * (reference: mpinese/nimpress, src/nimpress.nim).  The reference has no FFI of its own; the
 * seam cut here is the body of computePolygenicScores' row loop -- everything between "the
 * host has located the VCF record and holds its raw FORMAT buffer" and "scores[i] updated":
 *
 *     getRawDosages          nimpress.nim:367-391   (GT decode -> per-sample dosage)
 *     tallyAlleles           nimpress.nim:32-47
 *     imputeLocusDosages     nimpress.nim:417-447
 *     imputeSampleDosages    nimpress.nim:450-481
 *     the maxmis decision    nimpress.nim:565-571
 *     scores[i] += d*beta    nimpress.nim:639-641
 *     /= 2*nloci, += offset  nimpress.nim:643-649
 *
 * Plain C types only; no exceptions cross the boundary.  Every call returns NPS_OK (0) or a
 * negative nps_status; nps_last_error() returns a thread-local message for the last failure.
 * There is NO CPU backend: without a usable HIP device every compute entry point fails with
 * NPS_E_NODEVICE.  (The CPU restatement under oracle/ is test infrastructure and is never
 * linked or loaded by this library.)
 *
 * Threading: a context / cohort is used by one host thread at a time (the reference is
 * single-threaded and synchronous, nimpress.nim:634-641).  Pushes are asynchronous on the
 * context's HIP stream; nps_flush / nps_finish synchronise.  Multi-GPU: one context per device
 * (one process per GPU under torch.distributed / RCCL).
 *
 * Ownership: the caller owns every input buffer and may reuse it as soon as the call returns
 * (the reference reuses `gts` / `dosages` per row, nimpress.nim:381,633).  Output buffers are
 * caller-allocated.
 */
#ifndef NPS_H
#define NPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NPS_ABI_VERSION 1

typedef enum nps_status {
    NPS_OK = 0,
    NPS_E_INVAL = -1,       /* bad argument (replaces the reference's doAssert aborts) */
    NPS_E_NODEVICE = -2,    /* no HIP device / device index out of range */
    NPS_E_HIP = -3,         /* a HIP runtime call failed */
    NPS_E_NOMEM = -4,       /* host or device allocation failed */
    NPS_E_STATE = -5,       /* call not valid in the current state */
    NPS_E_UNSUPPORTED = -6, /* e.g. ploidy > 8 */
    NPS_E_TIMEOUT = -7      /* an in-kernel bounded wait expired (fused kernel) */
} nps_status;

/* Enum values follow the declaration order of nimpress.nim:412-414, so a Nim caller can pass
 * ord(imputeMethodLocus) etc. unchanged. */
typedef enum nps_imp_locus {   /* ImputeMethodLocus   nimpress.nim:412 */
    NPS_LOCUS_PS = 0, NPS_LOCUS_HOMREF = 1, NPS_LOCUS_FAIL = 2, NPS_LOCUS_IGNORE = 3
} nps_imp_locus;
typedef enum nps_imp_missing { /* ImputeMethodMissing nimpress.nim:413 */
    NPS_MISSING_HOMREF = 0, NPS_MISSING_IGNORE = 1
} nps_imp_missing;
typedef enum nps_imp_sample {  /* ImputeMethodSample  nimpress.nim:414 */
    NPS_SAMPLE_PS = 0, NPS_SAMPLE_HOMREF = 1, NPS_SAMPLE_FAIL = 2, NPS_SAMPLE_INT_PS = 3,
    NPS_SAMPLE_INT_FAIL = 4
} nps_imp_sample;

/* What the host found for a score row before touching genotypes (getImputedDosages,
 * nimpress.nim:526-558). */
typedef enum nps_row_kind {
    NPS_ROW_PRESENT = 0,   /* record found, FILTER ok: genotype data follows      :561      */
    NPS_ROW_UNCOVERED = 1, /* --cov given and locus not covered                   :526-531  */
    NPS_ROW_ABSENT = 2,    /* findVariant returned nil                            :536-551  */
    NPS_ROW_FILTERED = 3   /* FILTER not in {".","PASS"} and not --ignorefilt     :553-558  */
} nps_row_kind;

/* Branch of getImputedDosages taken for a row (reported back so the host can emit the
 * reference's warnings in order, nimpress.nim:527,554,567,575). */
typedef enum nps_reason {
    NPS_REASON_GENOTYPED = 0, NPS_REASON_UNCOVERED = 1, NPS_REASON_ABSENT = 2,
    NPS_REASON_FILTERED = 3, NPS_REASON_MAXMIS = 4
} nps_reason;

/* The scalar arguments of computePolygenicScores that reach the inner loop
 * (nimpress.nim:592-599; afMismatchPthresh only gates warnings and stays on the host). */
typedef struct nps_params {
    int32_t imp_locus;        /* nps_imp_locus   */
    int32_t imp_missing;      /* nps_imp_missing */
    int32_t imp_sample;       /* nps_imp_sample  */
    int32_t reserved;         /* must be 0 */
    double max_missing_rate;  /* --maxmis, compared as nmissing/N > rate (strict, double) :565 */
    int64_t min_cs;           /* --mincs, compared as ngenotyped >= mincs in double      :471 */
} nps_params;

/* Per-row result, in push order.  For GT rows all three tallies are exact integers
 * (tallyAlleles, nimpress.nim:32-47); for DS rows neffect is a float64 sum. */
typedef struct nps_locus_stat {
    uint64_t ngenotyped;
    uint64_t nmissing;
    double neffect;
    int32_t used;    /* return value of getImputedDosages (nimpress.nim:484) */
    int32_t reason;  /* nps_reason */
} nps_locus_stat;

/* One score row for the resident-cohort entry point (ScoreEntry fields that reach the loop,
 * nimpress.nim:221-228, plus what the host found). */
typedef struct nps_row_desc {
    double beta;
    double eaf;            /* may be NaN (makescore.R:533) */
    int32_t kind;          /* nps_row_kind; PRESENT rows consume cohort rows in order */
    int32_t ref_is_effect; /* scoreEntry.refseq == scoreEntry.easeq */
} nps_row_desc;

/* Device time per kernel class, measured with HIP events on the context's stream. */
typedef struct nps_profile {
    double ms_decode;      /* raw FORMAT -> packed codes (+tally) */
    double ms_tally;       /* tally of packed rows */
    double ms_params;      /* per-row LUT / decision kernel */
    double ms_accumulate;  /* score accumulation (two-pass path) */
    double ms_fused;       /* fused single-read tally+accumulate kernel */
    double ms_reduce;      /* partial-score combine / finish */
    uint64_t n_decode, n_tally, n_params, n_accumulate, n_fused, n_reduce; /* launches */
} nps_profile;

typedef struct nps_ctx nps_ctx;
typedef struct nps_cohort nps_cohort;
typedef struct nps_scoredef nps_scoredef;

/* ---- library ------------------------------------------------------------------------- */
int nps_abi_version(void);
const char *nps_last_error(void);
/* number of HIP devices visible (0 when there is none; never initialises a device) */
int nps_device_count(void);
/* Create the HIP context of `device` and load the library's code object now rather than inside the first call that
 * needs them (half a second of a cold process).  Thread-safe; meant to be called from a thread of its own while the
 * caller opens, inflates and parses its input files (the host layer does: DESIGN.md section 9, f1).  Has no effect
 * on results; NPS_E_NODEVICE without a GPU. */
int nps_warmup(int device);

/* ---- streaming scorer: one call per score row, in score-file order --------------------- */

/* Replaces the set-up of computePolygenicScores (nimpress.nim:623-633): zeroed scores for
 * n_samples, nloci = 0.  device = HIP device ordinal. */
int nps_create(nps_ctx **out, int device, uint64_t n_samples, const nps_params *params);
uint64_t nps_n_samples(const nps_ctx *ctx); /* what nps_create was given (vcf.n_samples, nimpress.nim:626) */
int nps_device(const nps_ctx *ctx);         /* its HIP device ordinal */

/* PRESENT row with FORMAT/GT.  `gts` is the buffer bcf_get_genotypes fills (what hts-nim's
 * `genotypes(variant.format, gts)` iterates, nimpress.nim:381-384): n_samples*ploidy int32,
 * allele a encoded (a+1)<<1|phased, missing allele 0, vector-end pad 0x80000001.
 * eaidx: 0 = REF, k = ALT[k-1] (nimpress.nim:375-379).  Replaces getRawDosages ..
 * `scores[i] += dosages[i]*beta` for this row (nimpress.nim:561-583, 639-641).
 * ploidy 1..2 rows go to the 2-bit matrix; ploidy 3..8 rows (dosage may exceed 2, the reference
 * counts every allele, nimpress.nim:385-390) are decoded on the device into a float dosage row
 * and scored with the DS kernels.  ploidy > 8: NPS_E_UNSUPPORTED. */
int nps_push_gt(nps_ctx *ctx, const int32_t *gts, int ploidy, int eaidx, int ref_is_effect,
                double beta, double eaf);

/* Same row, but `gt` is the typed GT vector exactly as a BCF2 record stores it (what htslib holds
 * BEFORE bcf_get_genotypes widens it to int32): n_samples*ploidy elements of elem_bytes = 1, 2
 * (or 4) bytes, same encoding; end-of-vector 0x81 / 0x8001 and the typed missing value 0x80 /
 * 0x8000 are negative and therefore skipped exactly like 0x80000001 (sign extension on the
 * device).  Moves 2 instead of 8 bytes per diploid genotype over PCIe. */
int nps_push_gt_raw(nps_ctx *ctx, const void *gt, int elem_bytes, int ploidy, int eaidx,
                    int ref_is_effect, double beta, double eaf);

/* PRESENT row straight from a PLINK 1 .bed file (variant-major mode): ceil(n_samples/4) bytes, 4
 * samples per byte, sample 0 in the low bits, values 0 = hom A1, 1 = missing, 2 = het, 3 = hom A2.
 * effect_is_a1 selects the allele the score row counts (.bim column 5 or 6).  The row is already
 * 2-bit and sample-minor: it is moved as it is and recoded on the device (build-defined extension;
 * the reference reads VCF/BCF only).
 * The argument is the row's CODE MAP (NPS_MAP_*): 0 / 1 are the two .bed maps; a fixed-width hard-call record of a
 * PLINK 2 .pgen file (storage mode 0x02: same packing, the code is the number of ALT alleles, 3 = missing) goes
 * through the same entry points with NPS_MAP_PGEN_ALT (the score row counts ALT) or NPS_MAP_PGEN_REF (it counts REF:
 * dosage = 2 - ALT count).  Also the per-row flags of nps_cohort_upload_bed and the flag of nps_cohort_push_bed. */
#define NPS_MAP_BED_A2 0
#define NPS_MAP_BED_A1 1
#define NPS_MAP_PGEN_ALT 2
#define NPS_MAP_PGEN_REF 3
int nps_push_bed(nps_ctx *ctx, const uint8_t *bed_row, int effect_is_a1, int ref_is_effect, double beta,
                 double eaf);

/* PRESENT row with FORMAT/DS (build-defined extension, the reference decodes GT only):
 * n_samples float32 ALT dosages, NaN = missing; ref_is_effect -> dosage = 2 - DS. */
int nps_push_ds(nps_ctx *ctx, const float *ds, int ref_is_effect, double beta, double eaf);

/* PRESENT row already as 2-bit codes: ceil(n_samples/16) little-endian uint32, sample i
 * in bits 2*(i%16).. of word i/16 (the library re-orders the bits of a word for its own layout); codes NPS_CODE_*: 0 = dosage 0, 1 = dosage 1, 3 = dosage 2,
 * 2 = missing (so that popcount(word) = effect alleles + missing samples); padding bits zero.
 * `row` is a host pointer. */
int nps_push_packed(nps_ctx *ctx, const uint32_t *row, int ref_is_effect, double beta, double eaf);

/* Row without genotype data: kind = UNCOVERED / ABSENT / FILTERED.  Replaces the early
 * returns of getImputedDosages (nimpress.nim:526-558). */
int nps_push_locus(nps_ctx *ctx, int kind, int ref_is_effect, double beta, double eaf);

/* Completes all pushed rows and copies out their nps_locus_stat in push order, starting after
 * the rows a previous flush returned.  stats_out may be NULL (cap ignored) to only synchronise.
 * *n_out = number of stats written. */
int nps_flush(nps_ctx *ctx, nps_locus_stat *stats_out, size_t cap, size_t *n_out);

/* nimpress.nim:643-649: scores[i] = sum/(2*nloci) + offset -> scores_out[n_samples];
 * *nloci_out = rows for which getImputedDosages returned true.  nloci = 0 gives NaN (0/0) as in
 * the reference.  The context can be reused after nps_reset. */
int nps_finish(nps_ctx *ctx, double offset, double *scores_out, uint64_t *nloci_out);
/* Same, but scores are written to a caller-allocated DEVICE buffer of n_samples doubles (e.g. a
 * torch tensor handed to an RCCL all-gather): no PCIe round trip. */
int nps_finish_device(nps_ctx *ctx, double offset, double *d_scores_out, uint64_t *nloci_out);
/* Row-sharded evaluation of ONE score over several GPUs (each holds a block of the score's rows
 * and all samples, so tallies stay local and exact): the state of the reference's loop before its
 * normalisation (nimpress.nim:639-641) -- d_sums_out[n_samples] = sum of dosage*beta over this
 * context's rows, *nloci_out = its used rows -- is written to a DEVICE buffer for the one exchange of
 * that layout, a sum all-reduce of both (RCCL).  nps_normalize_device then applies
 * nimpress.nim:643-649 to the reduced sums in place: d[i] = d[i] / (2*nloci) + offset. */
int nps_partial_device(nps_ctx *ctx, double *d_sums_out, uint64_t *nloci_out);
int nps_normalize_device(nps_ctx *ctx, double *d_sums_inout, uint64_t nloci, double offset);
int nps_reset(nps_ctx *ctx, const nps_params *params /* NULL = keep */);
void nps_destroy(nps_ctx *ctx);

/* ---- resident cohort: a packed genotype matrix kept in HBM ------------------------------
 * Ordering: nps_score_cohort[_def] returns while its kernels still run on the context's stream.  The
 * calls that write cohort rows (nps_cohort_upload, _upload_bed, _synth, _optimize) first wait for
 * ALL work queued on the device (hipDeviceSynchronize), so a cohort can be modified safely at any
 * time; destroying a cohort waits the same way.  A 2-bit upload / synth that ends inside a group of
 * four rows zeroes the remaining rows of that last group (rows are stored in groups of four). */
#define NPS_FMT_GT2 0  /* 2-bit codes, 16 per uint32, variant-major / sample-minor */
#define NPS_CODE_DOSAGE0 0u
#define NPS_CODE_DOSAGE1 1u
#define NPS_CODE_MISSING 2u
#define NPS_CODE_DOSAGE2 3u
/* float32 dosages, NaN = missing, variant-major / sample-minor.  A dosage is 0 <= DS <= 2 (the FORMAT/DS convention
 * for a diploid sample; the host readers refuse a file that breaks it).  The single-read kernel (NPS_MODE_FUSED, and
 * NPS_MODE_AUTO where the shape allows it) relies on that range: it hands the slices' dosage sums over as fixed-point
 * integers (order-independent, bit-reproducible).  nps_cohort_upload checks the rows it receives; while a cohort
 * holds a value outside [0, 2], NPS_MODE_AUTO scores it with the two-pass kernels and NPS_MODE_FUSED returns
 * NPS_E_INVAL.  NPS_MODE_TWOPASS and the streamed nps_push_ds rows take any finite value. */
#define NPS_FMT_DS32 1
/* 2-bit codes in the layout of the multi-score (matrix-core) path: superblocks of 128 rows x groups of 32
 * samples, 16 ROWS of one sample per 32-bit word; carries its whole-row tallies.  Filled by
 * nps_cohort_convert (from a NPS_FMT_GT2 cohort) or nps_cohort_synth[_rows]; scored by nps_score_cohort_multi. */
#define NPS_FMT_GT2M 2
/* 2-bit codes in the strip layout of the matrix-core single-score kernel: strips of 2048 samples x
 * superblocks of 128 rows x 1 KiB units (128 rows x 32 samples, row-major).  Same C-ABI as NPS_FMT_GT2 towards the
 * caller (nps_cohort_upload / _download speak plain rows of NPS_CODE_* codes; row0 of an upload or a synthetic fill
 * must be a multiple of 128, and rows after its end inside the last superblock written become zero); filled also by
 * nps_cohort_convert from a NPS_FMT_GT2 cohort.  Scored by nps_score_cohort[_def] (cohort_row0 a multiple of 128,
 * NPS_MODE_AUTO or NPS_MODE_FUSED) with results identical in meaning to the NPS_FMT_GT2 kernels: tallies, nloci
 * and decisions bit-exact, scores equal up to the quantisation of the row weights (2^-56 of the largest one: the size
 * of float64 rounding of the terms themselves).  The kernel's time does
 * not depend on the genotypes.  Any cohort size (the reference scores any N, nimpress.nim:626-628): with
 * P = ceil(n_samples / 2048) strips, P <= compute units (N <= 522 240 on an MI355X) is ONE read of the matrix --
 * floor(CUs / P) row teams per strip fill the chip for small cohorts; more strips than that (or NPS_MODE_TWOPASS)
 * are scored in two reads (a tally pass, then the same accumulation with the tallies given); NPS_MODE_FUSED
 * insists on the single read and returns NPS_E_UNSUPPORTED where it cannot be had. */
#define NPS_FMT_GT2X 3
/* nps_cohort_create only: "the 2-bit resident layout this library scores best at this size" -- since round 5 that is
 * NPS_FMT_GT2X at every size (nps_cohort_format tells; use row offsets that are multiples of 128).  Under NPS_MODE_AUTO
 * a run whose resident grid (P = ceil(N / 2048) strips x floor(CUs / P) row teams) covers at least nine tenths of the
 * compute units counts its tallies in the pass (tallyAlleles, nimpress.nim:563, inside the one read), every time.  Other
 * sizes keep the cohort's tallies with the cohort -- NPS_MODE_AUTO ATTACHES this cache to the `const nps_cohort` it is given
 * (dropped by any call that rewrites rows; nps_cohort_has_tallies tells) -- and score later runs with the tallies given:
 *   128 < P < 0.9 CUs (262 145 .. 471 040 samples on an MI355X): the first whole-cohort run
 *     is the single read in which the tallies are counted anyway and keeps them as a by-product (round 6: ONE read, 0.51 of
 *     the roofline at 300 000 samples, 0.65 at 400 000; until then a tally pass + a given-tallies pass, two reads, 0.38);
 *     later runs 0.73-0.75;
 *   P > compute units (beyond 522 240 samples): no resident grid exists: the first run counts the tallies in a pass of
 *     its own (two reads, 0.38-0.39), later runs read once.
 * nps_cohort_expect_passes asks for the same caching at the sizes whose grid does cover the chip.  (Until round 4 the
 * awkward sizes got NPS_FMT_GT2, whose table-lookup kernel runs at 0.47 .. 0.63 of the roofline depending on the genotypes.) */
#define NPS_FMT_GT_AUTO 4
/* Dosages in 2 bytes per genotype (round 5): k = dosage x 10^4 as uint16, 0 <= k <= 20 000, 0xFFFF = missing.  FORMAT/DS
 * values are decimal text with one to four places (Beagle, minimac, IMPUTE): for every such value in [0, 2] the float32 a
 * parser makes of the text is float32(double(k) x 1e-4), so the format is LOSSLESS for them and the kernel computes with
 * exactly the numbers a NPS_FMT_DS32 cohort of the same file would hold -- at half the bytes.  Towards the host the format
 * looks like NPS_FMT_DS32: nps_cohort_upload takes float32 rows (NaN = missing) and returns NPS_E_UNSUPPORTED, naming the
 * row, if a row holds a value no k stands for (the rows of the call's range are undefined after that: use NPS_FMT_DS32);
 * nps_cohort_download gives float32 rows back; nps_cohort_synth[_rows] fills rows with three-decimal values.  Scored by
 * nps_score_cohort[_def] under NPS_MODE_AUTO / NPS_MODE_FUSED with the single-read kernel only: a shape beyond its resident
 * grid, or NPS_MODE_TWOPASS, returns NPS_E_UNSUPPORTED.  Like all of FORMAT/DS a build-defined extension (the reference
 * decodes GT only). */
#define NPS_FMT_DS16 5

int nps_cohort_create(nps_cohort **out, int device, uint64_t n_samples, uint64_t n_rows,
                      int format);
/* row stride in bytes of the device layout (rows are padded to 256 B) */
uint64_t nps_cohort_row_stride(const nps_cohort *c);
uint64_t nps_cohort_n_rows(const nps_cohort *c);
int nps_cohort_format(const nps_cohort *c); /* NPS_FMT_* (never NPS_FMT_GT_AUTO) */
/* copy rows [row0,row0+nrows) from host memory laid out with host_stride bytes per row */
int nps_cohort_upload(nps_cohort *c, uint64_t row0, uint64_t nrows, const void *host_rows,
                      size_t host_stride);
int nps_cohort_download(const nps_cohort *c, uint64_t row0, uint64_t nrows, void *host_rows,
                        size_t host_stride);
/* Rows of a PLINK 1 .bed file (variant-major; see nps_push_bed) -> rows [row0,row0+nrows) of a
 * NPS_FMT_GT2 cohort; effect_is_a1[j] selects the counted allele of row j.  A .bed file is already
 * 2-bit and sample-minor: bytes go over PCIe unchanged (pass the mmap'ed file + 3 header bytes,
 * row_stride_bytes = ceil(n_samples/4)) and are recoded + interleaved on the device. */
int nps_cohort_upload_bed(nps_cohort *c, uint64_t row0, uint64_t nrows, const uint8_t *bed_rows,
                          size_t row_stride_bytes, const uint8_t *effect_is_a1);
/* One row of a NPS_FMT_GT2 cohort straight from the buffer a record holds -- the typed FORMAT/GT vector of a VCF/BCF
 * record (see nps_push_gt_raw; elem_bytes 4 = the bcf_get_genotypes buffer of nimpress.nim:381-384) or a PLINK .bed
 * row (see nps_push_bed) -- decoded on the device by the kernels of the streaming entry points, with the cohort as
 * the destination: getRawDosages (nimpress.nim:367-391) for a matrix that stays resident (a cohort holding the union
 * of several score files' loci, scored with nps_score_cohort_multi after nps_cohort_convert).  ploidy 1 or 2.  The
 * caller may reuse its buffer on return; calls that read the cohort wait for the pushed rows. */
int nps_cohort_push_gt_raw(nps_cohort *c, uint64_t row, const void *gt, int elem_bytes, int ploidy, int eaidx);
int nps_cohort_push_bed(nps_cohort *c, uint64_t row, const uint8_t *bed_row, int effect_is_a1);
/* Fill rows on the device with the counter-based synthetic generator (DESIGN.md "Synthetic
 * cohorts"): per-row uint32 thresholds, code(seed,row,sample) reproducible on the CPU. */
/* One-time layout change of a resident 2-bit cohort (no-op for the other formats): inside every group of
 * four rows the high-bit plane of the first row is replaced by the XOR of the four rows' planes, which
 * spreads the accumulation kernels' table lookups over twice as many LDS banks (~14 % fewer LDS
 * cycles per lookup on HWE genotypes, independent of which rows are frequent).  Results, statistics and
 * nps_cohort_download are unchanged; a later upload / synth puts the cohort back into the plain layout
 * first.  The transform is its own inverse. */
int nps_cohort_optimize(nps_cohort *c);
int nps_cohort_synth(nps_cohort *c, uint64_t row0, uint64_t nrows, uint64_t seed,
                     const uint32_t *t_het, const uint32_t *t_hom, const uint32_t *t_miss);
/* Same, but cohort rows [row0, row0+nrows) receive the generator's rows gen_row0, gen_row0+1, ...: a
 * cohort that holds a block or a chunk of a larger matrix (row-sharded runs, matrices larger than HBM). */
int nps_cohort_synth_rows(nps_cohort *c, uint64_t row0, uint64_t nrows, uint64_t gen_row0, uint64_t seed,
                          const uint32_t *t_het, const uint32_t *t_hom, const uint32_t *t_miss);
void nps_cohort_destroy(nps_cohort *c);

/* Score n_desc rows in order; PRESENT rows take cohort rows cohort_row0, cohort_row0+1, ...
 * Equivalent to the matching sequence of nps_push_* calls, without host traffic.
 * mode: NPS_MODE_AUTO picks the fused single-read kernel when the shape allows it. */
#define NPS_MODE_AUTO 0
#define NPS_MODE_TWOPASS 1 /* tally kernel, then accumulate kernel (reads the matrix twice) */
#define NPS_MODE_FUSED 2   /* persistent fused kernel (reads the matrix once) */
int nps_score_cohort(nps_ctx *ctx, const nps_cohort *c, uint64_t cohort_row0,
                     const nps_row_desc *rows, uint64_t n_desc, int mode);

/* A score definition (the rows of one .scores file, nimpress.nim:247-254, plus what the host
 * found for each) kept on the device, so that re-scoring needs no host->device traffic. */
int nps_scoredef_create(nps_scoredef **out, int device, const nps_row_desc *rows, uint64_t n_desc);
uint64_t nps_scoredef_n_present(const nps_scoredef *d); /* rows that consume a cohort row */
void nps_scoredef_destroy(nps_scoredef *d);
/* Error behaviour: everything that can be refused (arguments, ranges, NPS_E_UNSUPPORTED shapes, a
 * failed allocation) is checked before the context changes, so such a call can simply be repeated
 * (e.g. with another mode).  A HIP failure between the first and the last launch of a run leaves the
 * context's sums undefined: every later call returns NPS_E_STATE until nps_reset. */
int nps_score_cohort_def(nps_ctx *ctx, const nps_cohort *c, uint64_t cohort_row0,
                         const nps_scoredef *def, int mode);

/* ---- several score definitions in ONE pass over a resident cohort --------------------------
 * The reference evaluates one score per run (nimpress.nim:634-641); S scores over the same cohort are S
 * passes over the genotypes.  Here the S definitions are applied together: scores[N x S] =
 * dosage[N x M] . weights[M x S] on the matrix cores (int8 codes x base-256 digits of the fixed-point
 * weights, exact integer accumulation: DESIGN.md), the genotypes are read once for all S.
 *
 * Position j of every definition refers to cohort row cohort_row0 + j (a cohort holding the union of
 * the scores' loci); per (score, position) `kind` says what that score does with the row:
 * NPS_ROW_PRESENT (the genotypes count), UNCOVERED / ABSENT / FILTERED (the host's early returns of
 * getImputedDosages, nimpress.nim:526-558: a constant, no genotypes) or NPS_ROW_NOT_IN_SCORE (the score
 * file does not list the locus).  Results per score are those of the single-score entry points
 * (same decisions, nloci bit-exact, scores within ~1e-13 relative: the weights are quantised to 2^-49 of
 * the largest one).  beta must be finite (NPS_E_UNSUPPORTED otherwise). */
#define NPS_ROW_NOT_IN_SCORE 4
#define NPS_MULTI_MAX_SCORES 8
typedef struct nps_multi nps_multi;
typedef struct nps_multidef nps_multidef;
/* rows: [n_scores][n_desc], score-major.  The weights of a definition are kept in fixed point, 49 bits below the
 * largest weight of the score (seven base-256 digits on the matrix cores).  nps_multidef_create_bits(.., 41) keeps 41
 * bits (six digits): every term of a score is then within 2^-41 of the largest weight, a typical score within 1e-10
 * relative -- but a sample whose terms cancel to 1e-5 of the typical size is only within ~1e-5 of its own value, so
 * it is NOT inside the 1e-6 relative bar for every sample and stays an option; with 5, 6 or 8 scores the pass needs
 * a quarter fewer matrix instructions for it.
 * One scale per score: a definition whose non-zero |beta| span more than 2^25 (2^17 with 41-bit weights) is refused
 * with NPS_E_UNSUPPORTED -- a sample that carries only its small-beta rows would not keep 1e-6 relative -- and belongs
 * on the single-score path (nps_scoredef_create scores such a definition in magnitude bands of 2^30). */
int nps_multidef_create(nps_multidef **out, int device, const nps_row_desc *rows, int n_scores, uint64_t n_desc);
int nps_multidef_create_bits(nps_multidef **out, int device, const nps_row_desc *rows, int n_scores, uint64_t n_desc,
                             int weight_bits /* 49, 41; 0 = default (49) */);
void nps_multidef_destroy(nps_multidef *d);
int nps_multi_create(nps_multi **out, int device, uint64_t n_samples, const nps_params *params, int n_scores);
/* Width of the fixed-point weight `(imputed dosage - 3) x beta` that a MISSING genotype adds on top of the
 * `3 x beta` its code already received (the dosage weights always carry 56 bits).  56 (default): as exact as
 * the dosage weights.  32: the four leading base-256 digits of the operands that carry it are kept (the kernel
 * multiplies byte prefixes of four packed genotypes, so a weight is known to 2^-24 of the largest one at worst) --
 * per sample an error of at most (its missing genotypes) x 2^-24 x B before the division by 2 nloci,
 * B = max|beta| x (3 + max(2, 2 max|eaf|)) of the score, typically the square root of that count -- and with more
 * than 4 scores the pass needs up to a quarter fewer matrix instructions.  40: five leading digits, 2^-32 x B per missing
 * genotype, an eighth fewer matrix instructions with 7 or 8 scores (round 5; like 32 an option, not the default: a sample
 * whose terms cancel is not within 1e-6 of its own score over a million rows).  NaN imputation values
 * (imp-sample fail / int_fail below --mincs) are exact in both modes.  Applies to the following calls. */
int nps_multi_set_missing_weight_bits(nps_multi *m, int bits);
/* cohort_row0 must be a multiple of 128; calls accumulate (chunks of a larger matrix) until nps_multi_reset.
 * Errors as for nps_score_cohort_def: refused calls leave the context unchanged; a HIP failure during the pass
 * makes every later call return NPS_E_STATE until nps_multi_reset. */
int nps_score_cohort_multi(nps_multi *m, const nps_cohort *c, uint64_t cohort_row0, const nps_multidef *def);
/* scores_out: [n_scores][n_samples]; nloci_out: [n_scores]; offsets: [n_scores]  (nimpress.nim:643-649) */
int nps_multi_finish(nps_multi *m, const double *offsets, double *scores_out, uint64_t *nloci_out);
int nps_multi_finish_device(nps_multi *m, const double *offsets, double *d_scores_out, uint64_t *nloci_out);
/* Row-sharded evaluation of S scores over several GPUs (each GPU holds a block of the union's rows and ALL samples:
 * tallies stay local and exact, and the cohort is split instead of replicated): the state of the reference's loop
 * before its normalisation (nimpress.nim:639-641), d_sums_out[n_scores][n_samples] and nloci_out[n_scores] of this
 * context's rows, for the one exchange of that layout -- a sum all-reduce of both (RCCL).  The caller then applies
 * nimpress.nim:643-649: sums / (2 nloci) + offset (nimpress_amd/multi.py: normalize_matrix). */
int nps_multi_partial_device(nps_multi *m, double *d_sums_out, uint64_t *nloci_out);
/* what nps_multi_create was given (a host that hands scorers to nps_comm_allreduce_partial_multi is checked against these) */
int nps_multi_n_scores(const nps_multi *m);
uint64_t nps_multi_n_samples(const nps_multi *m);
int nps_multi_device(const nps_multi *m);
/* the same into host memory: sums_out[n_scores][n_samples] (a host caller that exchanges through its own transport) */
int nps_multi_partial(nps_multi *m, double *sums_out, uint64_t *nloci_out);
int nps_multi_reset(nps_multi *m, const nps_params *params /* NULL = keep */);
void nps_multi_destroy(nps_multi *m);
/* device time (HIP events) of the calls since the last reset: weight digits, the product, the fold */
int nps_multi_timing(nps_multi *m, double *ms_params, double *ms_product, double *ms_fold);
/* NPS_FMT_GT2 cohort (plain order) -> NPS_FMT_GT2M (with its row tallies) or NPS_FMT_GT2X cohort of the same shape */
int nps_cohort_convert(nps_cohort *dst, const nps_cohort *src);
/* the whole-row tallies a NPS_FMT_GT2M cohort carries (tallyAlleles, nimpress.nim:32-47), for warnings; also those of a
 * NPS_FMT_GT2X cohort after nps_cohort_keep_tallies */
int nps_cohort_row_tallies(const nps_cohort *c, uint64_t row0, uint64_t nrows, uint64_t *nmissing_out,
                           uint64_t *neffect_out);
/* Count tallyAlleles (nimpress.nim:32-47, called per row at :563) of EVERY row of a NPS_FMT_GT2X cohort once and keep the
 * result with the cohort (one read of the matrix).  nps_score_cohort[_def] under NPS_MODE_AUTO then scores the cohort with
 * the tallies given -- no recount, no hand-over between the strips -- which is what many score files over one cohort want
 * (BASELINE configs[3]): the decision chain of getImputedDosages (:565-583) sees exactly the same counts.  Any call that
 * rewrites rows (upload, synth, convert) drops the kept tallies; NPS_MODE_FUSED / NPS_MODE_TWOPASS never use them.
 * (Measured, round 6: it pays on cohorts of more than 262 144 samples -- 0.71-0.79 of the roofline against 0.51-0.73 in the
 * pass; below that the in-pass kernel is as fast or faster, and NPS_MODE_AUTO on its own never keeps tallies there.) */
int nps_cohort_keep_tallies(nps_cohort *c);
/* A hint (round 6): the caller will score this NPS_FMT_GT2X cohort `n_passes` times (several score files over one cohort,
 * BASELINE configs[3]; the reference runs computePolygenicScores once per file, nimpress.nim:747-753).  With n_passes >= 2
 * the first whole-cohort run under NPS_MODE_AUTO -- which counts the tallies in its one read anyway -- keeps them with the
 * cohort as a by-product (no extra read), and every later run scores with the tallies given (0.75-0.78 of the roofline
 * instead of 0.62-0.73).  Without the hint NPS_MODE_AUTO does this only where the single-read kernel's grid covers less
 * than nine tenths of the compute units (see NPS_FMT_GT_AUTO).  0 or 1: no such caching (the default).  The hint is ignored
 * where it would not pay: cohorts of at most 262 144 samples (several row teams per strip: the given-tallies kernel is no
 * faster there than the pass that counts them). */
int nps_cohort_expect_passes(nps_cohort *c, uint32_t n_passes);
int nps_cohort_has_tallies(const nps_cohort *c); /* 1: the cohort carries whole-row tallies */

/* ---- measurement ------------------------------------------------------------------------ */
int nps_profile_enable(nps_ctx *ctx, int on); /* record HIP events around every launch */
int nps_profile_get(nps_ctx *ctx, nps_profile *out, int reset);
/* the HIP stream (hipStream_t) the context launches on, for callers that add their own events */
void *nps_stream(nps_ctx *ctx);
/* How NPS_MODE_AUTO would lay a resident run of n_rows rows of `format` over the chip: `slices`
 * workgroups side by side over the samples (samples_per_slice each) times `teams` taking row batches in
 * turn; all zero when the shape does not fit the persistent grid (two-pass kernels are used).  For
 * reports and for tests that want to look at every slice. */
/* (NPS_FMT_GT2X: the single-read kernel's OWN strips -- where a strip has one row team it cuts strips of 62 units = 1 984 samples
 * from the cohort's unit sequence instead of the layout's 2 048; with the tallies given or more strips than compute units, the
 * layout's strips) */
int nps_fused_geometry(nps_ctx *ctx, int format, uint64_t n_rows, uint32_t *slices, uint32_t *teams,
                       uint32_t *samples_per_slice);

#ifdef __cplusplus
}
#endif
#endif /* NPS_H */
