/*
 * oracle/refcpu.c -- CPU restatement of the nimpress per-variant hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check in
 * __graft_entry__.py and the cpu_baseline leg of bench.py may load it.  The product
 * (libnps.so) never links, loads or calls anything in this directory and has no CPU
 * fallback.
 *
 * Every function is a literal, single-threaded restatement of one proc of the
 * reference, `/root/reference/src/nimpress.nim` (cited as nimpress.nim:LINE below):
 * same loop order, same float64 operations, no FMA contraction (build with
 * -ffp-contract=off; see oracle/Makefile).  The reference cannot be compiled here
 * (no Nim, no htslib), so parity is pinned through the reference's own golden
 * vectors: tests/golden/set1_cases.json (nimpress tests/test_set1.nim:36-190, 13 cases)
 * and tests/golden/stats_kats.json (tests/test_stats.nim:21-139, 87 known answers).
 *
 * Third-party behaviour restated here (not in the reference tree):
 *   hts-nim (brentp/hts-nim >= 0.2.21, nimpress.nimble:15) `value(Allele)`:
 *     int32 < 0 -> returned as is; else (v >> 1) - 1.  So a missing allele (0) has
 *     value -1 and the htslib vector-end pad 0x80000001 has a large negative value
 *     that equals neither an allele index nor -1 (it is skipped, nimpress.nim:386-390).
 *   htslib (1.10.2, Dockerfile:32) bcf_get_genotypes layout: n_samples*ploidy int32,
 *     allele a encoded (a+1)<<1 | phased.
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* enum values follow declaration order of nimpress.nim:412-414 */
enum { REF_LOCUS_PS = 0, REF_LOCUS_HOMREF = 1, REF_LOCUS_FAIL = 2, REF_LOCUS_IGNORE = 3 };
enum { REF_MISSING_HOMREF = 0, REF_MISSING_IGNORE = 1 };
enum { REF_SAMPLE_PS = 0, REF_SAMPLE_HOMREF = 1, REF_SAMPLE_FAIL = 2, REF_SAMPLE_INT_PS = 3,
       REF_SAMPLE_INT_FAIL = 4 };

/* Row status as the host sees it before any genotype is touched (nimpress.nim:526-558). */
enum { REF_ROW_PRESENT = 0, REF_ROW_UNCOVERED = 1, REF_ROW_ABSENT = 2, REF_ROW_FILTERED = 3 };

/* Reason codes reported per row (what branch of getImputedDosages was taken). */
enum {
    REF_REASON_GENOTYPED = 0,   /* nimpress.nim:581-585 */
    REF_REASON_UNCOVERED = 1,   /* :526-531 */
    REF_REASON_ABSENT = 2,      /* :536-551 */
    REF_REASON_FILTERED = 3,    /* :553-558 */
    REF_REASON_MAXMIS = 4       /* :565-571 */
};

typedef struct {
    int32_t imp_locus;
    int32_t imp_missing;
    int32_t imp_sample;
    int32_t _pad;
    double max_missing_rate;
    int64_t min_cs;
} ref_params;

typedef struct {
    double ngenotyped;
    double nmissing;
    double neffect;
    int32_t used;
    int32_t reason;
} ref_locus_stat;

/* ------------------------------------------------------------------------- */
/* nimpress.nim:32-47  tallyAlleles                                          */
void ref_tally_alleles(const double *raw, size_t n, double *ngenotyped, double *nmissing,
                       double *neffect) {
    double ng = 0.0, nm = 0.0, ne = 0.0;
    for (size_t i = 0; i < n; ++i) {
        if (isnan(raw[i])) {
            nm += 1.0;
        } else {
            ng += 1.0;
            ne += raw[i];
        }
    }
    *ngenotyped = ng;
    *nmissing = nm;
    *neffect = ne;
}

/* hts-nim value(Allele) -- see header comment */
static inline int64_t allele_value(int32_t a) {
    if (a < 0) return (int64_t)a;
    return (int64_t)(a >> 1) - 1;
}

/* nimpress.nim:367-391  getRawDosages, from the bcf_get_genotypes int32 buffer */
void ref_raw_dosages_gt(double *raw, const int32_t *gts, size_t n, int ploidy, int eaidx) {
    for (size_t i = 0; i < n; ++i) {
        raw[i] = 0.0;
        for (int k = 0; k < ploidy; ++k) {
            int64_t v = allele_value(gts[i * (size_t)ploidy + (size_t)k]);
            if (v == (int64_t)eaidx)
                raw[i] += 1;
            else if (v == -1)
                raw[i] = NAN;
        }
    }
}

/* Build-defined extension (the reference has no DS path):
 * FORMAT/DS float32, one ALT dosage per sample; NaN (any NaN payload, incl. the BCF
 * missing 0x7F800001) = missing; effect allele == REF -> 2 - DS. */
void ref_raw_dosages_ds(double *raw, const float *ds, size_t n, int ref_is_effect) {
    for (size_t i = 0; i < n; ++i) {
        float d = ds[i];
        if (isnan(d))
            raw[i] = NAN;
        else
            raw[i] = ref_is_effect ? 2.0 - (double)d : (double)d;
    }
}

/* nimpress.nim:417-447  imputeLocusDosages; returns 0 = drop row, 1 = use row */
int ref_impute_locus(double *dos, size_t n, double eaf, int ref_is_effect, int method) {
    if (method == REF_LOCUS_IGNORE) return 0;
    double v;
    switch (method) {
    case REF_LOCUS_PS: v = eaf * 2.0; break;
    case REF_LOCUS_HOMREF: v = ref_is_effect ? 2.0 : 0.0; break;
    default: v = NAN; break;
    }
    for (size_t i = 0; i < n; ++i) dos[i] = v;
    return 1;
}

/* nimpress.nim:450-481  imputeSampleDosages */
void ref_impute_sample(double *dos, size_t n, double eaf, int ref_is_effect, double neffect,
                       double ngenotyped, int64_t min_cs, int method) {
    double v;
    switch (method) {
    case REF_SAMPLE_PS: v = eaf * 2.0; break;
    case REF_SAMPLE_HOMREF: v = ref_is_effect ? 2.0 : 0.0; break;
    case REF_SAMPLE_FAIL: v = NAN; break;
    default: /* int_ps, int_fail  :470-477 */
        if (ngenotyped >= (double)min_cs)
            v = neffect / ngenotyped;
        else
            v = (method == REF_SAMPLE_INT_PS) ? eaf * 2.0 : NAN;
        break;
    }
    for (size_t i = 0; i < n; ++i)
        if (isnan(dos[i])) dos[i] = v;
}

/* nimpress.nim:484-585  getImputedDosages, minus file access and warnings.
 * `status` is what the host found (REF_ROW_*); for REF_ROW_PRESENT `raw` must already hold
 * the raw dosages (NaN = missing) produced by ref_raw_dosages_gt / _ds.  On return `raw`
 * holds the (possibly imputed) dosages; the return value is the "use locus" bool. */
int ref_get_imputed_dosages(double *raw, size_t n, int status, double eaf, int ref_is_effect,
                            const ref_params *p, ref_locus_stat *st) {
    st->ngenotyped = 0.0;
    st->nmissing = 0.0;
    st->neffect = 0.0;
    if (status == REF_ROW_UNCOVERED) { /* :526-531 */
        st->reason = REF_REASON_UNCOVERED;
        st->used = ref_impute_locus(raw, n, eaf, ref_is_effect, p->imp_locus);
        return st->used;
    }
    if (status == REF_ROW_ABSENT) { /* :536-551 */
        st->reason = REF_REASON_ABSENT;
        if (p->imp_missing == REF_MISSING_HOMREF) {
            double v = ref_is_effect ? 2.0 : 0.0;
            for (size_t i = 0; i < n; ++i) raw[i] = v;
            st->used = 1;
        } else {
            st->used = 0;
        }
        return st->used;
    }
    if (status == REF_ROW_FILTERED) { /* :553-558 */
        st->reason = REF_REASON_FILTERED;
        st->used = ref_impute_locus(raw, n, eaf, ref_is_effect, p->imp_locus);
        return st->used;
    }
    /* :561-563 */
    ref_tally_alleles(raw, n, &st->ngenotyped, &st->nmissing, &st->neffect);
    /* :565-571  strict >, double division */
    double missingrate = st->nmissing / (double)n;
    if (missingrate > p->max_missing_rate) {
        st->reason = REF_REASON_MAXMIS;
        st->used = ref_impute_locus(raw, n, eaf, ref_is_effect, p->imp_locus);
        return st->used;
    }
    /* :582-585 */
    ref_impute_sample(raw, n, eaf, ref_is_effect, st->neffect, st->ngenotyped, p->min_cs,
                      p->imp_sample);
    st->reason = REF_REASON_GENOTYPED;
    st->used = 1;
    return 1;
}

/* ------------------------------------------------------------------------- */
/* Score accumulation state: nimpress.nim:592-649 computePolygenicScores, unrolled into
 * begin / per-row / finish so that a driver can feed rows one at a time. */
typedef struct {
    size_t n;
    double *scores;  /* :626-628 */
    double *dosages; /* :633 */
    int64_t nloci;   /* :632 */
    ref_params p;
} ref_state;

ref_state *ref_begin(size_t n, const ref_params *p) {
    ref_state *s = (ref_state *)calloc(1, sizeof(ref_state));
    s->n = n;
    s->scores = (double *)malloc(sizeof(double) * (n ? n : 1));
    s->dosages = (double *)malloc(sizeof(double) * (n ? n : 1));
    for (size_t i = 0; i < n; ++i) s->scores[i] = 0.0;
    s->nloci = 0;
    s->p = *p;
    return s;
}

static void ref_accumulate(ref_state *s, double beta) { /* :639-641 */
    for (size_t i = 0; i < s->n; ++i) s->scores[i] += s->dosages[i] * beta;
    s->nloci += 1;
}

/* One score row with GT data (status PRESENT) */
void ref_row_gt(ref_state *s, const int32_t *gts, int ploidy, int eaidx, int ref_is_effect,
                double beta, double eaf, ref_locus_stat *st) {
    ref_raw_dosages_gt(s->dosages, gts, s->n, ploidy, eaidx);
    if (ref_get_imputed_dosages(s->dosages, s->n, REF_ROW_PRESENT, eaf, ref_is_effect, &s->p, st))
        ref_accumulate(s, beta);
}

void ref_row_ds(ref_state *s, const float *ds, int ref_is_effect, double beta, double eaf,
                ref_locus_stat *st) {
    ref_raw_dosages_ds(s->dosages, ds, s->n, ref_is_effect);
    if (ref_get_imputed_dosages(s->dosages, s->n, REF_ROW_PRESENT, eaf, ref_is_effect, &s->p, st))
        ref_accumulate(s, beta);
}

/* One score row given as already-decoded raw dosages (NaN = missing) */
void ref_row_raw(ref_state *s, const double *raw, int ref_is_effect, double beta, double eaf,
                 ref_locus_stat *st) {
    memcpy(s->dosages, raw, sizeof(double) * s->n);
    if (ref_get_imputed_dosages(s->dosages, s->n, REF_ROW_PRESENT, eaf, ref_is_effect, &s->p, st))
        ref_accumulate(s, beta);
}

/* One score row without genotype data: status = UNCOVERED / ABSENT / FILTERED */
void ref_row_locus(ref_state *s, int status, int ref_is_effect, double beta, double eaf,
                   ref_locus_stat *st) {
    if (ref_get_imputed_dosages(s->dosages, s->n, status, eaf, ref_is_effect, &s->p, st))
        ref_accumulate(s, beta);
}

/* the un-normalised sums and nloci so far (state of the loop at :641); does not free the state.  Used to
 * check the row-sharded multi-GPU path, whose exchange happens before the normalisation. */
void ref_partial(const ref_state *s, double *sums_out, int64_t *nloci_out) {
    if (sums_out) memcpy(sums_out, s->scores, sizeof(double) * s->n);
    if (nloci_out) *nloci_out = s->nloci;
}

/* :643-649; copies out scores and nloci, frees the state */
void ref_finish(ref_state *s, double offset, double *scores_out, int64_t *nloci_out) {
    for (size_t i = 0; i < s->n; ++i) s->scores[i] /= (double)s->nloci * 2.0;
    for (size_t i = 0; i < s->n; ++i) s->scores[i] += offset;
    if (scores_out) memcpy(scores_out, s->scores, sizeof(double) * s->n);
    if (nloci_out) *nloci_out = s->nloci;
    free(s->scores);
    free(s->dosages);
    free(s);
}

/* ------------------------------------------------------------------------- */
/* Whole-matrix convenience for the synthetic cohorts: rows are 2-bit codes
 * (0 = dosage 0, 1 = dosage 1, 3 = dosage 2, 2 = missing), 16 per little-endian uint32, row stride
 * `stride_words` -- the build-defined device layout.  Unpacks each row to the
 * bcf_get_genotypes int32 layout and runs the literal per-row path above, so this measures
 * the reference's own passes (decode, tally, impute, accumulate). */
void ref_codes_to_gt(const uint32_t *row, size_t n, int32_t *gts /* 2n */) {
    for (size_t i = 0; i < n; ++i) {
        unsigned c = (row[i >> 4] >> ((i & 15) * 2)) & 3u;
        /* effect allele index 1, other allele 0:  (a+1)<<1 */
        int32_t a0, a1;
        switch (c) {
        case 0: a0 = 2; a1 = 2; break;  /* 0/0 */
        case 1: a0 = 2; a1 = 4; break;  /* 0/1 */
        case 3: a0 = 4; a1 = 4; break;  /* 1/1 */
        default: a0 = 0; a1 = 0; break; /* ./.  (code 2) */
        }
        gts[2 * i] = a0;
        gts[2 * i + 1] = a1;
    }
}

/* kind[j]: REF_ROW_* ; rows with kind PRESENT consume the next packed row in order. */
void ref_score_packed(const uint32_t *codes, size_t stride_words, size_t n, size_t m,
                      const int32_t *kind, const int32_t *ref_is_effect, const double *beta,
                      const double *eaf, const ref_params *p, double offset, double *scores_out,
                      ref_locus_stat *stats_out, int64_t *nloci_out) {
    ref_state *s = ref_begin(n, p);
    int32_t *gts = (int32_t *)malloc(sizeof(int32_t) * 2 * (n ? n : 1));
    size_t r = 0;
    for (size_t j = 0; j < m; ++j) {
        ref_locus_stat st;
        if (kind[j] == REF_ROW_PRESENT) {
            ref_codes_to_gt(codes + r * stride_words, n, gts);
            ++r;
            ref_row_gt(s, gts, 2, 1, ref_is_effect[j], beta[j], eaf[j], &st);
        } else {
            ref_row_locus(s, kind[j], ref_is_effect[j], beta[j], eaf[j], &st);
        }
        if (stats_out) stats_out[j] = st;
    }
    free(gts);
    ref_finish(s, offset, scores_out, nloci_out);
}

/* ------------------------------------------------------------------------- */
/* Stats helpers: nimpress.nim:50-188 (used only for AF-mismatch warnings).   */

/* :51 */
static double ref_lbinom(int64_t n, int64_t k) {
    return lgamma((double)n + 1.0) - lgamma((double)k + 1.0) - lgamma((double)(n - k) + 1.0);
}

/* :54-60 */
double ref_dbinom(int64_t x, int64_t n, double p) {
    if ((x == 0 && p == 0.0) || (x == n && p == 1.0)) return 1.0;
    return exp(ref_lbinom(n, x) + (double)x * log(p) + (double)(n - x) * log(1.0 - p));
}

/* :63-117 */
static double ref_betacf(double a, double b, double x) {
    const double qab = a + b, qap = a + 1.0, qam = a - 1.0;
    const double FPMIN = 1.0e-30, EPS = 3.0e-7;
    const int MAXIT = 100;
    double c = 1.0;
    double d = 1.0 - qab * x / qap;
    if (fabs(d) < FPMIN) d = FPMIN;
    d = 1.0 / d;
    double result = d;
    for (int m = 1; m <= MAXIT; ++m) {
        double mf = (double)m;
        double aa1 = mf * (b - mf) * x / ((qam + 2 * mf) * (a + 2 * mf));
        d = 1.0 + aa1 * d;
        if (fabs(d) < FPMIN) d = FPMIN;
        c = 1.0 + aa1 / c;
        if (fabs(c) < FPMIN) c = FPMIN;
        d = 1.0 / d;
        result *= d * c;
        double aa2 = -(a + mf) * (qab + mf) * x / ((a + 2 * mf) * (qap + 2 * mf));
        d = 1.0 + aa2 * d;
        if (fabs(d) < FPMIN) d = FPMIN;
        c = 1.0 + aa2 / c;
        if (fabs(c) < FPMIN) c = FPMIN;
        d = 1.0 / d;
        double del = d * c;
        result *= del;
        if (fabs(del - 1.0) < EPS) return result;
    }
    return NAN;
}

/* :120-134 ; the reference asserts 0 <= x <= 1, here NaN is returned instead of aborting */
double ref_betai(double a, double b, double x) {
    if (!(x >= 0.0 && x <= 1.0)) return NAN;
    if (a == 0.0 || b == 0.0) return INFINITY;
    if (x == 0.0) return 0.0;
    if (x == 1.0) return 1.0;
    double bt = exp(lgamma(a + b) - lgamma(a) - lgamma(b) + a * log(x) + b * log(1.0 - x));
    if (x < (a + 1.0) / (a + b + 2.0)) return bt * ref_betacf(a, b, x) / a;
    return 1.0 - bt * ref_betacf(b, a, 1.0 - x) / b;
}

/* :138-152 */
double ref_pbinom(int64_t x, int64_t n, double p) {
    if (x < 0) return 0.0;
    if (x == n) return 1.0;
    return 1.0 - ref_betai((double)x + 1.0, (double)(n - x), p);
}

/* :155-188 */
double ref_binom_test(int64_t x, int64_t n, double p) {
    if (p == 0.0) return x == 0 ? 1.0 : 0.0;
    if (p == 1.0) return x == n ? 1.0 : 0.0;
    double probx = ref_dbinom(x, n, p);
    double expected = (double)n * p;
    if (fabs((double)x / expected - 1.0) < 1.0e-6) return 1.0;
    if ((double)x < expected) {
        int64_t y = 0;
        for (int64_t xi = (int64_t)ceil(expected); xi <= n; ++xi)
            if (ref_dbinom(xi, n, p) <= probx * (1.0 + 1.0e-7)) y += 1;
        return ref_pbinom(x, n, p) + (1.0 - ref_pbinom(n - y, n, p));
    } else {
        int64_t y = 0;
        for (int64_t xi = 0; xi <= (int64_t)floor(expected); ++xi)
            if (ref_dbinom(xi, n, p) <= probx * (1.0 + 1.0e-7)) y += 1;
        return ref_pbinom(y - 1, n, p) + (1.0 - ref_pbinom(x - 1, n, p));
    }
}

/* ------------------------------------------------------------------------- */
/* nimpress.nim:310-311 + 313-345: BED containment predicate (lapper is only an index). */
int ref_interval_contains(int64_t start0, int64_t end1, int64_t pos, int64_t stop) {
    return start0 < pos && end1 >= stop;
}

/* ------------------------------------------------------------------------- */
/* Synthetic cohort generator shared with the device (DESIGN.md "Synthetic cohorts").
 * Counter-based: code(seed,row,sample) depends on nothing else, so any tile can be
 * regenerated anywhere.  Thresholds are integers computed by the caller, so CPU and GPU
 * agree bit for bit.  This is the oracle's own copy of the 20-line generator; the device
 * copy lives in nimpress_amd/csrc/nps_synth.hip. */
static inline uint64_t ref_mix64(uint64_t z) { /* splitmix64 finaliser */
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

unsigned ref_synth_code(uint64_t seed, uint64_t row, uint64_t sample, uint32_t t_het,
                        uint32_t t_hom, uint32_t t_miss) {
    uint64_t h = ref_mix64(ref_mix64(seed ^ (row * 0xD1B54A32D192ED03ull)) + sample);
    uint32_t g = (uint32_t)h, ms = (uint32_t)(h >> 32);
    if (ms < t_miss) return 2u; /* missing */
    /* g < t_hom -> dosage 2 (code 3) ; g < t_het -> dosage 1 ; else dosage 0   (t_hom <= t_het) */
    return g < t_hom ? 3u : (g < t_het ? 1u : 0u);
}

void ref_synth_rows(uint32_t *codes, size_t stride_words, size_t n, size_t row0, size_t nrows,
                    uint64_t seed, const uint32_t *t_het, const uint32_t *t_hom,
                    const uint32_t *t_miss) {
    for (size_t r = 0; r < nrows; ++r) {
        uint32_t *row = codes + r * stride_words;
        memset(row, 0, sizeof(uint32_t) * stride_words);
        for (size_t i = 0; i < n; ++i) {
            unsigned c = ref_synth_code(seed, row0 + r, i, t_het[r], t_hom[r], t_miss[r]);
            row[i >> 4] |= (uint32_t)c << ((i & 15) * 2);
        }
    }
}

float ref_synth_ds(uint64_t seed, uint64_t row, uint64_t sample, uint32_t t_het, uint32_t t_hom,
                   uint32_t t_miss) {
    /* DS = genotype + noise in [-0.125, 0.125) on a 1/1024 grid, clipped to [0,2]; NaN = missing */
    uint64_t h = ref_mix64(ref_mix64(seed ^ (row * 0xD1B54A32D192ED03ull)) + sample);
    uint32_t g = (uint32_t)h, ms = (uint32_t)(h >> 32);
    if (ms < t_miss) return NAN;
    int c = g < t_hom ? 2 : (g < t_het ? 1 : 0);
    int noise = (int)((ms >> 8) & 255u) - 128; /* [-128,127] */
    float d = (float)c + (float)noise * (1.0f / 1024.0f);
    if (d < 0.0f) d = 0.0f;
    if (d > 2.0f) d = 2.0f;
    return d;
}

/* NPS_FMT_DS16 cohorts (round 5): the same draw, with a DECIMAL noise -- dosage = genotype + noise * 0.004, clipped to
 * [0, 2], i.e. k / 10^4 with k = clamp(10000 c + 40 noise, 0, 20000) -- as the float32 a parser makes of that decimal
 * text: (float)((double)k * 1e-4) (tests/test_host_logic.py checks that against strtof for every k). */
float ref_ds16_value(uint32_t k) { return k == 0xffffu ? NAN : (float)((double)k * 1e-4); }
uint32_t ref_synth_ds16_code(uint64_t seed, uint64_t row, uint64_t sample, uint32_t t_het, uint32_t t_hom,
                             uint32_t t_miss) {
    uint64_t h = ref_mix64(ref_mix64(seed ^ (row * 0xD1B54A32D192ED03ull)) + sample);
    uint32_t g = (uint32_t)h, ms = (uint32_t)(h >> 32);
    if (ms < t_miss) return 0xffffu;
    int c = g < t_hom ? 2 : (g < t_het ? 1 : 0);
    int noise = (int)((ms >> 8) & 255u) - 128;
    int v = c * 10000 + noise * 40;
    return (uint32_t)(v < 0 ? 0 : (v > 20000 ? 20000 : v));
}
static float synth_ds_kind(int kind, uint64_t seed, uint64_t row, uint64_t sample, uint32_t t_het, uint32_t t_hom,
                           uint32_t t_miss) { /* kind 1: ref_synth_ds; 2: the NPS_FMT_DS16 generator */
    return kind == 2 ? ref_ds16_value(ref_synth_ds16_code(seed, row, sample, t_het, t_hom, t_miss))
                     : ref_synth_ds(seed, row, sample, t_het, t_hom, t_miss);
}
void ref_synth_rows_ds16(float *ds, size_t stride, size_t n, size_t row0, size_t nrows, uint64_t seed,
                         const uint32_t *t_het, const uint32_t *t_hom, const uint32_t *t_miss) {
    for (size_t r = 0; r < nrows; ++r)
        for (size_t i = 0; i < n; ++i)
            ds[r * stride + i] = synth_ds_kind(2, seed, row0 + r, i, t_het[r], t_hom[r], t_miss[r]);
}

void ref_synth_rows_ds(float *ds, size_t stride, size_t n, size_t row0, size_t nrows,
                       uint64_t seed, const uint32_t *t_het, const uint32_t *t_hom,
                       const uint32_t *t_miss) {
    for (size_t r = 0; r < nrows; ++r)
        for (size_t i = 0; i < n; ++i)
            ds[r * stride + i] =
                ref_synth_ds(seed, row0 + r, i, t_het[r], t_hom[r], t_miss[r]);
}

/* ------------------------------------------------------------------------- */
/* Full-size checks (bench.py score_delta leg, tests/test_gpu_parity.py): a SUBSET of the samples of a
 * synthetic cohort scored over ALL its rows with the restated procs above.
 *
 * In the reference the per-sample work of a row depends on the other samples only through the row's
 * tally: imputeSampleDosages receives (neffectallele, ngenotyped) as arguments (nimpress.nim:450-452,
 * call at :582-583) and the maxmis decision is nmissing / nsamples (:565).  So the k chosen samples
 * can be scored exactly as the reference scores them inside the whole cohort, given every row's
 * whole-row tally -- which ref_tally_synth_rows recounts literally (decode :367-391 + tally :32-47
 * over all n samples) for as many rows as the caller wants to pay for; for the other rows the caller
 * passes the tallies the device reported.  OpenMP only splits the SAMPLES (every sample keeps the
 * reference's row order and operations) or the ROWS of the recount. */
static void subset_row_gt(double *dos, int32_t *gts, const uint64_t *samples, size_t k, uint64_t seed,
                          uint64_t row, uint32_t th, uint32_t tm, uint32_t tmi) {
    for (size_t i = 0; i < k; ++i) {
        const unsigned c = ref_synth_code(seed, row, samples[i], th, tm, tmi);
        /* the bcf_get_genotypes pair of ref_codes_to_gt: effect allele index 1 */
        gts[2 * i] = (c == 3u) ? 4 : (c == 2u ? 0 : 2);
        gts[2 * i + 1] = (c == 0u) ? 2 : (c == 2u ? 0 : 4);
    }
    ref_raw_dosages_gt(dos, gts, k, 2, 1);
}

/* the decision chain of getImputedDosages :565-583 for a PRESENT row whose whole-row tally is given */
static int subset_impute(double *dos, size_t k, size_t n_total, double ngen, double nmiss, double neff,
                         double eaf, int rie, const ref_params *p) {
    const double missingrate = nmiss / (double)n_total; /* :565 */
    if (missingrate > p->max_missing_rate) return ref_impute_locus(dos, k, eaf, rie, p->imp_locus);
    ref_impute_sample(dos, k, eaf, rie, neff, ngen, p->min_cs, p->imp_sample);
    return 1;
}

void ref_score_subset(int is_ds, const uint64_t *samples, size_t k, size_t n_total, size_t row0, size_t m,
                      uint64_t seed, const uint32_t *t_het, const uint32_t *t_hom, const uint32_t *t_miss,
                      const double *beta, const double *eaf, const int32_t *rie, const double *row_ngen,
                      const double *row_nmiss, const double *row_neff, const ref_params *p, int threads,
                      double *sums_out /* k un-normalised sums */, int64_t *nloci_out) {
    int64_t nloci = 0;
    for (size_t j = 0; j < m; ++j) { /* which rows are used does not depend on the sample */
        const double missingrate = row_nmiss[j] / (double)n_total;
        if (!(missingrate > p->max_missing_rate) || p->imp_locus != REF_LOCUS_IGNORE) nloci += 1;
    }
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        int nt = 1, me = 0;
#ifdef _OPENMP
        nt = omp_get_num_threads();
        me = omp_get_thread_num();
#endif
        const size_t per = (k + (size_t)nt - 1) / (size_t)nt;
        const size_t i0 = (size_t)me * per < k ? (size_t)me * per : k;
        const size_t i1 = i0 + per < k ? i0 + per : k;
        const size_t kk = i1 - i0;
        if (kk) {
            double *dos = (double *)malloc(sizeof(double) * kk);
            int32_t *gts = (int32_t *)malloc(sizeof(int32_t) * 2 * kk);
            float *ds = (float *)malloc(sizeof(float) * kk);
            double *sc = sums_out + i0;
            for (size_t i = 0; i < kk; ++i) sc[i] = 0.0;
            for (size_t j = 0; j < m; ++j) {
                if (is_ds) {
                    for (size_t i = 0; i < kk; ++i)
                        ds[i] = synth_ds_kind(is_ds, seed, row0 + j, samples[i0 + i], t_het[j], t_hom[j], t_miss[j]);
                    ref_raw_dosages_ds(dos, ds, kk, rie[j]);
                } else {
                    subset_row_gt(dos, gts, samples + i0, kk, seed, row0 + j, t_het[j], t_hom[j], t_miss[j]);
                }
                if (subset_impute(dos, kk, n_total, row_ngen[j], row_nmiss[j], row_neff[j], eaf[j], rie[j], p))
                    for (size_t i = 0; i < kk; ++i) sc[i] += dos[i] * beta[j]; /* :639-640 */
            }
            free(dos);
            free(gts);
            free(ds);
        }
    }
    if (nloci_out) *nloci_out = nloci;
}

/* literal whole-row recount (decode + tallyAlleles over all n samples) of selected rows */
void ref_tally_synth_rows(int is_ds, const uint64_t *rows, size_t nr, size_t n, uint64_t seed,
                          const uint32_t *t_het, const uint32_t *t_hom, const uint32_t *t_miss,
                          const int32_t *rie /* per selected row */, int threads, double *ngen,
                          double *nmiss, double *neff) {
    if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
    {
        double *dos = (double *)malloc(sizeof(double) * (n ? n : 1));
        int32_t *gts = is_ds ? NULL : (int32_t *)malloc(sizeof(int32_t) * 2 * (n ? n : 1));
        float *ds = is_ds ? (float *)malloc(sizeof(float) * (n ? n : 1)) : NULL;
        uint64_t *ids = (uint64_t *)malloc(sizeof(uint64_t) * (n ? n : 1));
        for (size_t i = 0; i < n; ++i) ids[i] = i;
#pragma omp for schedule(dynamic, 1)
        for (size_t r = 0; r < nr; ++r) {
            if (is_ds) {
                for (size_t i = 0; i < n; ++i) ds[i] = synth_ds_kind(is_ds, seed, rows[r], i, t_het[r], t_hom[r], t_miss[r]);
                ref_raw_dosages_ds(dos, ds, n, rie[r]);
            } else {
                subset_row_gt(dos, gts, ids, n, seed, rows[r], t_het[r], t_hom[r], t_miss[r]);
            }
            ref_tally_alleles(dos, n, &ngen[r], &nmiss[r], &neff[r]);
        }
        free(dos);
        free(gts);
        free(ds);
        free(ids);
    }
}

/* ------------------------------------------------------------------------- */
/* CPU baseline for bench.py: the literal per-row path (decode, tally, impute, accumulate;
 * nimpress.nim:561-583, 639-641) over m rows whose bcf_get_genotypes buffers cycle through
 * n_distinct pre-built rows.  Returns wall seconds measured around the loop only.
 * with_binomtest != 0 adds what the reference ALWAYS executes between the maxmis decision and the
 * sample imputation (nimpress.nim:573): binomTest(neffect, 2*ngenotyped, eaf), the O(N) enumeration of
 * :155-188, whose result only gates a warning (*warned_out counts p < afmisp). */
#include <time.h>
static double ref_now(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

double ref_bench_gt_full(const int32_t *gts_rows, size_t n_distinct, size_t n, size_t m,
                         const double *beta, const double *eaf, const ref_params *p,
                         int with_binomtest, double afmisp, double *scores_out, int64_t *nloci_out,
                         int64_t *warned_out) {
    ref_state *s = ref_begin(n, p);
    int64_t warned = 0;
    const double t0 = ref_now();
    for (size_t j = 0; j < m; ++j) {
        ref_locus_stat st;
        const int32_t *gts = gts_rows + (j % n_distinct) * 2 * n;
        if (!with_binomtest) {
            ref_row_gt(s, gts, 2, 1, 0, beta[j], eaf[j], &st);
            continue;
        }
        /* the same row with the binomTest call in its place (:561-583) */
        ref_raw_dosages_gt(s->dosages, gts, n, 2, 1);
        ref_tally_alleles(s->dosages, n, &st.ngenotyped, &st.nmissing, &st.neffect);
        const double missingrate = st.nmissing / (double)n;
        if (missingrate > p->max_missing_rate) {
            if (ref_impute_locus(s->dosages, n, eaf[j], 0, p->imp_locus)) ref_accumulate(s, beta[j]);
            continue;
        }
        if (!isnan(eaf[j]) &&
            ref_binom_test((int64_t)st.neffect, ((int64_t)n - (int64_t)st.nmissing) * 2, eaf[j]) < afmisp)
            warned += 1;
        ref_impute_sample(s->dosages, n, eaf[j], 0, st.neffect, st.ngenotyped, p->min_cs,
                          p->imp_sample);
        ref_accumulate(s, beta[j]);
    }
    const double t1 = ref_now();
    ref_finish(s, 0.0, scores_out, nloci_out);
    if (warned_out) *warned_out = warned;
    return t1 - t0;
}

double ref_bench_gt(const int32_t *gts_rows, size_t n_distinct, size_t n, size_t m,
                    const double *beta, const double *eaf, const ref_params *p,
                    double *scores_out, int64_t *nloci_out) {
    return ref_bench_gt_full(gts_rows, n_distinct, n, m, beta, eaf, p, 0, 0.0, scores_out, nloci_out,
                             NULL);
}

/* The "whole socket" figure BASELINE.md asks for beside the single-threaded one: the same four passes
 * per row (decode :383-391, tally :41-46, impute :479-481 / :444-445, accumulate :639-640), each split
 * over the samples with OpenMP.  NOT a restatement of the reference (which has one thread): tallies are
 * integer-valued, so the reduction order does not change them, and every sample's score is still the
 * same sequence of operations.  Default CLI methods only (imp-locus ps, imp-sample int_ps).  The thread
 * count is an argument (libgomp reads OMP_NUM_THREADS once, when it is first loaded -- usually by numpy or
 * torch long before).  Returns seconds; *threads_out = threads used. */
double ref_bench_gt_allcores(const int32_t *gts_rows, size_t n_distinct, size_t n, size_t m,
                             const double *beta, const double *eaf, const ref_params *p, int threads,
                             double *scores_out, int64_t *nloci_out, int *threads_out) {
    double *scores = (double *)malloc(sizeof(double) * (n ? n : 1));
    double *dos = (double *)malloc(sizeof(double) * (n ? n : 1));
    int64_t nloci = 0;
    if (threads < 1) threads = 1;
#ifndef _OPENMP
    threads = 1;
#endif
#pragma omp parallel for schedule(static) num_threads(threads)
    for (size_t i = 0; i < n; ++i) scores[i] = 0.0;
    const double t0 = ref_now();
    for (size_t j = 0; j < m; ++j) {
        const int32_t *gts = gts_rows + (j % n_distinct) * 2 * n;
        double ngen = 0.0, nmiss = 0.0, neff = 0.0;
#pragma omp parallel for schedule(static) num_threads(threads)
        for (size_t i = 0; i < n; ++i) { /* :383-391 */
            double d = 0.0;
            for (int a = 0; a < 2; ++a) {
                const int64_t v = allele_value(gts[2 * i + a]);
                if (v == 1) d += 1.0;
                else if (v == -1) d = NAN;
            }
            dos[i] = d;
        }
#pragma omp parallel for schedule(static) num_threads(threads) reduction(+ : ngen, nmiss, neff)
        for (size_t i = 0; i < n; ++i) { /* :41-46 */
            if (isnan(dos[i])) nmiss += 1.0;
            else { ngen += 1.0; neff += dos[i]; }
        }
        const double missingrate = nmiss / (double)n;
        double fill;
        int all;
        if (missingrate > p->max_missing_rate) { /* :565-571, imp-locus ps */
            fill = eaf[j] * 2.0;
            all = 1;
        } else { /* :470-477, imp-sample int_ps */
            fill = ngen >= (double)p->min_cs ? neff / ngen : eaf[j] * 2.0;
            all = 0;
        }
        const double b = beta[j];
#pragma omp parallel for schedule(static) num_threads(threads)
        for (size_t i = 0; i < n; ++i) { /* :479-481 / :444-445, then :639-640 */
            const double d = (all || isnan(dos[i])) ? fill : dos[i];
            scores[i] += d * b;
        }
        nloci += 1;
    }
    const double t1 = ref_now();
    for (size_t i = 0; i < n; ++i) scores[i] = scores[i] / ((double)nloci * 2.0);
    if (scores_out) memcpy(scores_out, scores, sizeof(double) * n);
    if (nloci_out) *nloci_out = nloci;
    if (threads_out) *threads_out = threads;
    free(scores);
    free(dos);
    return t1 - t0;
}
