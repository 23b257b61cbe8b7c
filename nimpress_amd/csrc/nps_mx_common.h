// nps_mx_common.h -- what the strip kernels (nps_mx.hip: tallies in the pass; nps_mxg.hip: tallies given) share: argument
// block, the per-row precomputed part, weight -> FP6 operand bytes, the per-row decisions.  Internal (not part of the C-ABI).
#pragma once
#include "nps_kernels.h"

namespace nps {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v3i __attribute__((ext_vector_type(3)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned long long v2ul __attribute__((ext_vector_type(2)));
#define NPS_LDS __attribute__((address_space(3)))

// Data waves 0..5 carry 9 units each, the control waves 6, 7 five units each plus the per-row work of 64 rows each.
// The control chain: the tables of superblock k are made in step k, in front of the step's barrier -- the look at the
// rows' tally words is issued at the start of the step and travels while the control wave tallies and parks its own
// units; the operands follow when it is back.  History: until the end of round 4 the look was issued,
// waited for and turned into operands BEFORE the wave's own tallying (24.3 -> 22.5-23.2 ms at 245 strips).  Two other
// schedules existed as template instantiations (a run-time flag with both paths in one kernel cost 20 %): "early" --
// the tables of k + 1 made during the second half of step k, the look issued with the returning add of the
// publication of k + 2 (three table buffers) -- which won below 96 strips per team over the old order and loses to the
// present one at every size (ms per 1M rows, present / early: 29 strips 2.65 / 2.70, 49 strips 4.23 / 4.33, 64 strips
// 5.34 / 5.56, 98 strips 8.94 / 9.72, 245 strips 23.4 / 27.6), and "mid" (the look a third of a step earlier).  Both
// are gone.  (Control waves without units -- 11 / 10 units per data wave -- need 273 VGPRs: 136 spilled.  Keeping the
// first stage of the publication inside an XCD's L2 is not possible: workgroup- and agent-scope atomics are the same
// instruction on gfx950 -- sc1 only selects system scope -- and execute at the memory side.)
#ifndef NPS_MX_DW
#define NPS_MX_DW 6   // data waves; NPS_MX_UD units each, the two control waves NPS_MX_UC each: 64 in all
#define NPS_MX_UD 9
#define NPS_MX_UC 5
#endif
constexpr int kDW = NPS_MX_DW;           // data waves 0..kDW-1; the two control waves follow
constexpr int kMxThreads = (kDW + 2) * 64;
constexpr int kUD = NPS_MX_UD;
constexpr int kUC = NPS_MX_UC;           // units of the two control waves, which do the per-row work of 64 rows each
constexpr int kTabBufs = 2;
constexpr uint32_t kFlushSb = 1024;      // superblocks between flushes of the float32 digit sums (131 072 rows x 75 < 2^24)
constexpr uint32_t kLdsTables = 131072;  // [kTabBufs][3 operands][128 rows][16 bytes]
constexpr uint32_t kLdsTally = kLdsTables + kTabBufs * 6144;  // [2][128] uint32: nmissing << 16 | neffect of the strip
constexpr uint32_t kLdsBytes = kLdsTally + 1024;
constexpr uint32_t kMxSpinLimit = 1u << 20;

struct MxPre;
struct MxArgs {
    const v4u *units;        // the cohort
    uint64_t n_sb_cohort;    // its superblocks (a strip is n_sb_cohort * units-of-the-strip KiB)
    uint32_t sb0, n_sb;      // this run: first superblock, superblocks
    uint64_t n_rows;         // rows of this run
    uint64_t n_samples;
    uint32_t P, nu_last;     // strips, units of the last one
    // (round 5) virtual strips of the first form: the workgroups' strips are U <= 64 consecutive units of the cohort's unit
    // sequence, whatever 64-unit strip of the LAYOUT they lie in (a wave's units cross at most one layout boundary): P and
    // nu_last above count THEM; the layout's own numbers are below.  U = 64: the strips are the layout's.
    uint32_t U = 64, P_phys = 0, nu_last_phys = 0;
    uint32_t Q;              // row teams per strip: superblock k of the run belongs to team k % Q (grid = Q * P workgroups)
    const nps_row_desc *desc;
    const MxPre *pre;        // per score row, from mx_prep_kernel: what does not depend on the tallies
    DevParams prm;
    int64_t t_maxmis;        // the largest nmissing for which nmissing / N > --maxmis is false (-1: none)
    double scale;            // 2^F
    unsigned long long *tally;  // [n_sb * 128], zero on entry (GIVEN: the complete whole-row tallies, from mx_tally_kernel)
    unsigned long long *tally1;  // [groups of grp_strips strips][n_sb * 128] (sized for groups of 16), zero on entry: first stage of the hand-over
    nps_locus_stat *stats;
    unsigned long long *nloci;
    double *const_sum;       // [2 Q], zero on entry: slot 2 team + control wave = the locus constants of that wave's rows over
                             // --maxmis (plain stores; mx_fold_kernel adds the slots in fixed order: bit-reproducible whatever
                             // order the teams finish in, which a float atomicAdd per team was not)
    float *cpart;            // [n_flush][Q][P][64][2][256]
    unsigned int *timeout;
    uint32_t ctl_prio;       // the control waves run at raised issue priority (see fused_mx_kernel)
    uint32_t grp_strips;     // strips per first-stage group of the hand-over (16 .. 64: tally1 is sized for groups of 16)
};
static __device__ __forceinline__ v2i tr4(const char *p) {
    return __builtin_amdgcn_ds_read_tr4_b64_v2i32((NPS_LDS v2i *)p);
}
static __device__ __forceinline__ v3i tr6(const char *p) {
    return __builtin_amdgcn_ds_read_tr6_b96_v3i32((NPS_LDS v3i *)p);
}
static __device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {  // popcount(x) + acc, one instruction
    uint32_t d;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(acc));
    return d;
}
// where row r of a unit lives in its 1 KiB LDS image: a lane's two rows stay together (one ds_write_b128), and
// the 32 lanes of a half wave read 256 different bytes in both transposed reads
static __host__ __device__ inline int mx_rowoff(int r) {
    return 8 * (r & 15) + 128 * ((r >> 5) & 1) + 256 * ((r >> 4) & 1) + 512 * (r >> 6);
}

// four 4-bit fields -> four 6-bit fields
static __device__ __forceinline__ uint32_t spread4(uint32_t x) {
    x = (x & 0x00FFu) | ((x & 0xFF00u) << 4);
    x = (x & 0x00F00Fu) | ((x & 0x0F00F0u) << 2);
    return x;
}
// an integer weight, |w| < 2^56, as sixteen FP6 operand bytes packed 6 bits apart: fourteen hexadecimal digits of
// |w| (an e2m3 byte 00dddd is d/8: the subnormals and the first binade are one linear run), the sign bit in every
// digit, a spare column and the flag column
static __device__ __forceinline__ void mx_codes(long long w, uint32_t flag, uint32_t (&c)[3]) {
    const unsigned long long aw = (unsigned long long)(w < 0 ? -w : w);
    const uint32_t lo = (uint32_t)aw, hi = (uint32_t)(aw >> 32);
    const uint32_t c0 = spread4(lo & 0xFFFFu), c1 = spread4(lo >> 16), c2 = spread4(hi & 0xFFFFu), c3 = spread4((hi >> 16) & 0xFFu);
    c[0] = c0 | (c1 << 24);
    c[1] = (c1 >> 8) | (c2 << 16);
    c[2] = (c2 >> 16) | (c3 << 8) | (flag << 26);
    const uint32_t neg = (uint32_t)(w >> 63);  // all ones for a negative weight: the sign bit of every digit
    c[0] |= neg & 0x20820820u;
    c[1] |= neg & 0x08208208u;
    c[2] |= neg & 0x02082082u;
}

// Per score row, everything that does not depend on the row's tally (one launch per pass, before the fused kernel):
// the weight of a unit of dosage w1 = round(beta 2^F) with its operand bytes, and the weight wfb of a missing
// genotype whenever the imputed dosage is known beforehand (ps / homref / fail, and the fall-back of int_ps /
// int_fail below --mincs).
struct MxPre {
    uint32_t c[3];
    uint32_t flags;  // 1: beta is not finite (every sample's sum becomes NaN); 2: wfb stands for NaN
    long long w1, wfb;
};
static_assert(sizeof(MxPre) == 32, "MxPre layout");
static_assert(kDW * kUD + 2 * kUC == 64, "units of a strip");

// n / d for 0 <= n <= d < 2^27 (0 / 0 = NaN), within an ulp: the weights it feeds are rounded to 2^-56 anyway
static __device__ __forceinline__ double fast_ratio(double n, double d) {
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    const double q = n * r;
    return __builtin_fma(__builtin_fma(-d, q, n), r, q);
}

// One row, from its complete tally word: the decisions of getImputedDosages (nimpress.nim:565-571), the locus
// constant (:417-447) or the sample imputation value (:450-481), as the three weight operands of the row.
// what mx_row needs of the row's precomputed part beyond MxPre itself: made BEFORE the row's tally word is back
struct MxPreX {
    double w1d;        // (double)w1
    long long w3, w4;  // 3 w1, 4 w1
};
static __device__ __forceinline__ void mx_row(const MxArgs &a, unsigned long long x, bool live, uint64_t row,
                                              const MxPre &pre, const MxPreX &px, bool write_stats, uint32_t (&wc)[3],
                                              uint32_t (&wme)[3], uint32_t (&wmo)[3], int &used, double &cst,
                                              bool need_odd = true) {
    wc[0] = wc[1] = wc[2] = wme[0] = wme[1] = wme[2] = wmo[0] = wmo[1] = wmo[2] = 0u;
    used = 0;
    cst = 0.0;
    if (!live) return;
    const uint32_t nmiss = (uint32_t)(x >> 28) & 0xFFFFFFFu, neff = (uint32_t)x & 0xFFFFFFFu;  // (both < 2^28)
    const uint32_t ngen = (uint32_t)a.n_samples - nmiss;
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    int reason;
    if ((int64_t)nmiss > a.t_maxmis) {  // == (double)nmiss / (double)N > --maxmis, t_maxmis found with that very division
        reason = NPS_REASON_MAXMIS;
        if (a.prm.imp_locus != NPS_LOCUS_IGNORE) {  // (rare: the row's score entry is fetched here)
            const double beta = a.desc[row].beta, eaf = a.desc[row].eaf;
            const bool rie = a.desc[row].ref_is_effect != 0;
            const double c = a.prm.imp_locus == NPS_LOCUS_PS       ? eaf * 2.0
                             : a.prm.imp_locus == NPS_LOCUS_HOMREF ? (rie ? 2.0 : 0.0)
                                                                   : nan;
            used = 1;
            cst = c * beta;
        }
    } else {
        reason = NPS_REASON_GENOTYPED;
        used = 1;
        // the common path without branches: the internal imputation value is worked out for every row and
        // selected (the control wave's step waits for exactly this chain of dependent operations)
        const bool internal = a.prm.imp_sample == NPS_SAMPLE_INT_PS || a.prm.imp_sample == NPS_SAMPLE_INT_FAIL;
        const double dgen = (double)ngen;
        const bool use_int = internal && dgen >= a.prm.min_cs;
        const double imp = fast_ratio((double)neff, dgen);
        const bool imp_nan = imp != imp;
        const long long wint = __double2ll_rn((imp_nan ? 0.0 : imp) * px.w1d);
        const bool bad = use_int ? imp_nan : (pre.flags & 2u) != 0;
        const long long wi = use_int ? (imp_nan ? px.w3 : wint) : pre.wfb;
        const bool dead = (pre.flags & 1u) != 0;  // a non-finite beta makes every sample's sum NaN (0 * NaN, NaN + x), as in the reference
        uint32_t e[3], o[3];
        mx_codes(wi - px.w3, bad ? 1u : 0u, e);  // a missing genotype has code 3 (4 in the odd operand)
        if (need_odd) mx_codes(wi - px.w4, bad ? 1u : 0u, o);  // (nps_mx.hip reads odd samples as code / 2 too: one operand)
        else o[0] = o[1] = o[2] = 0u;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            wc[i] = dead ? 0u : pre.c[i];
            wme[i] = dead ? 0u : e[i];
            wmo[i] = dead ? 0u : o[i];
        }
        if (dead) cst = nan;
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


}  // namespace nps
