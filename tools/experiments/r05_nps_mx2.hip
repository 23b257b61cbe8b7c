// nps_mx2.hip -- the strip kernel, second form (round 5): what does not depend on a row's tally no longer waits for it.
//
// nps_mx.hip parks the CODES of a superblock for two steps, until the row tallies of all strips are in, and only then
// accumulates  code * Wc + is_missing * Wm.  But Wc = beta is known beforehand; only Wm = (imputed dosage - 3) beta -- and
// the rare row over --maxmis -- depend on the tally (nimpress.nim:565-583).  Here
//
//   * code * Wc is accumulated in the step the superblock ARRIVES: registers -> a wave-private 1 KiB LDS image ->
//     ds_read_b64_tr_b4 -> two MFMAs per unit with the beta digits of the rows (tally-independent, from mx_prep_kernel);
//   * what is parked for the deferred part is the IS-MISSING MASK, one bit per genotype: the masks of rows 2l and 2l+1 of a
//     lane interleave into one 64-bit word, half a slot.  The same 128 KiB of LDS hold FOUR superblocks of masks instead
//     of two of codes, so the hand-over chain between the strips (publication -> first stage -> row word -> look ->
//     operands) has three steps instead of one and a half -- the third step DESIGN.md 4.2 asked for -- and the two
//     register banks that used to hold "one superblock being tallied, one waiting to be parked" are both loads in flight:
//     128 KiB per compute unit instead of 64;
//   * rows that turn out to be over --maxmis (the reference replaces EVERY sample's dosage by the locus constant,
//     nimpress.nim:565-571, 417-447) have had their code * beta added already: they go on a list, and the pass's epilogue
//     (mx_fold_kernel) re-reads just those rows and takes the products back -- exact integers, order independent.
//     1 row in 1000 on the bench cohort; a cohort where more than 2 % of the rows end up there is scored by the first
//     form from the next pass on (nps_engine.hip: mx_prefer_v1).
//
// Pipeline of a workgroup, step k (one barrier per step, L = 3):
//     phase A:  tally k, masks k -> ring slot k % 4, code x beta of k (eager), loads of k+2 -> the freed registers
//               control waves: look at the row words of k-L (issued first, consumed after their own units), operands of k-L
//     barrier
//     phase B:  is_missing x Wm of k-L (ring slot -> MFMA)        control waves: publish the tallies of k
// Everything else -- layout, operand encodings, the two-stage hand-over, bounded waits, teams -- is nps_mx.hip's.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "nps_kernels.h"
#include "nps_mx_common.h"

namespace nps {

// Seven data waves x nine units, ONE control wave with the strip's 64th unit and the per-row work of all 128 rows of a
// superblock (two rows per lane).  With two control waves of five units each (the first draft, nps_mx.hip's split) the
// control waves' own unit loads were in the way of every hand-timed wait: memory returns in order, the compiler counts
// only the operations it knows, and the wave that sets the step's pace stalled for its own refills twice per step
// (profiles/r05_mx2_timers.txt).  A control wave with one unit has one load in flight.
#ifndef NPS_MX2_DW   // (6 x 10 + two control waves of 2 units: 25.3 ms but six spilled VGPRs and wrong NaN flags -- not shipped)
#define NPS_MX2_DW 7
#define NPS_MX2_UD 9
#endif
constexpr int k2DW = NPS_MX2_DW;                          // data waves 0..k2DW-1
constexpr int k2UD = NPS_MX2_UD;                          // units of a data wave
constexpr int k2NC = 8 - k2DW;                            // control waves (waves k2DW..7)
constexpr int k2UC = (64 - k2DW * k2UD) / k2NC;           // units of a control wave
constexpr int k2Threads = 512;
constexpr int k2RPL = 2 / k2NC;                           // rows of a superblock per lane of a control wave
static_assert((k2NC == 1 || k2NC == 2) && k2DW * k2UD + k2NC * k2UC == 64 && k2UC >= 1, "units of a strip");
constexpr int k2R = 4;                                    // ring slots of parked is-missing masks
constexpr int k2L = k2R - 1;                              // steps between a superblock's arrival and its is-missing part
constexpr uint32_t k2Ring = 0;                            // [k2R][64 units][64 lanes x 8 bytes]
constexpr uint32_t k2Trans = k2R * 32768u;                // [waves][2 units][1 KiB]: the code image on its way to the transposed read
constexpr uint32_t k2Bc = k2Trans + (k2DW + 1) * 2048u;    // [2][128 rows][16 bytes]: beta digits
constexpr uint32_t k2Bm = k2Bc + 2 * 2048u;               // [2][even / odd operand][128 rows][16 bytes]: is-missing weights
constexpr uint32_t k2Tally = k2Bm + 2 * 4096u;            // [2][128] uint32: nmissing << 16 | neffect of the strip
constexpr uint32_t k2Pre = k2Tally + 1024u;               // [128 rows][16 bytes]: (w1, wfb) of the rows whose operands are made next (LDS-DMA)
constexpr uint32_t k2Look = k2Pre + 2048u;                // [128 rows] uint64: their tally words, as the look found them (LDS-DMA)
constexpr uint32_t k2LdsBytes = k2Look + 1024u;
static_assert(k2LdsBytes <= 160u * 1024u, "LDS of a compute unit");

// where row r's is-missing operands live in a Bm table: inside a block of 32 rows the even rows first, then the odd ones
// (a parked mask word holds rows 2l and 2l+1 bit-interleaved: the transposed read hands a lane 16 even rows, then 16 odd)
// (mx_perm: nps_mx_common.h)

#ifdef NPS_MX_TIMERS
__device__ unsigned long long g_mx2_timers[16][16];
#define MXT(i) do { if (timing) { const unsigned long long now_ = __builtin_readcyclecounter(); tph[i] += now_ - tlast; tlast = now_; } } while (0)
#else
#define MXT(i) do { } while (0)
#endif

// The control waves' own memory traffic never has a VGPR destination the compiler could mis-time: the beta digits of a
// superblock, the (w1, wfb) of the rows whose operands are due and the look at their tally words all arrive by LDS-DMA
// (lane l's 16 bytes -> LDS byte address dst + 16 l; counted in vmcnt, in order, by hand), and are read from LDS like any
// other LDS data.  The one exception is the returning add of the publication (mx2_add_ret): issued and waited for inside
// one straight-line region.  (nps_mx.hip's look was an inline-asm load into a register pair that had to stay untouched
// across a whole phase: ADVICE round 4.)
static __device__ __forceinline__ void mx2_dma16(const void *gsrc, uint32_t lds_dst_any, bool coherent) {
    unsigned keep;
    const uint32_t lds_dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst_any);  // (wave-uniform by construction)
    if (coherent)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gsrc), "s"(lds_dst)
                     : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(gsrc), "s"(lds_dst)
                     : "memory");
}
template <int N>
static __device__ __forceinline__ void mx2_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}

// DBG (diagnostics builds): 1 no tally popcounts, 2 no accumulation (either part), 4 no hand-over, 32 no eager part,
// 64 no deferred part
template <int NU, bool GUARD, bool CTL, int DBG>
static __device__ __forceinline__ void mx2_body(const MxArgs &a, char *const smem) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int g = lane >> 4, q = lane & 15;
    const uint32_t strip = blockIdx.x % a.P, team = blockIdx.x / a.P;
    const uint32_t n_t = team < a.n_sb ? (a.n_sb - team + a.Q - 1) / a.Q : 0u;  // superblocks of this team
    if (n_t == 0) return;  // (workgroup-uniform; the plan keeps Q <= n_sb)
    constexpr bool is_ctl = CTL && !(DBG & 128);  // (DBG 128: no control work at all -- the data path's ceiling)
    const int nu = strip == a.P - 1 ? (int)a.nu_last : 64;
    const int u0 = wave < k2DW ? wave * k2UD : k2DW * k2UD + (wave - k2DW) * k2UC;
    const int cw = wave - k2DW;  // control waves: 0 .. k2NC-1; rows lane + 64 (cw k2RPL + h), h < k2RPL
    const int n_my = GUARD ? max(0, min(NU, nu - u0)) : NU;  // wave-uniform
    constexpr bool full = !GUARD;

    char *const ring = smem + k2Ring + (size_t)u0 * 512;
    char *const trans = smem + k2Trans + (size_t)wave * 2048;
    const int woff = mx_rowoff(2 * lane);
    const int r1off = mx_rowoff(32 * g + q), r2off = mx_rowoff(32 * g + 16 + q);
    const int fr0 = (32 * g + q) * 16, fr1 = (32 * g + 16 + q) * 16;
    const int moff = 8 * lane;  // the mask word of rows 2 lane, 2 lane + 1 (and, read back transposed, of "row" 16 g + q = lane)
    const v4u *const base =
        a.units + ((uint64_t)strip * 64 * a.n_sb_cohort + (uint64_t)(a.sb0 + team) * nu + u0) * 64 + lane;
    const uint64_t sb_stride = (uint64_t)nu * 64 * a.Q;  // from one superblock of the team to its next

    v4f C[NU][2];
#pragma unroll
    for (int u = 0; u < NU; ++u) C[u][0] = C[u][1] = v4f{0.f, 0.f, 0.f, 0.f};
#ifndef NPS_MX2_BANKS
#define NPS_MX2_BANKS 1
#endif
    // register banks of the wave's units on their way in: ONE (superblock k + 1 is loaded while k is consumed, unit by
    // unit: 64 KiB in flight per compute unit, as in nps_mx.hip) -- two banks (k + 1 and k + 2 in flight) need 282 VGPRs
    // at nine units per wave (26 spilled), see DESIGN.md 4.2
    constexpr int kBanks = NPS_MX2_BANKS;
    v4u bank[kBanks][NU];
#pragma unroll
    for (int s = 0; s < kBanks; ++s)
#pragma unroll
        for (int u = 0; u < NU; ++u) bank[s][u] = v4u{0u, 0u, 0u, 0u};

    auto load_unit = [&](uint32_t k, int u) -> v4u {
        return __builtin_nontemporal_load(base + (uint64_t)k * sb_stride + u * 64);
    };
    auto load_sb = [&](uint32_t k, v4u(&dst)[NU]) {
        if (k >= n_t) return;
#pragma unroll
        for (int u = 0; u < NU; ++u)
            if (full || u < n_my) dst[u] = load_unit(k, u);
    };

    struct Tal {
        uint32_t xa = 0, ya = 0, za = 0, xb = 0, yb = 0, zb = 0;
    };
    // ---- the data path -----------------------------------------------------------------------------------------
    auto tally_add = [&](uint32_t kt, const Tal &t) {
        const unsigned long long pa = (t.xa + t.ya - 3u * t.za) | (t.za << 16), pb = (t.xb + t.yb - 3u * t.zb) | (t.zb << 16);
        unsigned long long *T = reinterpret_cast<unsigned long long *>(smem + k2Tally) + (kt & 1) * 64 + lane;
        atomicAdd(T, pa | (pb << 32));  // the 16-bit fields of a word never carry: a strip has 2048 samples
    };
    // phase A of step k: superblock k has arrived in `b`; its registers are refilled with superblock k + 2
    auto front = [&](uint32_t k, v4u(&b)[NU]) {
        if (k >= n_t || n_my == 0) return;
        char *const slot = ring + (size_t)(k % k2R) * 32768;
        const char *tab = smem + k2Bc + (k & 1) * 2048;
        v8i Bc;
        {
            const v3i c0 = tr6(tab + fr0), c1 = tr6(tab + fr1);
            Bc = v8i{c0[0], c0[1], c0[2], c1[0], c1[1], c1[2], 0, 0};
        }
        Tal t;
        auto one = [&](int u, bool refill) {
#ifdef NPS_MX2_SCHED
            __builtin_amdgcn_sched_barrier(NPS_MX2_SCHED);
#endif
            const v4u w = b[u];
            const uint32_t sx = w.x >> 1, sy = w.y >> 1, sz = w.z >> 1, sw = w.w >> 1;
            const uint32_t mx = w.x & sx & 0x55555555u, my = w.y & sy & 0x55555555u;
            const uint32_t mz = w.z & sz & 0x55555555u, mw = w.w & sw & 0x55555555u;
            if (!(DBG & 1)) {
                t.xa = bcnt_acc(w.y, bcnt_acc(w.x, t.xa));
                t.ya = bcnt_acc((w.x & 0xAAAAAAAAu) | (sy & 0x55555555u), t.ya);
                t.za = bcnt_acc(mx | (my << 1), t.za);
                t.xb = bcnt_acc(w.w, bcnt_acc(w.z, t.xb));
                t.yb = bcnt_acc((w.z & 0xAAAAAAAAu) | (sw & 0x55555555u), t.yb);
                t.zb = bcnt_acc(mz | (mw << 1), t.zb);
            }
            // the parked word: bit 0 / 2 of nibble j = sample 2j / 2j + 1 missing in row 2 lane, bit 1 / 3 = in row 2 lane + 1
            *reinterpret_cast<v2i *>(slot + u * 512 + moff) = v2i{(int)(mx | (mz << 1)), (int)(my | (mw << 1))};
            char *const tp = trans + (u & 1) * 1024;
            *reinterpret_cast<v4u *>(tp + woff) = w;
            if (refill) b[u] = load_unit(k + kBanks, u);
            if (DBG & (2 | 32)) return;
            const v2i t01 = tr4(tp + r1off), t23 = tr4(tp + r2off);
            const uint32_t x[4] = {(uint32_t)t01[0], (uint32_t)t01[1], (uint32_t)t23[0], (uint32_t)t23[1]};
            v8i ce = {0, 0, 0, 0, 0, 0, 0, 0}, co = ce;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                ce[r] = (int)(x[r] & 0x33333333u);         // even sample: 00hl = code / 2
                co[r] = (int)((x[r] >> 1) & 0x66666666u);  // odd sample: 0hl0 = 0, 1, 2, 4
            }
            C[u][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ce, Bc, C[u][0], 4, 2, 0, 128, 0, 130);
            C[u][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(co, Bc, C[u][1], 4, 2, 0, 127, 0, 130);
        };
        if (k + kBanks < n_t) {  // steady state: no per-unit conditions
#pragma unroll
            for (int u = 0; u < NU; ++u)
                if (full || u < n_my) one(u, true);
        } else {
#pragma unroll
            for (int u = 0; u < NU; ++u)
                if (full || u < n_my) one(u, false);
        }
        tally_add(k, t);
    };
    // phase B of step k: scores += is_missing * Wm for superblock j = k - L (masks parked L steps ago, weights just made)
    auto deferred = [&](uint32_t j) {
        if (j >= n_t || n_my == 0 || (DBG & (2 | 64))) return;
        const char *slot = ring + (size_t)(j % k2R) * 32768;
        const char *tab = smem + k2Bm + (j & 1) * 4096;
        v8i Bme, Bmo;
        {
            const v3i e0 = tr6(tab + fr0), e1 = tr6(tab + fr1);
            const v3i o0 = tr6(tab + 2048 + fr0), o1 = tr6(tab + 2048 + fr1);
            Bme = v8i{e0[0], e0[1], e0[2], e1[0], e1[1], e1[2], 0, 0};
            Bmo = v8i{o0[0], o0[1], o0[2], o1[0], o1[1], o1[2], 0, 0};
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (full || u < n_my) {
                const v2i x = tr4(slot + u * 512 + moff);  // nibble i: "row" 16 g + i = rows 32 g + 2 i (bits 0, 2), + 1 (bits 1, 3)
                const uint32_t lo = (uint32_t)x[0], hi = (uint32_t)x[1], lo1 = lo >> 1, hi1 = hi >> 1;
                const v8i me = {(int)(lo & 0x11111111u), (int)(hi & 0x11111111u), (int)(lo1 & 0x11111111u), (int)(hi1 & 0x11111111u), 0, 0, 0, 0};
                const v8i mo = {(int)(lo & 0x44444444u), (int)(hi & 0x44444444u), (int)(lo1 & 0x44444444u), (int)(hi1 & 0x44444444u), 0, 0, 0, 0};
                C[u][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(me, Bme, C[u][0], 4, 2, 0, 128, 0, 130);
                C[u][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(mo, Bmo, C[u][1], 4, 2, 0, 126, 0, 130);
            }
        }
    };

    auto store_c = [&](uint32_t f, bool zero) {
        float *dst = a.cpart + (((((uint64_t)f * a.Q + team) * a.P + strip) * 64 + u0) * 2) * 256 + lane * 4;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (full || u < n_my) {
                *reinterpret_cast<v4f *>(dst + (u * 2 + 0) * 256) = C[u][0];
                *reinterpret_cast<v4f *>(dst + (u * 2 + 1) * 256) = C[u][1];
                if (zero) C[u][0] = C[u][1] = v4f{0.f, 0.f, 0.f, 0.f};
            }
        }
    };

    // ---- control state (wave k2DW: all 128 rows of every superblock, rows lane and lane + 64) ------------------------
    uint32_t nloci_local = 0;
    double cst_local = 0.0;
    bool timed_out = false;
    // READY-MADE operands (a.ops != nullptr: --imp-sample ps / homref / fail, where the weight of a missing genotype does not
    // depend on the row's tally; mx_prep_kernel made both operands of every row): the Bm table of a superblock arrives by
    // LDS-DMA as it is, and all that is left to do per row once its tally word is complete is the --maxmis comparison, the
    // statistics and -- for the rare row over --maxmis -- zeroing its two table rows.  The control wave's 4 900 cycles of
    // operand arithmetic per step (profiles/r05_mx2_timers.txt) were what this form waited for.
    const bool ready = a.ops != nullptr;  // (uniform over the launch)
    bool rd_over[k2RPL], rd_used[k2RPL];  // of the superblock whose table is on its way: decided in ctl_tables, applied a step later
#pragma unroll
    for (int h = 0; h < k2RPL; ++h) rd_over[h] = rd_used[h] = false;
    bool rd_pending = false;
    uint32_t rd_j = 0;
    auto sb_row0 = [&](uint32_t k) -> uint64_t { return ((uint64_t)team + (uint64_t)k * a.Q) * 128; };
    const uint32_t lds_bc = (uint32_t)(uintptr_t)(smem + k2Bc);      // + (k & 1) * 2048
    const uint32_t lds_pre = (uint32_t)(uintptr_t)(smem + k2Pre);
    const uint32_t lds_look = (uint32_t)(uintptr_t)(smem + k2Look);
    // the beta digits of the 128 rows of superblock k -> table buffer k & 1 (the first 16 bytes of a row's MxPre: twelve
    // bytes of FP6 digits, then the flags word, which the transposed 96-bit reads never see): two DMAs of 64 rows
    auto ctl_dma_bc = [&](uint32_t k) {
#pragma unroll
        for (int h = 0; h < k2RPL; ++h)
            mx2_dma16(a.pre + sb_row0(k) + 64 * (cw * k2RPL + h) + lane, lds_bc + (k & 1u) * 2048u + (uint32_t)(cw * k2RPL + h) * 1024u, false);
    };
    // ... (w1, wfb) of the rows of superblock j, and the look at their tally words (one DMA: two words per lane)
    auto ctl_dma_look = [&](uint32_t j) {
        if (!ready) {
#pragma unroll
            for (int h = 0; h < k2RPL; ++h)
                mx2_dma16(reinterpret_cast<const char *>(a.pre + sb_row0(j) + 64 * (cw * k2RPL + h) + lane) + 16,
                          lds_pre + (uint32_t)(cw * k2RPL + h) * 1024u, false);
        }
        if (lane < 32 * k2RPL) mx2_dma16(a.tally + sb_row0(j) + 64 * k2RPL * cw + 2 * lane, lds_look + (uint32_t)(cw * k2RPL) * 512u, true);
    };
    // ready-made operands of superblock j -> table buffer j & 1 (4 KiB: this wave's share in pieces of 64 table rows).  Issued
    // behind the barrier after which no wave reads that buffer any more; landed at the vmcnt(0) on top of the next step.
    const uint32_t lds_bm = (uint32_t)(uintptr_t)(smem + k2Bm);
    auto ctl_dma_bm = [&](uint32_t j) {
#pragma unroll
        for (int q4 = 0; q4 < 4 / k2NC; ++q4) {
            const uint32_t piece = (uint32_t)(cw * (4 / k2NC) + q4);
            mx2_dma16(a.ops + ((uint64_t)team + (uint64_t)j * a.Q) * 256 + piece * 64 + lane, lds_bm + (j & 1u) * 4096u + piece * 1024u, false);
        }
    };
    // complete tallies of superblock j (x) -> the row's two is-missing operands in table buffer j & 1; a row over
    // --maxmis goes on the list of rows whose eager code x beta the epilogue takes back
    auto ctl_build = [&](uint32_t j, int crow, unsigned long long x, bool valid, bool ok, const v2ul pw) {
        const uint64_t row = sb_row0(j) + crow;
        // pw = (w1, wfb) as mx_prep_kernel left them for this kernel: w1 = INT64_MIN stands for a non-finite beta, wfb =
        // INT64_MIN for an imputed dosage that is NaN in the reference
        MxPre tpre;
        tpre.c[0] = tpre.c[1] = tpre.c[2] = 0u;
        const bool dead = (long long)pw[0] == kMxDeadW1, nanfb = (long long)pw[1] == kMxNanWfb;
        tpre.w1 = dead ? 0 : (long long)pw[0];
        tpre.wfb = nanfb ? 3 * tpre.w1 : (long long)pw[1];
        tpre.flags = (dead ? 1u : 0u) | (nanfb ? 2u : 0u);
        MxPreX px;
        px.w1d = (double)tpre.w1;
        px.w3 = 3 * tpre.w1;
        px.w4 = 4 * tpre.w1;
        uint32_t wc[3], wme[3], wmo[3];
        int used;
        double cst;
        const bool live = valid && ok;
        mx_row(a, x, live, row, tpre, px, strip == 0 && a.stats != nullptr, wc, wme, wmo, used, cst);  // (strip 0 of the row's team)
        char *p = smem + k2Bm + (j & 1) * 4096 + mx_perm(crow) * 16;
        *reinterpret_cast<v4u *>(p) = v4u{wme[0], wme[1], wme[2], 0u};
        *reinterpret_cast<v4u *>(p + 2048) = v4u{wmo[0], wmo[1], wmo[2], 0u};
        nloci_local += (uint32_t)__popcll(__ballot(used != 0));
        cst_local += cst;
        const bool over = live && (int64_t)((uint32_t)(x >> 28) & 0xFFFFFFFu) > a.t_maxmis;
        if (over && strip == 0) a.fix_rows[atomicAdd(a.fix_count, 1u)] = (uint32_t)row;
    };
#ifdef NPS_MX_TIMERS
    unsigned long long dbg_store[4] = {0, 0, 0, 0};
    unsigned long long *dbg_counts = (strip == a.P / 2 && team == 0) ? dbg_store : nullptr;
#endif
    // Has the look landed?  Nothing is counted: both staging areas are armed with values no row can have (a tally word of
    // all ones; w1 = INT64_MAX, the weights are below 2^56) before the DMAs go out, and the wave reads LDS until they are
    // gone.  (Counting vmcnt by hand would need loads, LDS-DMAs and returning atomics to complete in issue order among
    // each other; the first build that assumed it ran its fall-back path every step.)
    constexpr unsigned long long kLookArmed = ~0ull;
    constexpr long long kPreArmed = 0x7fffffffffffffffll;
    auto ctl_arm = [&]() {
#pragma unroll
        for (int h = 0; h < k2RPL; ++h) {
            const int crow = lane + 64 * (cw * k2RPL + h);
            *reinterpret_cast<unsigned long long *>(smem + k2Look + crow * 8) = kLookArmed;
            *reinterpret_cast<v2ul *>(smem + k2Pre + crow * 16) = v2ul{(unsigned long long)kPreArmed, 0ull};
        }
    };
    auto ctl_tables = [&](uint32_t j) {  // (j < n_t)
        unsigned long long x[k2RPL];
        v2ul pw[k2RPL];
        uint32_t spins = 0;
        for (;;) {
            asm volatile("" ::: "memory");
            bool landed = true;
#pragma unroll
            for (int h = 0; h < k2RPL; ++h) {
                const int crow = lane + 64 * (cw * k2RPL + h);
                x[h] = *reinterpret_cast<const unsigned long long *>(smem + k2Look + crow * 8);
                pw[h] = *reinterpret_cast<const v2ul *>(smem + k2Pre + crow * 16);
                landed = landed && x[h] != kLookArmed && (ready || (long long)pw[h][0] != kPreArmed);
            }
            if (__all(landed) || timed_out) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 1023u) == 0 && spins >= kMxSpinLimit) {
                if (lane == 0) __hip_atomic_store(a.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                timed_out = true;
            }
        }
#ifdef NPS_MX_TIMERS
        if (dbg_counts) { dbg_counts[0] += spins; dbg_counts[2] += 1; }
#endif
        ctl_arm();  // (for the next look: the DMAs that overwrite it are issued after this wave's next barrier)
        bool valid[k2RPL], ok[k2RPL];
        bool all_ok = true;
#pragma unroll
        for (int h = 0; h < k2RPL; ++h) {
            valid[h] = sb_row0(j) + lane + 64 * (cw * k2RPL + h) < a.n_rows;
            if (!valid[h]) x[h] = 0ull;
            ok[h] = !valid[h] || (uint32_t)(x[h] >> 56) == a.P || (DBG & 4);
            all_ok = all_ok && ok[h];
        }
        spins = 0;
        while (!__all(all_ok) && !timed_out) {  // (a strip that is more than two steps behind: the others wait here)
            __builtin_amdgcn_s_sleep(1);
            all_ok = true;
#pragma unroll
            for (int h = 0; h < k2RPL; ++h) {
                if (!ok[h]) {
                    x[h] = __hip_atomic_load(&a.tally[sb_row0(j) + lane + 64 * (cw * k2RPL + h)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok[h] = (uint32_t)(x[h] >> 56) == a.P;
                }
                all_ok = all_ok && ok[h];
            }
            if ((++spins & 255u) == 0) {
                const unsigned int t = __hip_atomic_load(a.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t != 0 || spins >= kMxSpinLimit) {
                    if (lane == 0) __hip_atomic_store(a.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    timed_out = true;
                }
            }
        }
#ifdef NPS_MX_TIMERS
        if (dbg_counts) dbg_counts[1] += spins;
#endif
        if (!ready) {
#pragma unroll
            for (int h = 0; h < k2RPL; ++h) ctl_build(j, lane + 64 * (cw * k2RPL + h), x[h], valid[h], ok[h], pw[h]);
            return;
        }
        // ready-made operands: the decisions of getImputedDosages (nimpress.nim:565-571) from the complete tally word, the
        // row's statistics, and what the table rows -- on their way by DMA -- will need once they have landed
        rd_j = j;
        rd_pending = true;
#pragma unroll
        for (int h = 0; h < k2RPL; ++h) {
            const int crow = lane + 64 * (cw * k2RPL + h);
            const uint64_t row = sb_row0(j) + crow;
            const bool live = valid[h] && ok[h];
            const uint32_t nmiss = (uint32_t)(x[h] >> 28) & 0xFFFFFFFu, neff = (uint32_t)x[h] & 0xFFFFFFFu;
            const bool over = live && (int64_t)nmiss > a.t_maxmis;
            int used = live ? 1 : 0, reason = NPS_REASON_GENOTYPED;
            double cst = 0.0;
            if (over) {  // (rare: the row's score entry is fetched here)
                reason = NPS_REASON_MAXMIS;
                used = 0;
                if (a.prm.imp_locus != NPS_LOCUS_IGNORE) {
                    const double beta = a.desc[row].beta, eaf = a.desc[row].eaf;
                    const bool rie = a.desc[row].ref_is_effect != 0;
                    const double c = a.prm.imp_locus == NPS_LOCUS_PS       ? eaf * 2.0
                                     : a.prm.imp_locus == NPS_LOCUS_HOMREF ? (rie ? 2.0 : 0.0)
                                                                           : __longlong_as_double(0x7ff8000000000000ll);
                    used = 1;
                    cst = c * beta;
                }
                if (strip == 0) a.fix_rows[atomicAdd(a.fix_count, 1u)] = (uint32_t)row;
            }
            rd_over[h] = over;
            rd_used[h] = live && !over;  // (a non-finite beta of such a row makes the sums NaN: told by the landed table's flags)
            if (live && strip == 0 && a.stats != nullptr) {
                nps_locus_stat st;
                st.ngenotyped = (uint32_t)a.n_samples - nmiss;
                st.nmissing = nmiss;
                st.neffect = (double)neff;
                st.used = used;
                st.reason = reason;
                a.stats[row] = st;
            }
            nloci_local += (uint32_t)__popcll(__ballot(used != 0));
            cst_local += cst;
        }
    };
    // the table of superblock rd_j has landed (vmcnt(0) on top of the step that consumes it, before its barrier): rows over
    // --maxmis lose their operands; a used row whose beta is not finite makes every sum NaN through the locus constants
    auto ctl_ready_apply = [&]() {
        if (!rd_pending) return;
        rd_pending = false;
#pragma unroll
        for (int h = 0; h < k2RPL; ++h) {
            const int crow = lane + 64 * (cw * k2RPL + h);
            char *p = smem + k2Bm + (rd_j & 1u) * 4096 + mx_perm(crow) * 16;
            if (rd_over[h]) {
                *reinterpret_cast<v4u *>(p) = v4u{0u, 0u, 0u, 0u};
                *reinterpret_cast<v4u *>(p + 2048) = v4u{0u, 0u, 0u, 0u};
            } else if (rd_used[h] && (*reinterpret_cast<const uint32_t *>(p + 12) & 1u)) {
                cst_local += __longlong_as_double(0x7ff8000000000000ll);
            }
        }
    };
    // the two-stage publication of nps_mx.hip: the strips arrive in groups on a word of their group; the strip whose add
    // completes a group (told by the value its add returned) adds the group's sum to the row's word
    const uint32_t grp = strip / a.grp_strips, grp_size = min(a.grp_strips, a.P - grp * a.grp_strips);

    // ---- prologue: superblock 0 on its way, the beta digits of superblocks 0 and 1 in place
    if (tid < 256) reinterpret_cast<uint32_t *>(smem + k2Tally)[tid] = 0u;
    load_sb(0, bank[0]);
    if (kBanks > 1) load_sb(1, bank[kBanks - 1]);
    if (is_ctl) {
        ctl_arm();
        ctl_dma_bc(0);
        if (n_t > 1) ctl_dma_bc(1);
        mx2_wait_vm<0>();
    }
    __syncthreads();

    // ---- steps
#ifdef NPS_MX_TIMERS
    const bool timing = strip == a.P / 2 && team == 0;
    unsigned long long tph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    auto step = [&](uint32_t k, v4u(&b)[NU]) {
        MXT(8);
        const uint32_t jd = k - (uint32_t)k2L;      // the superblock whose is-missing part is accumulated in this step
        const bool due = k >= (uint32_t)k2L && jd < n_t;
        const uint32_t jt = jd + 1;                  // ... and the one whose operands the control wave makes (for the next step)
        const bool due_t = k + 1 >= (uint32_t)k2L && jt < n_t;
        if (is_ctl) {
            // The compiler counts only the memory operations it knows: its wait for this wave's unit load (issued a step
            // ago) would also wait for every DMA issued since -- so everything is waited for HERE, before the look goes out
            // (the unit has had a whole step, the digits' DMA most of one) ...
            if (k < n_t && n_my != 0) {
#pragma unroll
                for (int u = 0; u < NU; ++u) asm volatile("" : "+v"(b[u]));
            }
            mx2_wait_vm<0>();
            if (ready) ctl_ready_apply();  // (the table this step's deferred part reads has landed: before this step's barrier)
            // ... and the look is issued BEFORE this step's load
            if (due_t) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the staging areas were re-armed with LDS writes)
                ctl_dma_look(jt);
            }
        }
        MXT(0);
        front(k, b);
        MXT(1);
        MXT(6);  // (the beta digits of k + 1, a DMA of the last step, have landed: the wait at the top of the step)
        __syncthreads();
        MXT(2);
        unsigned long long pub_old[k2RPL], pub_add[k2RPL];
        bool pub_live[k2RPL];
#pragma unroll
        for (int h = 0; h < k2RPL; ++h) {
            pub_old[h] = pub_add[h] = 0ull;
            pub_live[h] = false;
        }
        if (is_ctl) {
            // publication of the strip's tallies of superblock k (complete in LDS: barrier passed): returning adds,
            // asm so that THEIR wait is placed by hand (the compiler would wait where it first reads the result)
            uint32_t *T = reinterpret_cast<uint32_t *>(smem + k2Tally) + (k & 1) * 128;
#pragma unroll
            for (int h = 0; h < k2RPL; ++h) {
                const int crow = lane + 64 * (cw * k2RPL + h);
                const uint32_t v = T[crow];
                T[crow] = 0u;
                const uint64_t row = sb_row0(k) + crow;
                pub_live[h] = k < n_t && row < a.n_rows && !(DBG & 4);
                pub_add[h] = (1ull << 56) | ((unsigned long long)(v >> 16) << 28) | (v & 0xFFFFu);
                // ("+v" on the zeroed pub_old: the add writes the register pair the merge after this branch already
                //  lives in, so no copy of a value that has not arrived yet can be placed behind the instruction)
                if (pub_live[h])
                    asm volatile("global_atomic_add_x2 %0, %1, %2, off sc0"
                                 : "+v"(pub_old[h])
                                 : "v"(&a.tally1[(uint64_t)grp * a.n_sb * 128 + row]), "v"(pub_add[h])
                                 : "memory");
            }
            if (k + 2 < n_t) ctl_dma_bc(k + 2);  // (buffer k & 1: every wave has read the digits of k before this barrier)
            if (ready && due_t) ctl_dma_bm(jt);  // (buffer jt & 1: last read by the deferred part of the step before this one)
        }
        MXT(3);
        if (due) deferred(jd);
        MXT(4);
        if (is_ctl) {
            if (due_t) ctl_tables(jt);
            MXT(7);
            // the returning adds are back.  vmcnt(0) also waits for this step's unit load -- which the top of the next step,
            // a moment from here, waits for anyway.  ONE statement for both results: with a wait per branch the compiler
            // once placed a copy of the result IN FRONT of one of them (a read of registers whose data had not arrived).
            if (k2RPL == 2)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(pub_old[0]), "+v"(pub_old[k2RPL - 1]) : : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(pub_old[0]) : : "memory");
#pragma unroll
            for (int h = 0; h < k2RPL; ++h)
                if (pub_live[h]) {
                    const unsigned long long tot = pub_old[h] + pub_add[h];
                    if ((uint32_t)(tot >> 56) == grp_size)
                        __hip_atomic_fetch_add(&a.tally[sb_row0(k) + lane + 64 * (cw * k2RPL + h)], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
        }
        MXT(5);
        if (((k + 1) & (kFlushSb - 1)) == 0 && k + 1 < n_t) store_c(k / kFlushSb, true);
    };
    if (kBanks == 1) {
        const uint32_t n_steps = n_t + (uint32_t)k2L;
        for (uint32_t k = 0; k < n_steps; ++k) step(k, bank[0]);
    } else {
        const uint32_t n_steps = (n_t + (uint32_t)k2L + 1) / 2 * 2;
        for (uint32_t k = 0; k < n_steps; k += 2) {
            step(k + 0, bank[0]);
            step(k + 1, bank[kBanks - 1]);
        }
    }
    store_c((n_t - 1) / kFlushSb, false);
#ifdef NPS_MX_TIMERS
    if (timing && lane == 0) {
        for (int i = 0; i < 16; ++i) g_mx2_timers[wave][i] = tph[i];
        if (is_ctl) {
            g_mx2_timers[wave][10] = dbg_store[0];  // LDS polls until the look had landed, summed over the steps
            g_mx2_timers[wave][11] = dbg_store[1];  // polls of the row words themselves (a word was incomplete at the look)
            g_mx2_timers[wave][12] = dbg_store[2];
        }
    }
#endif
    if (is_ctl && strip == 0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cst_local += __shfl_down(cst_local, o, 64);
        if (lane == 0) {
            if (nloci_local) atomicAdd(a.nloci, (unsigned long long)nloci_local);
            if (cst_local != 0.0) a.const_sum[2 * team + cw] = cst_local;  // (NaN != 0 is true)
        }
    }
}

template <int DBG>
__global__ __launch_bounds__(k2Threads, k2Threads / 256) void fused_mx2_kernel(const MxArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6;
    const int nu = blockIdx.x % a.P == a.P - 1 ? (int)a.nu_last : 64;
#ifdef NPS_MX2_PROBE  // (register probes: one body per build)
#if NPS_MX2_PROBE == 1
    mx2_body<k2UD, false, false, DBG>(a, smem);
#elif NPS_MX2_PROBE == 2
    mx2_body<k2UC, false, true, DBG>(a, smem);
#elif NPS_MX2_PROBE == 3
    mx2_body<k2UD, true, false, DBG>(a, smem);
#else
    mx2_body<k2UC, true, true, DBG>(a, smem);
#endif
    return;
#endif
    // both bodies pass the same barriers; which one a wave runs is wave-uniform
    if (wave >= k2DW) {
        if (a.ctl_prio) __builtin_amdgcn_s_setprio(3);
        if (nu - (k2DW * k2UD + (wave - k2DW) * k2UC) >= k2UC)
            mx2_body<k2UC, false, true, DBG>(a, smem);
        else
            mx2_body<k2UC, true, true, DBG>(a, smem);  // (a ragged last strip without its 64th unit: the per-row work alone)
    } else {
        if (nu - wave * k2UD >= k2UD)
            mx2_body<k2UD, false, false, DBG>(a, smem);
        else
            mx2_body<k2UD, true, false, DBG>(a, smem);
    }
}

hipError_t launch_fused_mx2(hipStream_t st, const MxPlan &plan, const void *d_units, uint64_t n_sb_cohort, uint64_t sb0,
                            uint64_t n_samples, uint64_t n_rows, const nps_row_desc *d_desc, DevParams prm,
                            int64_t t_maxmis, int F, void *d_pre, unsigned long long *d_tally,
                            unsigned long long *d_tally1, nps_locus_stat *d_stats, unsigned long long *d_nloci,
                            double *d_const_sum, float *d_cpart, unsigned int *d_timeout, uint32_t *d_fix_rows,
                            unsigned int *d_fix_count, void *d_ops) {
    if (plan.given || n_rows > 0xffffffffull) return hipErrorInvalidValue;
    // the rows' is-missing operands ready made, wherever they do not depend on the rows' tallies
    const bool internal = prm.imp_sample == NPS_SAMPLE_INT_PS || prm.imp_sample == NPS_SAMPLE_INT_FAIL;
    if (internal) d_ops = nullptr;
    hipError_t pe = launch_mx_prep(st, d_desc, n_rows, prm, F, d_pre, 1, d_ops);
    if (pe != hipSuccess) return pe;
    const void *fn = (const void *)fused_mx2_kernel<0>;
#ifdef NPS_DIAGNOSTICS
    static const int dbg = getenv("NPS_MX_DEBUG") ? atoi(getenv("NPS_MX_DEBUG")) : 0;
    switch (dbg) {
    case 1: fn = (const void *)fused_mx2_kernel<1>; break;
    case 2: fn = (const void *)fused_mx2_kernel<2>; break;
    case 4: fn = (const void *)fused_mx2_kernel<4>; break;
    case 5: fn = (const void *)fused_mx2_kernel<5>; break;
    case 7: fn = (const void *)fused_mx2_kernel<7>; break;
    case 32: fn = (const void *)fused_mx2_kernel<32>; break;
    case 64: fn = (const void *)fused_mx2_kernel<64>; break;
    case 128: fn = (const void *)fused_mx2_kernel<128>; break;
    case 129: fn = (const void *)fused_mx2_kernel<129>; break;
    default: break;
    }
#endif
    static const void *attr_set = nullptr;
    if (attr_set != fn) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k2LdsBytes);
        if (e != hipSuccess) return e;
        attr_set = fn;
    }
    MxArgs a;
    a.units = (const v4u *)d_units;
    a.n_sb_cohort = n_sb_cohort;
    a.sb0 = (uint32_t)sb0;
    a.n_sb = plan.n_sb;
    a.n_rows = n_rows;
    a.n_samples = n_samples;
    a.P = plan.P;
    a.nu_last = plan.nu_last;
    a.Q = plan.Q;
    a.desc = d_desc;
    a.pre = (const MxPre *)d_pre;
    a.prm = prm;
    a.t_maxmis = t_maxmis;
    a.scale = std::ldexp(1.0, F);
    a.tally = d_tally;
    a.tally1 = d_tally1;
    a.stats = d_stats;
    a.nloci = d_nloci;
    a.const_sum = d_const_sum;
    a.cpart = d_cpart;
    a.timeout = d_timeout;
    a.ctl_prio = 1u;  // (the control wave's per-row work is the step's longest path: 29.9 -> 28.5 ms at 245 strips)
    a.grp_strips = 48u;
    a.fix_rows = d_fix_rows;
    a.fix_count = d_fix_count;
    a.part = nullptr;
    a.ops = (v4u *)d_ops;
    a.const_part = nullptr;
    a.done = nullptr;
    a.R = 0;
#ifdef NPS_DIAGNOSTICS
    if (getenv("NPS_MX_PRIO")) a.ctl_prio = (uint32_t)atoi(getenv("NPS_MX_PRIO"));
    if (getenv("NPS_MX_GRP")) a.grp_strips = (uint32_t)std::max(16, atoi(getenv("NPS_MX_GRP")));
#endif
    const dim3 grid(plan.P * plan.Q);
    void *args[] = {&a};
#ifdef NPS_MX_TIMERS
    {
        hipError_t e = hipLaunchCooperativeKernel(fn, grid, dim3(k2Threads), args, k2LdsBytes, st);
        if (e != hipSuccess) return e;
        (void)hipStreamSynchronize(st);
        unsigned long long h[16][16];
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_mx2_timers), sizeof(h));
        static const char *nm[16] = {"bc+look-issue", "front", "barrier", "publish-begin", "deferred", "publish-end", "look-wait",
                                     "tables", "loop/flush", "-", "-", "-", "-", "-", "-", "-"};
        for (int w : {0, 3, k2DW - 1, k2DW}) {
            fprintf(stderr, "mx2 timers wave %d (cycles per step):", w);
            for (int i = 0; i < 9; ++i) fprintf(stderr, "  %s %.0f", nm[i], (double)h[w][i] / (plan.n_sb / plan.Q));
            if (w >= k2DW)
                fprintf(stderr, "  | landing polls %.2f and row-word polls %.3f per step (%llu steps)", (double)h[w][10] / (double)h[w][12],
                        (double)h[w][11] / (double)h[w][12], h[w][12]);
            fprintf(stderr, "\n");
        }
        return hipSuccess;
    }
#endif
    return hipLaunchCooperativeKernel(fn, grid, dim3(k2Threads), args, k2LdsBytes, st);
}

}  // namespace nps
