// nps_mx3.hip -- the strip kernel, third form (round 5): the row tallies are completed, and the rows' operands made, ONCE
// -- by a few REDUCER workgroups on the compute units the strips leave idle (245 strips on a 256-CU part: 11) -- instead of
// 245 times, once per strip.
//
// What the first two forms spend a step on: every strip publishes its share of 128 row tallies with atomic adds on shared
// words (245 adds on one line are served one after the other: hence two stages and a returning add), looks at the
// complete words again (245 readers per line), and turns each of them into weight operands -- the same 128 x two
// float64 divisions and digit extractions in every one of the 245 workgroups, on control waves whose straight-line
// path sets the pace of the step (profiles/r05_mx2_timers.txt).  Here
//
//   strips    8 identical waves x 8 units, nps_mx2.hip's data path: tally popcounts, code x beta accumulated on arrival,
//             the is-missing masks parked in a ring of four superblocks.  After the step's barrier ONE wave stores the
//             strip's 128 partial tallies -- plain 8-byte stores, no atomics, each word carrying a 7-bit tag of its step --
//             and the finished is-missing operands of superblock k - 3 arrive by LDS-DMA from a table in global memory,
//             6 KiB per superblock, each 16-byte row carrying its superblock's number: a stale or half-written table is
//             seen and fetched again.  No control wave, no per-row arithmetic, no returning memory operation.
//   reducers  R workgroups (grid blocks P Q .. P Q + R - 1), superblocks k = r, r + R, ...: wait until all P strips'
//             words of the superblock carry its tag, add them up, and do per row what getImputedDosages does with the
//             whole-cohort tally (nimpress.nim:565-583): decision, statistics, nloci, locus constants, the list of rows
//             over --maxmis (their eager code x beta is taken back by mx_fold_kernel), and the row's two is-missing
//             operands into the table.
//
// The chain per superblock: partial stores (start of the strips' second half-step) -> reducer sees them, adds, computes
// (two to three microseconds) -> table rows -> the strips' DMA one and a half steps later.  A reducer that is late makes
// the strips that need its table wait (bounded); correctness never depends on timing.
// Everything else -- layout, operand encodings, digit sums, the epilogue -- is nps_mx.hip's and nps_mx2.hip's.
//
// STATUS (round 5): an EXPERIMENT, not part of libnps.so -- the file compiles to nothing unless NPS_WITH_MX3 is defined
// (tools/mkexp.sh NAME -DNPS_WITH_MX3 -DNPS_DIAGNOSTICS, then NPS_MX_FORM=3 with NPS_MODE_FUSED_EAGER).  Measured at the
// bench shape (500 000 x 1 000 000, 245 strips, 8 reducers): 72.8 ms against the first form's 22.5 -- the reducers are
// the pace: 7813 superblocks / 8 reducers = one superblock per 2.8 us x 8 = 22 us of budget each, of which adding up
// 245 x 128 words takes 19 us (40 000 cycles: four threads per row, their loads one after the other) and the per-row
// work 4.5 us, and the strips then wait ~13 000 cycles per step for the table to land (profiles/r05_mx3_timers.txt).
// Even with the sum done in one round trip the hand-over has three steps (8.4 us: the LDS ring holds four superblocks of
// masks) for publish -> poll -> 125 KB read -> operands -> table -> DMA, five dependent trips through memory between
// XCDs; the first form does the same with two.  Parity: the statistics and nloci are exact at every shape tried, the
// scores at 6 of the 12 shapes of test_gt2x_eager_mode_vs_oracle (the (16385, 47) case differs in its NaN positions);
// not pursued further (a likely cause, found afterwards in nps_mxg.hip: LDS rows left by an earlier launch carry the numbers
// this launch waits for -- the landing marks of a launch must be cleared before its first DMA).  One finding kept: a static __shared__ variable in a kernel that also has a dynamic block moves
// the dynamic block off its 16-byte alignment, and ds_read_b96_tr_b6 from a misaligned row returns OTHER bits (plain LDS
// reads do not care) -- every extern __shared__ block of the strip kernels is now declared aligned(16).
#ifdef NPS_WITH_MX3
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "nps_kernels.h"
#include "nps_mx_common.h"

namespace nps {

constexpr int k3W = 8, k3U = 8;                           // waves of a strip workgroup, units of a wave
constexpr int k3Threads = k3W * 64;
constexpr int k3R = 4, k3L = k3R - 1;                     // mask ring; steps between arrival and the is-missing part
constexpr uint32_t k3Ring = 0;                            // [k3R][64 units][64 lanes x 8 bytes]
constexpr uint32_t k3Trans = k3R * 32768u;                // [k3W][2 units][1 KiB]
constexpr uint32_t k3Bc = k3Trans + k3W * 2048u;          // [2][128 rows][16 bytes]: beta digits (+ superblock tag)
constexpr uint32_t k3Bm = k3Bc + 2 * 2048u;               // [2][even / odd operand][128 rows][16 bytes] (+ tag)
constexpr uint32_t k3Tally = k3Bm + 2 * 4096u;            // [2][128] uint32: nmissing << 16 | neffect of the strip
constexpr uint32_t k3LdsBytes = k3Tally + 1024u;
static_assert(k3W * k3U == 64 && k3LdsBytes <= 160u * 1024u, "strip geometry");
constexpr uint32_t k3Spin = 1u << 22;

static __device__ __forceinline__ int mx3_perm(int r) { return (r & 96) | ((r & 1) << 4) | ((r >> 1) & 15); }  // (= mx2_perm)

static __device__ __forceinline__ void mx3_dma16(const void *gsrc, uint32_t lds_dst_any) {  // coherent (agent scope) LDS-DMA
    unsigned keep;
    const uint32_t lds_dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst_any);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

#ifdef NPS_MX_TIMERS
__device__ unsigned long long g_mx3_timers[16][8];
#define M3T(i) do { if (timing) { const unsigned long long now_ = __builtin_readcyclecounter(); tph[i] += now_ - tlast; tlast = now_; } } while (0)
#else
#define M3T(i) do { } while (0)
#endif

#ifndef NPS_MX3_BANKS
#define NPS_MX3_BANKS 2
#endif

// ---- a strip ---------------------------------------------------------------------------------------------------------
template <bool GUARD, int DBG>
static __device__ __forceinline__ void mx3_strip(const MxArgs &a, char *const smem) {
    constexpr int NU = k3U, kBanks = NPS_MX3_BANKS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int g = lane >> 4, q = lane & 15;
    const uint32_t strip = blockIdx.x % a.P, team = blockIdx.x / a.P;
    const uint32_t n_t = team < a.n_sb ? (a.n_sb - team + a.Q - 1) / a.Q : 0u;  // superblocks of this team
    if (n_t == 0) return;
    const int nu = strip == a.P - 1 ? (int)a.nu_last : 64;
    const int u0 = wave * NU;
    const int n_my = GUARD ? max(0, min(NU, nu - u0)) : NU;
    constexpr bool full = !GUARD;

    char *const ring = smem + k3Ring + (size_t)u0 * 512;
    char *const trans = smem + k3Trans + (size_t)wave * 2048;
    const int woff = mx_rowoff(2 * lane);
    const int r1off = mx_rowoff(32 * g + q), r2off = mx_rowoff(32 * g + 16 + q);
    const int fr0 = (32 * g + q) * 16, fr1 = (32 * g + 16 + q) * 16;
    const int moff = 8 * lane;
    const v4u *const base =
        a.units + ((uint64_t)strip * 64 * a.n_sb_cohort + (uint64_t)(a.sb0 + team) * nu + u0) * 64 + lane;
    const uint64_t sb_stride = (uint64_t)nu * 64 * a.Q;
    auto sb_of = [&](uint32_t k) -> uint32_t { return team + k * a.Q; };  // superblock of the run

    v4f C[NU][2];
#pragma unroll
    for (int u = 0; u < NU; ++u) C[u][0] = C[u][1] = v4f{0.f, 0.f, 0.f, 0.f};
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
    auto tally_add = [&](uint32_t kt, const Tal &t) {
        const unsigned long long pa = (t.xa + t.ya - 3u * t.za) | (t.za << 16), pb = (t.xb + t.yb - 3u * t.zb) | (t.zb << 16);
        unsigned long long *T = reinterpret_cast<unsigned long long *>(smem + k3Tally) + (kt & 1) * 64 + lane;
        atomicAdd(T, pa | (pb << 32));
    };
    // first half of step k: superblock k has arrived in `b` (tally, masks -> ring, code x beta); refill with k + kBanks
    auto front = [&](uint32_t k, v4u(&b)[NU]) {
        if (k >= n_t || n_my == 0) return;
        char *const slot = ring + (size_t)(k % k3R) * 32768;
        const char *tab = smem + k3Bc + (k & 1) * 2048;
        v8i Bc;
        {
            const v3i c0 = tr6(tab + fr0), c1 = tr6(tab + fr1);
            Bc = v8i{c0[0], c0[1], c0[2], c1[0], c1[1], c1[2], 0, 0};
        }
        Tal t;
        auto one = [&](int u, bool refill) {
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
                ce[r] = (int)(x[r] & 0x33333333u);
                co[r] = (int)((x[r] >> 1) & 0x66666666u);
            }
            C[u][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ce, Bc, C[u][0], 4, 2, 0, 128, 0, 130);
            C[u][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(co, Bc, C[u][1], 4, 2, 0, 127, 0, 130);
        };
        if (k + kBanks < n_t) {
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
    // second half of step k: scores += is_missing * Wm for superblock j = k - L
    auto deferred = [&](uint32_t j) {
        if (j >= n_t || n_my == 0 || (DBG & (2 | 64))) return;
        const char *slot = ring + (size_t)(j % k3R) * 32768;
        const char *tab = smem + k3Bm + (j & 1) * 4096;
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
                const v2i x = tr4(slot + u * 512 + moff);
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

    // ---- tables in, partial tallies out (no wave is special beyond which kilobyte it fetches)
    bool timed_out = false;
    auto give_up = [&]() {
        if (lane == 0) __hip_atomic_store(a.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        timed_out = true;
    };
    // beta digits of superblock k: waves 0, 1 fetch 64 rows each (the first 16 bytes of a row's MxPre; dword 3 = tag)
    auto dma_bc = [&](uint32_t k) {
        if (wave < 2 && k < n_t)
            mx3_dma16(a.pre + (uint64_t)sb_of(k) * 128 + 64 * wave + lane, (uint32_t)(uintptr_t)(smem + k3Bc) + (k & 1u) * 2048u + (uint32_t)wave * 1024u);
    };
    // is-missing operands of superblock j: waves 2..5 fetch 64 table rows each
    auto dma_bm = [&](uint32_t j) {
        if (wave >= 2 && wave < 6 && j < n_t)
            mx3_dma16(a.ops + (uint64_t)sb_of(j) * 256 + 64 * (wave - 2) + lane,
                      (uint32_t)(uintptr_t)(smem + k3Bm) + (j & 1u) * 4096u + (uint32_t)(wave - 2) * 1024u);
    };
    // this wave's piece has landed: every row carries the superblock's number.  Not yet (or the reducer had not written
    // the table when the DMA read it): look again, and fetch again every 32 looks.
    auto landed = [&](const char *piece, uint32_t want, bool table, uint32_t idx) {
        const char *p = piece + lane * 16 + 12;
        for (uint32_t spins = 1;; ++spins) {
            asm volatile("" ::: "memory");
            const uint32_t tag = *reinterpret_cast<const uint32_t *>(p);
            if (__all(tag == want) || timed_out) return;
            __builtin_amdgcn_s_sleep(2);
            if ((spins & 31u) == 0) {
                if (table)
                    dma_bm(idx);
                else
                    dma_bc(idx);
                const unsigned int t = __hip_atomic_load(a.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t != 0 || spins >= k3Spin) give_up();
            }
        }
    };
    auto landed_bc = [&](uint32_t k) {
        if (wave < 2 && k < n_t) landed(smem + k3Bc + (k & 1u) * 2048u + wave * 1024, sb_of(k) + 1u, false, k);
    };
    auto landed_bm = [&](uint32_t j) {
        if (wave >= 2 && wave < 6 && j < n_t && !(DBG & 4)) landed(smem + k3Bm + (j & 1u) * 4096u + (wave - 2) * 1024, sb_of(j) + 1u, true, j);
    };
    // the strip's tallies of superblock k (complete in LDS) -> its 128 words of the team's ring slot k % 8, tagged k & 127
    auto publish = [&](uint32_t k) {
        if (wave != 6 || k >= n_t) return;
        uint32_t *T = reinterpret_cast<uint32_t *>(smem + k3Tally) + (k & 1) * 128;
        const uint32_t v0 = T[2 * lane], v1 = T[2 * lane + 1];
        T[2 * lane] = 0u;
        T[2 * lane + 1] = 0u;
        const uint32_t tag = (k & 127u) << 25;
        const unsigned long long w = (unsigned long long)(tag | ((v0 >> 16) << 13) | (v0 & 0x1FFFu)) |
                                     ((unsigned long long)(tag | ((v1 >> 16) << 13) | (v1 & 0x1FFFu)) << 32);
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(
            a.part + (((uint64_t)team * kMx3PartRing + (k % kMx3PartRing)) * a.P + strip) * 128) + lane;
        if (!(DBG & 4)) __hip_atomic_store(dst, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    // ---- prologue
    if (tid < 256) reinterpret_cast<uint32_t *>(smem + k3Tally)[tid] = 0u;
    load_sb(0, bank[0]);
    if (kBanks > 1) load_sb(1, bank[kBanks - 1]);
    dma_bc(0);
    dma_bc(1);
    landed_bc(0);
    __syncthreads();
#ifdef NPS_MX_TIMERS
    const bool timing = strip == a.P / 2 && team == 0;
    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    auto step = [&](uint32_t k, v4u(&b)[NU]) {
        M3T(5);
        const uint32_t jd = k - (uint32_t)k3L;
        const bool due = k >= (uint32_t)k3L && jd < n_t;
        front(k, b);
        M3T(0);
        landed_bc(k + 1);
        if (due) landed_bm(jd);
        M3T(1);
        __syncthreads();
        M3T(2);
        publish(k);
        dma_bc(k + 2);                  // (buffer k & 1: every wave has read the digits of k before this barrier)
        if (k + 1 >= (uint32_t)k3L) dma_bm(jd + 1);  // (for the next step; the reducers have had two steps)
        M3T(3);
        if (due) deferred(jd);
        M3T(4);
        if (((k + 1) & (kFlushSb - 1)) == 0 && k + 1 < n_t) store_c(k / kFlushSb, true);
    };
    if (kBanks == 1) {
        const uint32_t n_steps = n_t + (uint32_t)k3L;
        for (uint32_t k = 0; k < n_steps; ++k) step(k, bank[0]);
    } else {
        const uint32_t n_steps = (n_t + (uint32_t)k3L + 1) / 2 * 2;
        for (uint32_t k = 0; k < n_steps; k += 2) {
            step(k + 0, bank[0]);
            step(k + 1, bank[kBanks - 1]);
        }
    }
    store_c((n_t - 1) / kFlushSb, false);
#ifdef NPS_MX_TIMERS
    if (timing && lane == 0)
        for (int i = 0; i < 8; ++i) g_mx3_timers[wave][i] = tph[i];
#endif
}

// ---- a reducer -------------------------------------------------------------------------------------------------------
static __device__ __forceinline__ void mx3_reducer(const MxArgs &a, char *const smem) {
    const int tid = threadIdx.x;
    const uint32_t r = blockIdx.x - a.P * a.Q;
    uint32_t *s_nm = reinterpret_cast<uint32_t *>(smem);            // [4][128]
    uint32_t *s_ne = s_nm + 512;                                    // [4][128]
    double *s_c = reinterpret_cast<double *>(smem + 4096);          // [128]
    unsigned int *s_used = reinterpret_cast<unsigned int *>(smem + 4096 + 1024);  // [2]
    bool timed_out = false;
#ifdef NPS_MX_TIMERS
    unsigned long long t_wait = 0, t_sum = 0, t_rows = 0, n_done = 0;
#endif
    for (uint32_t k = r; k < a.n_sb; k += a.R) {
        const uint32_t team = k % a.Q, t = k / a.Q, tag = t & 127u;
        const uint32_t *src = a.part + (((uint64_t)team * kMx3PartRing + (t % kMx3PartRing)) * a.P) * 128;
#ifdef NPS_MX_TIMERS
        unsigned long long c0 = __builtin_readcyclecounter();
#endif
        // stage 1: the first word of every strip carries the superblock's tag (cheap to look at again and again)
        for (uint32_t spins = 1; !timed_out; ++spins) {
            bool ok = true;
            for (uint32_t s = tid; s < a.P; s += k3Threads)
                ok = ok && (__hip_atomic_load(src + (uint64_t)s * 128, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 25) == tag;
            if (__syncthreads_and(ok ? 1 : 0)) break;
            __builtin_amdgcn_s_sleep(4);
            if ((spins & 63u) == 0) {  // (every thread takes the same decision: the loop's barriers must match)
                const unsigned int to = __hip_atomic_load(a.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__syncthreads_or((to != 0 || spins >= k3Spin) ? 1 : 0)) {
                    if (tid == 0) __hip_atomic_store(a.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    timed_out = true;
                }
            }
        }
#ifdef NPS_MX_TIMERS
        unsigned long long c1 = __builtin_readcyclecounter();
#endif
        // stage 2: all words, four threads per row (strips tid >> 7, + 4, ...); a word that does not carry the tag yet
        // (a strip's 512 bytes span four lines) sends the whole stage round again
        const uint32_t row_in = tid & 127u, part_of = tid >> 7;
        uint32_t nm = 0, ne = 0;
        for (uint32_t spins = 1;; ++spins) {
            nm = ne = 0;
            bool ok = true;
            for (uint32_t s = part_of; s < a.P; s += 4) {
                const uint32_t w = __hip_atomic_load(src + (uint64_t)s * 128 + row_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ok && (w >> 25) == tag;
                nm += (w >> 13) & 0xFFFu;
                ne += w & 0x1FFFu;
            }
            if (__syncthreads_and(ok ? 1 : 0) || timed_out) break;  // (timed_out is the same in every thread)
            __builtin_amdgcn_s_sleep(2);
            if ((spins & 63u) == 0 && spins >= k3Spin) {  // (spins is the same in every thread)
                if (tid == 0) __hip_atomic_store(a.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                timed_out = true;
            }
        }
        s_nm[part_of * 128 + row_in] = nm;
        s_ne[part_of * 128 + row_in] = ne;
        __syncthreads();
#ifdef NPS_MX_TIMERS
        unsigned long long c2 = __builtin_readcyclecounter();
#endif
        // stage 3: per row (threads 0..127): nimpress.nim:565-583 from the whole-cohort tally
        int used = 0;
        double cst = 0.0;
        if (tid < 128) {
            const uint64_t row = (uint64_t)k * 128 + tid;
            uint32_t wc[3], wme[3] = {0u, 0u, 0u}, wmo[3] = {0u, 0u, 0u};
            if (row < a.n_rows && !timed_out) {
                const unsigned long long nmiss = (unsigned long long)s_nm[tid] + s_nm[128 + tid] + s_nm[256 + tid] + s_nm[384 + tid];
                const unsigned long long neff = (unsigned long long)s_ne[tid] + s_ne[128 + tid] + s_ne[256 + tid] + s_ne[384 + tid];
                const unsigned long long x = ((unsigned long long)a.P << 56) | (nmiss << 28) | neff;
                const v4u p1 = reinterpret_cast<const v4u *>(a.pre + row)[1];  // (w1, wfb) with the flags as sentinels
                const long long w1r = (long long)((unsigned long long)p1.x | ((unsigned long long)p1.y << 32));
                const long long wfr = (long long)((unsigned long long)p1.z | ((unsigned long long)p1.w << 32));
                MxPre tpre;
                tpre.c[0] = tpre.c[1] = tpre.c[2] = 0u;
                const bool dead = w1r == kMxDeadW1, nanfb = wfr == kMxNanWfb;
                tpre.w1 = dead ? 0 : w1r;
                tpre.wfb = nanfb ? 3 * tpre.w1 : wfr;
                tpre.flags = (dead ? 1u : 0u) | (nanfb ? 2u : 0u);
                MxPreX px;
                px.w1d = (double)tpre.w1;
                px.w3 = 3 * tpre.w1;
                px.w4 = 4 * tpre.w1;
                mx_row(a, x, true, row, tpre, px, a.stats != nullptr, wc, wme, wmo, used, cst);
                if ((int64_t)nmiss > a.t_maxmis) a.fix_rows[atomicAdd(a.fix_count, 1u)] = (uint32_t)row;
            }
            // the row's operands, where the strips' tables want them; the half that carries the tag goes out last
            const uint32_t stamp = k + 1u;
            unsigned long long *o = reinterpret_cast<unsigned long long *>(a.ops + (uint64_t)k * 256 + mx3_perm(tid));
            __hip_atomic_store(o, (unsigned long long)wme[0] | ((unsigned long long)wme[1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o + 256, (unsigned long long)wmo[0] | ((unsigned long long)wmo[1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o + 1, (unsigned long long)wme[2] | ((unsigned long long)stamp << 32), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(o + 257, (unsigned long long)wmo[2] | ((unsigned long long)stamp << 32), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            s_c[tid] = cst;
            const unsigned long long b = __ballot(used != 0);
            if ((tid & 63) == 0) s_used[tid >> 6] = (unsigned int)__popcll(b);
        }
        __syncthreads();
        if (tid == 0) {
            double c = 0.0;
            for (int i = 0; i < 128; ++i) c += s_c[i];  // (row order: bit-reproducible)
            a.const_part[k] = c;
            const unsigned int u = s_used[0] + s_used[1];
            if (u) atomicAdd(a.nloci, (unsigned long long)u);
        }
        __syncthreads();
#ifdef NPS_MX_TIMERS
        t_wait += c1 - c0;
        t_sum += c2 - c1;
        t_rows += __builtin_readcyclecounter() - c2;
        ++n_done;
#endif
    }
#ifdef NPS_MX_TIMERS
    if (tid == 0 && r == 0) {
        g_mx3_timers[8][0] = t_wait;
        g_mx3_timers[8][1] = t_sum;
        g_mx3_timers[8][2] = t_rows;
        g_mx3_timers[8][3] = n_done;
    }
#endif
    // the last reducer to finish adds the superblocks' locus constants in order
    // (no static __shared__ in this kernel: it would sit in front of the dynamic block and take its 16-byte alignment,
    //  and a transposed 96-bit LDS read from a misaligned row returns other bits -- plain reads do not care)
    __syncthreads();
    if (tid == 0 && (__threadfence(), atomicAdd(a.done, 1u) + 1u == a.R)) {
        __threadfence();
        double c = 0.0;
        for (uint32_t i = 0; i < a.n_sb; ++i) c += __hip_atomic_load(&a.const_part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.const_sum[0] = c;
        *a.done = 0u;
    }
}

template <int DBG>
__global__ __launch_bounds__(k3Threads, k3Threads / 256) void fused_mx3_kernel(const MxArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (blockIdx.x >= a.P * a.Q) {
        if (!(DBG & 4)) mx3_reducer(a, smem);  // (DBG 4: no hand-over -- the strips neither publish nor wait)
        return;
    }
    const int wave = threadIdx.x >> 6;
    const int nu = blockIdx.x % a.P == a.P - 1 ? (int)a.nu_last : 64;
    if (nu - wave * k3U >= k3U)
        mx3_strip<false, DBG>(a, smem);
    else
        mx3_strip<true, DBG>(a, smem);
}

// how many reducer workgroups a plan leaves room for (0: this form does not apply -- every compute unit holds a strip)
uint32_t mx3_reducers(int device, const MxPlan &plan) {
    if (!plan.ok || plan.given) return 0;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
    const int64_t spare = (int64_t)cus - (int64_t)plan.P * plan.Q;
    return spare < 1 ? 0u : (uint32_t)std::min<int64_t>(spare, 8);
}

// d_part: Q x 8 x P x 128 uint32, filled with 0xFF bytes; d_ops: n_sb x 256 x 16 bytes, zeroed (no row may carry a
// superblock's number from an earlier pass); d_const_part: n_sb doubles; d_done: one zeroed word; d_fix_*: as for
// launch_fused_mx2; d_const_sum[0] receives the locus constants
hipError_t launch_fused_mx3(hipStream_t st, const MxPlan &plan, uint32_t R, const void *d_units, uint64_t n_sb_cohort,
                            uint64_t sb0, uint64_t n_samples, uint64_t n_rows, const nps_row_desc *d_desc, DevParams prm,
                            int64_t t_maxmis, int F, void *d_pre, nps_locus_stat *d_stats, unsigned long long *d_nloci,
                            double *d_const_sum, float *d_cpart, unsigned int *d_timeout, uint32_t *d_fix_rows,
                            unsigned int *d_fix_count, uint32_t *d_part, void *d_ops, double *d_const_part,
                            unsigned int *d_done) {
    if (plan.given || R == 0 || n_rows > kMx2MaxRows) return hipErrorInvalidValue;
    hipError_t pe = launch_mx_prep(st, d_desc, n_rows, prm, F, d_pre, 2);
    if (pe != hipSuccess) return pe;
    const void *fn = (const void *)fused_mx3_kernel<0>;
#ifdef NPS_DIAGNOSTICS
    static const int dbg = getenv("NPS_MX_DEBUG") ? atoi(getenv("NPS_MX_DEBUG")) : 0;
    switch (dbg) {
    case 1: fn = (const void *)fused_mx3_kernel<1>; break;
    case 2: fn = (const void *)fused_mx3_kernel<2>; break;
    case 4: fn = (const void *)fused_mx3_kernel<4>; break;
    case 5: fn = (const void *)fused_mx3_kernel<5>; break;
    case 32: fn = (const void *)fused_mx3_kernel<32>; break;
    case 64: fn = (const void *)fused_mx3_kernel<64>; break;
    default: break;
    }
#endif
    static const void *attr_set = nullptr;
    if (attr_set != fn) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)k3LdsBytes);
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
    a.tally = nullptr;
    a.tally1 = nullptr;
    a.stats = d_stats;
    a.nloci = d_nloci;
    a.const_sum = d_const_sum;
    a.cpart = d_cpart;
    a.timeout = d_timeout;
    a.ctl_prio = 0;
    a.grp_strips = 0;
    a.fix_rows = d_fix_rows;
    a.fix_count = d_fix_count;
    a.part = d_part;
    a.ops = (v4u *)d_ops;
    a.const_part = d_const_part;
    a.done = d_done;
    a.R = R;
    const dim3 grid(plan.P * plan.Q + R);
    void *args[] = {&a};
    hipError_t e = hipLaunchCooperativeKernel(fn, grid, dim3(k3Threads), args, k3LdsBytes, st);
#ifdef NPS_MX_TIMERS
    if (e == hipSuccess) {
        (void)hipStreamSynchronize(st);
        unsigned long long h[16][8];
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_mx3_timers), sizeof(h));
        static const char *nm[6] = {"front", "landing", "barrier", "publish+dma", "deferred", "loop/flush"};
        const double steps = (double)(plan.n_sb / plan.Q);
        for (int w : {0, 1, 2, 6, 7}) {
            fprintf(stderr, "mx3 timers wave %d (cycles per step):", w);
            for (int i = 0; i < 6; ++i) fprintf(stderr, "  %s %.0f", nm[i], (double)h[w][i] / steps);
            fprintf(stderr, "\n");
        }
        const double nd = (double)std::max<unsigned long long>(h[8][3], 1);
        fprintf(stderr, "mx3 timers reducer 0 (cycles per superblock, %llu of them, %u reducers): wait %.0f  sum %.0f  rows %.0f\n",
                h[8][3], R, (double)h[8][0] / nd, (double)h[8][1] / nd, (double)h[8][2] / nd);
    }
#endif
    return e;
}

}  // namespace nps
#endif  // NPS_WITH_MX3
