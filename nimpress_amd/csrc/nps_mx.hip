// nps_mx.hip -- the single-read kernel for a resident 2-bit cohort in the strip layout (NPS_FMT_GT2X), with the
// accumulation on the matrix cores (gfx950, wave64).  Same loop as nps_fused.hip -- tallyAlleles
// (nimpress.nim:32-47), the maxmis decision (:565-571), imputeLocus/SampleDosages (:417-481) and
// `scores[i] += dosage*beta` (:639-641) in ONE read of the matrix -- but nothing in it depends on the genotype
// distribution: no table lookups, no float64 adds per genotype.
//
//   * A row's contribution to a sample is  code * Wc + is_missing * Wm  with code = dosage (3 for a missing
//     genotype), Wc = beta, Wm = (imputed dosage - 3) * beta: a matrix product over the rows.  The weights are
//     56-bit fixed-point integers (scale 2^F per score definition) in sign-magnitude, split into fourteen hexadecimal
//     digits: an FP6 (e2m3) operand byte 00dddd IS the digit d/8, so a digit needs no encoding at all.  The
//     2-bit codes are FP4 (e2m1) operands as they lie: a nibble 00hl is 0, 0.5, 1, 1.5 -- linear in the code -- and
//     the second genotype of a nibble, moved two bits down, is the same again (until round 5: one bit down, in the exponent
//     field, 0, 1, 2, 4 -- a shift less per word, but a second is-missing operand per row for the control waves, whose path
//     sets the step).  Six vector instructions per 16 genotypes make the four operand registers (two code operands, two
//     is-missing operands); v_mfma_scale_f32_16x16x128_f8f6f4 with unit-making block scales accumulates exact integers
//     in float32 (flushed long before 2^24), sixteen columns = fourteen digits, a spare and one NaN flag.
//   * Layout: strips of 2048 samples x superblocks of 128 rows; a unit = 128 rows x 32 samples = 1 KiB,
//     row-major [row][8 bytes], two samples per nibble.  One workgroup (8 waves) owns a strip and streams all
//     its superblocks; a lane's 16-byte load is two whole rows of a unit, so the tally is popcounts on the
//     registers as they arrive.  The same registers, parked in LDS and read back with ds_read_b64_tr_b4 (a
//     16 x 16 transpose of nibbles per 16 lanes), are the MFMA A operand: 32 rows of one sample pair per lane.
//   * Hand-over as in nps_fused.hip: per row one 64-bit agent-scope atomic add per strip
//     (arrivals<<56 | nmissing<<28 | neffect), polled by the strip's control wave a step and a half later; the
//     control wave turns the complete tallies into the superblock's weight digits (16 bytes per row and
//     operand) in LDS, and every wave fetches its B fragments from there with ds_read_b96_tr_b6.
//     Every wait is bounded (NPS_E_TIMEOUT instead of a hang); the grid is launched cooperatively.
//
// Pipeline of a workgroup, step k (one barrier per step):
//     loads of superblock k+3 -> registers | tally k+2 (registers) | park k+1 (registers -> LDS) | barrier |
//     accumulate k (LDS -> MFMA)            control wave: tables of k before the barrier, publish k+2 after it
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "nps_kernels.h"
#include "nps_mx_common.h"

namespace nps {

#ifdef NPS_MX_TIMERS
__device__ unsigned long long g_mx_timers[16][16];  // [wave][phase]: cycles summed over the steps of one workgroup
#define MXT(i) do { if (timing) { const unsigned long long now_ = __builtin_readcyclecounter(); tph[i] += now_ - tlast; tlast = now_; } } while (0)
#else
#define MXT(i) do { } while (0)
#endif
// Per score row, everything that does not depend on the row's tally (MxPre); rows [n_rows, n_pad) are zeroed (their digits
// are fetched with the last superblock's)
__global__ __launch_bounds__(256) void mx_prep_kernel(const nps_row_desc *__restrict__ desc, uint64_t n_rows, uint64_t n_pad,
                                                      DevParams prm, double scale, MxPre *__restrict__ pre) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n_pad) return;
    MxPre o;
    o.c[0] = o.c[1] = o.c[2] = 0u;
    o.flags = 0u;
    o.w1 = o.wfb = 0;
    if (j >= n_rows) {
        pre[j] = o;
        return;
    }
    const double beta = desc[j].beta, eaf = desc[j].eaf;
    const bool rie = desc[j].ref_is_effect != 0;
    const double inf = __builtin_huge_val(), nan = __longlong_as_double(0x7ff8000000000000ll);
    if (!(fabs(beta) < inf)) {
        o.flags = 1u;
    } else {
        o.w1 = __double2ll_rn(beta * scale);
        mx_codes(o.w1, 0u, o.c);
        const double imp = prm.imp_sample == NPS_SAMPLE_HOMREF ? (rie ? 2.0 : 0.0)
                           : (prm.imp_sample == NPS_SAMPLE_FAIL || prm.imp_sample == NPS_SAMPLE_INT_FAIL) ? nan
                                                                                                          : eaf * 2.0;
        const double t = imp * beta;  // the product the reference adds for a missing sample (nimpress.nim:639)
        if (!(fabs(t) < inf)) {
            o.flags = 2u;
            o.wfb = 3 * o.w1;
        } else {
            o.wfb = __double2ll_rn(t * scale);
        }
    }
    pre[j] = o;
}

// NU: units the wave can own (register arrays); GUARD: it may own fewer (ragged last strip) -- without guards the
// compiler pipelines across units; CTL: the control wave (its own code path, so that the data waves' registers
// hold nothing of it)
// DBG (diagnostics builds only): 1 no tally popcounts, 2 no accumulation, 4 no hand-over (no atomics, no wait),
// 8 no weight digits (zero tables), 16 no parking
// GIVEN: a.tally holds the complete tallies (mx_tally_kernel ran before): no popcounts, no hand-over, no waiting --
// the workgroups are independent of each other and the grid need not be resident as a whole
//
// Teams: workgroup b works for strip b % P in team b / P and walks the superblocks team, team + Q, ... of the run
// (local index t <-> superblock team + t Q): whatever addresses memory uses the superblock, whatever alternates
// between the two halves of a double buffer uses t.  The hand-over of a row involves the P strips of ONE team.
template <int NU, bool GUARD, bool CTL, int DBG, bool GIVEN, int UA, int U2, int UB, int UC, bool VS>
static __device__ __forceinline__ void mx_body(const MxArgs &a, char *const smem) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int g = lane >> 4, q = lane & 15;
    const uint32_t strip = blockIdx.x % a.P, team = blockIdx.x / a.P;
    const uint32_t n_t = team < a.n_sb ? (a.n_sb - team + a.Q - 1) / a.Q : 0u;  // superblocks of this team
    if (n_t == 0) return;  // (workgroup-uniform; the plan keeps Q <= n_sb)
    constexpr bool is_ctl = CTL;
    const int nu = strip == a.P - 1 ? (int)a.nu_last : (VS ? (int)a.U : 64);
    // units per wave: UA for data waves 0, 1, U2 for data waves 2, 3 (which share SIMD 2 / 3 with the control waves), UB for data
    // waves 4, 5 (the second data wave of SIMD 0 / 1), UC for the control waves
    const int u0 = wave < 2   ? wave * UA
                   : wave < 4 ? 2 * UA + (wave - 2) * U2
                   : wave < 6 ? 2 * UA + 2 * U2 + (wave - 4) * UB
                              : 2 * UA + 2 * U2 + 2 * UB + (wave - 6) * UC;
    const int crow = lane + 64 * (wave - kDW);  // control waves: the row of the superblock this lane works for
    const int n_my = GUARD ? max(0, min(NU, nu - u0)) : NU;  // wave-uniform
    constexpr bool full = !GUARD;

    char *const slot0 = smem + (size_t)u0 * 1024, *const slot1 = smem + 65536 + (size_t)u0 * 1024;
    const int woff = mx_rowoff(2 * lane);
    const int r1off = mx_rowoff(32 * g + q), r2off = mx_rowoff(32 * g + 16 + q);
    const int fr0 = (32 * g + q) * 16, fr1 = (32 * g + 16 + q) * 16;
    // VS: the strip is a.U consecutive units of the cohort's unit sequence; this wave's first unit is unit g0 of that sequence,
    // i.e. unit g0 % 64 of layout strip g0 / 64, and its units number bnd, bnd + 1, ... lie at the start of the next layout strip.
    // Everything here is wave-uniform and kept in scalar registers (readfirstlane: the compiler does not know that the wave
    // number is uniform); a load's address is a scalar base + the lane's 16 bytes.  (With per-lane base pointers for both
    // layout strips the ten-unit body spilled 174 registers.)
    const uint32_t g0 = VS ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(strip * a.U + (uint32_t)u0)) : 0u;
    const uint32_t ps_a = VS ? g0 >> 6 : strip;
    const int bnd = VS ? 64 - (int)(g0 & 63u) : 64;
    const uint32_t nu_a = VS ? (ps_a == a.P_phys - 1 ? a.nu_last_phys : 64u) : (uint32_t)nu;
    const uint32_t nu_b = VS ? (ps_a + 1 == a.P_phys - 1 ? a.nu_last_phys : 64u) : 0u;
    const v4u *const base =  // (!VS: a pointer per lane, as ever)
        a.units + ((uint64_t)ps_a * 64 * a.n_sb_cohort + (uint64_t)(a.sb0 + team) * nu_a + (VS ? 0u : (uint32_t)u0)) * 64 + (VS ? 0 : lane);
    const uint64_t sb_stride = (uint64_t)nu_a * 64 * a.Q;  // from one superblock of the team to its next
    const char *const sbase_a = reinterpret_cast<const char *>(base) + (uint64_t)(g0 & 63u) * 1024;                   // (VS)
    const char *const sbase_b = reinterpret_cast<const char *>(
        a.units + ((uint64_t)(ps_a + 1) * 64 * a.n_sb_cohort + (uint64_t)(a.sb0 + team) * nu_b) * 64);               // (VS)
    const uint64_t sstride_a = sb_stride * 16, sstride_b = (uint64_t)nu_b * 64 * a.Q * 16;                            // bytes
    const uint32_t lane_off = (uint32_t)lane * 16u;

    v4f C[NU][2];
#pragma unroll
    for (int u = 0; u < NU; ++u) C[u][0] = C[u][1] = v4f{0.f, 0.f, 0.f, 0.f};
    v4u bank[2][NU];  // two superblocks of the wave's units in registers: one being tallied, one waiting to be parked
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int u = 0; u < NU; ++u) bank[s][u] = v4u{0u, 0u, 0u, 0u};

    // (VS) a superblock's two segment bases are made once per superblock.  (The 135 scalars this instantiation keeps in vector
    // lanes are not read back in the steady loop of an unguarded body -- profiles/r06_mx_variants.txt, part 4.)
    struct SbPtr {
        const char *pa, *pb;
    };
    const char *const sbase_b2 = sbase_b - (int64_t)bnd * 1024;
    auto sb_ptr = [&](uint32_t k) -> SbPtr {
        SbPtr r{nullptr, nullptr};
        if constexpr (VS) {
            r.pa = sbase_a + (uint64_t)k * sstride_a;
            r.pb = sbase_b2 + (uint64_t)k * sstride_b;
        }
        return r;
    };
    auto load_unit = [&](uint32_t k, int u, const SbPtr &sp) -> v4u {
        if constexpr (VS) {
            // (an explicit global pointer: the pinned scalars hide where they came from, and a flat load would count on the
            //  LDS counter as well)
            typedef const v4u __attribute__((address_space(1))) *gptr;
            const char *p = (u < bnd ? sp.pa : sp.pb) + (uint32_t)u * 1024u;
            return __builtin_nontemporal_load((gptr)(uintptr_t)(p + lane_off));
        } else {
            return __builtin_nontemporal_load(base + (uint64_t)k * sb_stride + u * 64);
        }
    };
    auto load_sb = [&](uint32_t k, v4u(&dst)[NU]) {
        if (k >= n_t) return;
        const SbPtr sp = sb_ptr(k);
#pragma unroll
        for (int u = 0; u < NU; ++u)
            if (full || u < n_my) dst[u] = load_unit(k, u, sp);
    };

    // tallyAlleles over the wave's units of a superblock: a lane holds rows 2*lane, 2*lane+1 (32 samples each) of
    // every unit.  Codes 01 / 10 / 11 = dosage 1 / dosage 2 / missing: with X = set bits, Y = set high bits,
    // Z = missing,  effect alleles = X + Y - 3 Z.
    struct Tal {
        uint32_t xa = 0, ya = 0, za = 0, xb = 0, yb = 0, zb = 0;
    };
    auto tally_unit = [&](Tal &t, const v4u w) {
        if ((DBG & 1) || GIVEN) return;
        const uint32_t sx = w.x >> 1, sy = w.y >> 1, sz = w.z >> 1, sw = w.w >> 1;
        t.xa = bcnt_acc(w.y, bcnt_acc(w.x, t.xa));
        t.ya = bcnt_acc((w.x & 0xAAAAAAAAu) | (sy & 0x55555555u), t.ya);
        t.za = bcnt_acc((w.x & sx & 0x55555555u) | ((w.y & sy & 0x55555555u) << 1), t.za);
        t.xb = bcnt_acc(w.w, bcnt_acc(w.z, t.xb));
        t.yb = bcnt_acc((w.z & 0xAAAAAAAAu) | (sw & 0x55555555u), t.yb);
        t.zb = bcnt_acc((w.z & sz & 0x55555555u) | ((w.w & sw & 0x55555555u) << 1), t.zb);
    };
    auto tally_add = [&](uint32_t kt, const Tal &t) {
        if (GIVEN) return;
        const unsigned long long pa = (t.xa + t.ya - 3u * t.za) | (t.za << 16), pb = (t.xb + t.yb - 3u * t.zb) | (t.zb << 16);
        unsigned long long *T = reinterpret_cast<unsigned long long *>(smem + kLdsTally) + (kt & 1) * 64 + lane;
        atomicAdd(T, pa | (pb << 32));  // the 16-bit fields of a word never carry: a strip has 2048 samples
    };
    // first half of step k, unit after unit: tally superblock k+2 (bank b_tal), park superblock k+1 (bank b_park ->
    // LDS slot) and refill that register with the unit's rows of superblock k+3
    auto front = [&](uint32_t k, const v4u(&b_tal)[NU], v4u(&b_park)[NU], char *s_park) {
        if (n_my == 0) return;
        Tal t;
        if (k + 3 < n_t) {  // steady state: no per-unit conditions
            const SbPtr sp = sb_ptr(k + 3);
#pragma unroll
            for (int u = 0; u < NU; ++u)
                if (full || u < n_my) {
                    tally_unit(t, b_tal[u]);
                    if (!(DBG & 16)) *reinterpret_cast<v4u *>(s_park + u * 1024 + woff) = b_park[u];
                    b_park[u] = load_unit(k + 3, u, sp);
                }
            tally_add(k + 2, t);
        } else {
            const bool do_t = k + 2 < n_t, do_p = k + 1 < n_t;
#pragma unroll
            for (int u = 0; u < NU; ++u)
                if (full || u < n_my) {
                    if (do_t) tally_unit(t, b_tal[u]);
                    if (do_p) *reinterpret_cast<v4u *>(s_park + u * 1024 + woff) = b_park[u];
                }
            if (do_t) tally_add(k + 2, t);
        }
    };

    // scores += code * Wc + is_missing * Wm for the wave's units of superblock k (parked in `slot`)
    auto accumulate = [&](uint32_t k, const char *slot) {
        if (k >= n_t || n_my == 0 || (DBG & 2)) return;
        const char *tab = smem + kLdsTables + (k % kTabBufs) * 6144;
        v8i Bc, Bme;
        {
            const v3i c0 = tr6(tab + fr0), c1 = tr6(tab + fr1);
            const v3i e0 = tr6(tab + 2048 + fr0), e1 = tr6(tab + 2048 + fr1);
            Bc = v8i{c0[0], c0[1], c0[2], c1[0], c1[1], c1[2], 0, 0};
            Bme = v8i{e0[0], e0[1], e0[2], e1[0], e1[1], e1[2], 0, 0};
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if (full || u < n_my) {
                const v2i t01 = tr4(slot + u * 1024 + r1off), t23 = tr4(slot + u * 1024 + r2off);
                const uint32_t w[4] = {(uint32_t)t01[0], (uint32_t)t01[1], (uint32_t)t23[0], (uint32_t)t23[1]};
                v8i ce = {0, 0, 0, 0, 0, 0, 0, 0}, co = ce, me = ce, mo = ce;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint32_t s1 = w[r] >> 1;
                    ce[r] = (int)(w[r] & 0x33333333u);       // even sample: 00hl = code / 2
                    co[r] = (int)((w[r] >> 2) & 0x33333333u);  // odd sample, moved down: 00hl = code / 2 as well
                    me[r] = (int)(w[r] & s1 & 0x11111111u);  // even sample missing: 0001 = 1/2
                    mo[r] = (int)(w[r] & s1 & 0x44444444u);  // odd sample missing: 0100 = 2
                }
                // block scales (E8M0): codes 0,1,2,3 | 0,1,2,4; missing bit 1; digits d/8 -> d
                C[u][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ce, Bc, C[u][0], 4, 2, 0, 128, 0, 130);
                C[u][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(co, Bc, C[u][1], 4, 2, 0, 128, 0, 130);
                C[u][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(me, Bme, C[u][0], 4, 2, 0, 128, 0, 130);
                C[u][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(mo, Bme, C[u][1], 4, 2, 0, 126, 0, 130);
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

    // ---- control state (waves 6 and 7: 64 rows of every superblock each)
    uint32_t nloci_local = 0;
    double cst_local = 0.0;
    bool timed_out = false;
    MxPre npre;
    npre.c[0] = npre.c[1] = npre.c[2] = npre.flags = 0u;
    npre.w1 = npre.wfb = 0;
    auto ctl_fetch_pre = [&](uint32_t k) {  // the row's precomputed part, a step before it is needed
        const uint64_t row = ((uint64_t)team + (uint64_t)k * a.Q) * 128 + crow;
        if (k < n_t && row < a.n_rows) {
            const v4u *p = reinterpret_cast<const v4u *>(a.pre + row);
            const v4u p0 = p[0], p1 = p[1];
            npre.c[0] = p0.x;
            npre.c[1] = p0.y;
            npre.c[2] = p0.z;
            npre.flags = p0.w;
            npre.w1 = (long long)((unsigned long long)p1.x | ((unsigned long long)p1.y << 32));
            npre.wfb = (long long)((unsigned long long)p1.z | ((unsigned long long)p1.w << 32));
        }
    };
    // the row's tally word of superblock k (a relaxed agent-scope load; GIVEN: a plain one)
    auto ctl_word = [&](uint32_t k, bool &valid) -> unsigned long long {
        const uint64_t row = ((uint64_t)team + (uint64_t)k * a.Q) * 128 + crow;
        valid = k < n_t && row < a.n_rows;
        return !valid  ? 0ull
               : GIVEN ? a.tally[row]
                       : __hip_atomic_load(&a.tally[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    // complete tallies of superblock k (x) -> the row's two operands (beta digits, is-missing digits) in table buffer k % kTabBufs
    MxPreX px;  // (of npre: made by ctl_pre_extra() while the look at the row's word is still under way)
    px.w1d = 0.0;
    px.w3 = px.w4 = 0;
    auto ctl_pre_extra = [&]() {
        px.w1d = (double)npre.w1;
        px.w3 = 3 * npre.w1;
        px.w4 = 4 * npre.w1;
    };
    auto ctl_build = [&](uint32_t k, unsigned long long x, bool valid, bool ok) {
        const uint64_t row = ((uint64_t)team + (uint64_t)k * a.Q) * 128 + crow;
        uint32_t wc[3], wme[3], wmo[3];
        int used;
        double cst;
        mx_row(a, x, valid && ok && !(DBG & 8), row, npre, px, strip == 0 && a.stats != nullptr, wc, wme, wmo, used, cst, false);  // (strip 0 of the row's team)
        char *p = smem + kLdsTables + (k % kTabBufs) * 6144 + crow * 16;
        *reinterpret_cast<v4u *>(p) = v4u{wc[0], wc[1], wc[2], 0u};
        *reinterpret_cast<v4u *>(p + 2048) = v4u{wme[0], wme[1], wme[2], 0u};
        nloci_local += (uint32_t)__popcll(__ballot(used != 0));
        cst_local += cst;
    };
    // wait (bounded) until every row of superblock k is complete, then build its tables.  v1: the tally words were
    // published by every strip more than a step ago: normally one poll.  v2: only reached when the look in the
    // previous step's second half found a word incomplete.
    auto ctl_tables = [&](uint32_t k, bool have_x = false, unsigned long long x_in = 0ull) {
        if (k >= n_t) return;
        const uint64_t row = ((uint64_t)team + (uint64_t)k * a.Q) * 128 + crow;
        bool valid;
        unsigned long long x;
        if (have_x) {
            valid = row < a.n_rows;
            x = valid ? x_in : 0ull;
        } else {
            x = ctl_word(k, valid);
        }
        bool ok = !valid || GIVEN || (uint32_t)(x >> 56) == a.P || (DBG & 4);
        uint32_t spins = 0;
        while (!GIVEN && !__all(ok) && !timed_out) {
            __builtin_amdgcn_s_sleep(1);
            if (!ok) {
                x = __hip_atomic_load(&a.tally[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = (uint32_t)(x >> 56) == a.P;
            }
            if ((++spins & 255u) == 0) {
                const unsigned int t = __hip_atomic_load(a.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (t != 0 || spins >= kMxSpinLimit) {
                    if (lane == 0) __hip_atomic_store(a.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    timed_out = true;
                }
            }
        }
        if (!have_x) ctl_pre_extra();
        ctl_build(k, x, valid, ok);
    };
    // The strip's tallies of superblock kp are complete in LDS (barrier passed).  Same-line atomics are served one
    // after the other at the memory side (~25 ns each), so 245 strips adding to one row's word would take longer
    // than the step and a half the pipeline allows: the strips arrive in groups (of 48, a.grp_strips: 16 at first) on a
    // word of their group, and the strip whose add completes a group (told by the value its add returned) adds the
    // group's sum to the row's word: at most 48 adds in a row on a group's word, at most 6 on the row's.
    const uint32_t grp = strip / a.grp_strips, grp_size = min(a.grp_strips, a.P - grp * a.grp_strips);
    unsigned long long pub_old = 0ull, pub_add = 0ull;
    bool pub_live = false;
    auto ctl_publish_begin = [&](uint32_t kp) {
        if (GIVEN) return;
        uint32_t *T = reinterpret_cast<uint32_t *>(smem + kLdsTally) + (kp & 1) * 128;
        const uint32_t v = T[crow];
        T[crow] = 0u;
        const uint64_t row = ((uint64_t)team + (uint64_t)kp * a.Q) * 128 + crow;
        pub_live = kp < n_t && row < a.n_rows && !(DBG & 4);
        if (pub_live) {
            pub_add = (1ull << 56) | ((unsigned long long)(v >> 16) << 28) | (v & 0xFFFFu);
            pub_old = __hip_atomic_fetch_add(&a.tally1[(uint64_t)grp * a.n_sb * 128 + row], pub_add, __ATOMIC_RELAXED,
                                             __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto ctl_publish_end = [&](uint32_t kp) {
        if (pub_live) {
            const unsigned long long tot = pub_old + pub_add;
            if ((uint32_t)(tot >> 56) == grp_size)
                __hip_atomic_fetch_add(&a.tally[((uint64_t)team + (uint64_t)kp * a.Q) * 128 + crow], tot, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
        }
    };

    // ---- prologue: superblocks 0 and 1 tallied and published, 0 parked, 2 on its way
    if (tid < 256) reinterpret_cast<uint32_t *>(smem + kLdsTally)[tid] = 0u;
    load_sb(0, bank[0]);
    load_sb(1, bank[1]);
    if (is_ctl) ctl_fetch_pre(0);
    __syncthreads();
    if (n_my != 0) {
        Tal t;
#pragma unroll
        for (int u = 0; u < NU; ++u)
            if (full || u < n_my) tally_unit(t, bank[0][u]);
        tally_add(0, t);
    }
    __syncthreads();
    if (is_ctl) {
        ctl_publish_begin(0);
        ctl_publish_end(0);
    }
    if (n_my != 0 && n_t > 1) {
        Tal t;
#pragma unroll
        for (int u = 0; u < NU; ++u)
            if (full || u < n_my) tally_unit(t, bank[1][u]);
        tally_add(1, t);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u)
        if (full || u < n_my) *reinterpret_cast<v4u *>(slot0 + u * 1024 + woff) = bank[0][u];
    load_sb(2, bank[0]);
    __syncthreads();
    if (is_ctl) {
        ctl_publish_begin(1);
        ctl_publish_end(1);
    }

    // ---- steps
    unsigned long long x_look = 0ull;  // (late schedule) the look in flight
#ifdef NPS_MX_TIMERS
    const bool timing = strip == a.P / 2 && team == 0;
    unsigned long long tph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    auto step = [&](uint32_t k, const v4u(&b_tal)[NU], v4u(&b_park)[NU], char *s_park, const char *s_acc) {
        MXT(8);
        // late schedule: the look at the row's word travels while the control wave does its own tallying and parking
        // (its result is waited for by hand: the compiler does not know the load, so its own counted waits only become
        // stricter.  With a load the compiler does know -- the atomic builtin, or a buffer load with sc1 -- it waits
        // with vmcnt(0) at the first use, i.e. for the five unit loads front() has just issued: the branches around
        // them make it lose count.  The price of the asm: x_look must stay in its register until the hand-written
        // wait -- the compiler may not copy it meanwhile, and has no reason to; if a future compiler did, every
        // parity test of tests/test_gpu_mx.py would fail, the tables would be made of stale words).  Before round 4's last change the look was issued, waited for and turned into tables BEFORE front():
        // 24.3 -> 22.5-23.2 ms at 245 strips.
        constexpr bool kUnder = !GIVEN;
        // which of the two counts the look's wait will use, made a scalar HERE: asked for at the wait itself (a
        // v_readfirstlane between front() and the wait, on the control wave's critical path) the pass takes 23.23 ms,
        // this way 22.80 (same box, three alternating runs each)
        int steady = (is_ctl && kUnder) ? __builtin_amdgcn_readfirstlane((k + 3 < n_t && n_my == NU) ? 1 : 0) : 0;
        asm volatile("" : "+s"(steady));
        if (is_ctl && kUnder) {
            const uint64_t row = ((uint64_t)team + (uint64_t)k * a.Q) * 128 + crow;
            x_look = 0ull;
            if (k < n_t && row < a.n_rows)  // (an agent-scope relaxed load, as ctl_word's)
                asm volatile("global_load_dwordx2 %0, %1, off sc1" : "+v"(x_look) : "v"(&a.tally[row]) : "memory");  // ("+v": the load writes the pair the zeroed x_look already lives in -- no merge copy behind it)
        }
        if (is_ctl && !kUnder) ctl_tables(k);  // (given tallies: plain loads)
        MXT(0);
        front(k, b_tal, b_park, s_park);
        MXT(1);
        if (is_ctl && kUnder) {
            ctl_pre_extra();  // (what the operands need of the precomputed row, before the look is waited for)
            asm volatile("" : : "v"(px.w1d), "v"(px.w3), "v"(px.w4));
            // the wave's loads return in order: once at most the NU loads front() issued after the look are
            // outstanding, the look has returned (whatever else -- an add of the last step -- is still under way)
            // ONE statement holds both counts.  With a wait statement per branch (until round 5) the compiler placed a copy
            // of x_look IN FRONT of the vmcnt(0) one -- a read of a register pair whose data might not have arrived, on
            // the last steps of a pass and on ragged strips; tests/test_isa_checks.py found it and now guards the window.
            asm volatile("s_cmp_lg_u32 %1, 0\n\ts_cbranch_scc1 1f\n\ts_waitcnt vmcnt(0)\n\ts_branch 2f\n1:\n\ts_waitcnt vmcnt(%2)\n2:"
                         : "+v"(x_look)
                         : "s"(steady), "n"(NU)
                         : "scc", "memory");
            MXT(6);
            ctl_tables(k, true, x_look);
        }
        MXT(7);
        __syncthreads();
        MXT(2);
        if (is_ctl) {  // the returning add and the loads are in flight during the wave's own accumulation
            ctl_publish_begin(k + 2);
            ctl_fetch_pre(k + 1);
        }
        MXT(3);
        accumulate(k, s_acc);
        MXT(4);
        if (is_ctl) ctl_publish_end(k + 2);
        MXT(5);
        if (((k + 1) & (kFlushSb - 1)) == 0 && k + 1 < n_t) store_c(k / kFlushSb, true);
    };
    const uint32_t n_steps = (n_t + 1) / 2 * 2;
    for (uint32_t k = 0; k < n_steps; k += 2) {
        step(k + 0, bank[0], bank[1], slot1, slot0);
        step(k + 1, bank[1], bank[0], slot0, slot1);
    }
    store_c((n_t - 1) / kFlushSb, false);
#ifdef NPS_MX_TIMERS
    if (timing && lane == 0)
        for (int i = 0; i < 16; ++i) g_mx_timers[wave][i] = tph[i];
#endif
    if (is_ctl && strip == 0) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cst_local += __shfl_down(cst_local, o, 64);
        if (lane == 0) {
            if (nloci_local) atomicAdd(a.nloci, (unsigned long long)nloci_local);
            if (cst_local != 0.0) a.const_sum[2 * team + (wave - kDW)] = cst_local;  // (NaN != 0 is true)
        }
    }
}

// UA / UB / UC: units of data waves 0..3 / of data waves 4, 5 / of a control wave.  Three splits are built (round 5):
//   9 + 5   the control waves carry five units beside their 64 rows each;
//   10 + 2  (253 VGPRs, possible since the rows have ONE is-missing operand) the control waves -- whose path is the step's
//           longest: look, operands, publication AND their own units -- carry two.  Where a strip's workgroup is alone with
//           its rows (one row team: more than 128 strips) that shortens the step: 500 000 samples 22.0 -> 21.3 ms, 400 000
//           18.8 -> 18.6; with two or more teams per strip the nine-unit split is as fast or faster (200 000 samples: 9.1
//           against 9.7 ms), so the launcher picks by the plan.
//   10 + 9 + 2, VS  strips of 62 units cut from the unit sequence instead of the layout's 64 (MxArgs.U): more strips, hence more
//           compute units at work, and nineteen units instead of twenty on the two SIMDs that set the step.
template <int DBG, bool GIVEN, int UA, int U2, int UB, int UC, bool VS, int U = (VS ? 62 : 64)>
__global__ __launch_bounds__(kMxThreads, (kDW + 2) / 4) void fused_mx_kernel(const MxArgs a) {
    static_assert(kDW == 6 && 2 * UA + 2 * U2 + 2 * UB + 2 * UC == U && UC >= 1 && (VS || U == 64), "units of a strip");
    extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 slots][64 units][1 KiB] | tables | tallies
    const int wave = threadIdx.x >> 6;
    const int nu = blockIdx.x % a.P == a.P - 1 ? (int)a.nu_last : (VS ? (int)a.U : 64);
    // both bodies pass the same barriers; which one a wave runs is wave-uniform
    if (wave >= kDW) {
        // the control waves are the step's critical path and share their SIMD with a data wave: up to ~200 strips per
        // team they issue first (200 000 samples 9.95 -> 9.29 ms, 400 000 20.5 -> 19.6, 100 000 and 250 000 +1 %); at
        // 245 strips the step is set by the hand-over chain instead and the priority costs up to 2.5 %
        if (a.ctl_prio) __builtin_amdgcn_s_setprio(3);
        // (the unguarded body for every strip but a ragged last one: the guarded loops do not pipeline across units)
        const int uc0 = 2 * UA + 2 * U2 + 2 * UB + (wave - kDW) * UC;
        if (nu - uc0 >= UC)
            mx_body<UC, false, true, DBG, GIVEN, UA, U2, UB, UC, VS>(a, smem);
        else
            mx_body<UC, true, true, DBG, GIVEN, UA, U2, UB, UC, VS>(a, smem);
    } else if (wave < 2) {
        if (nu - wave * UA >= UA)
            mx_body<UA, false, false, DBG, GIVEN, UA, U2, UB, UC, VS>(a, smem);
        else
            mx_body<UA, true, false, DBG, GIVEN, UA, U2, UB, UC, VS>(a, smem);
    } else if (wave < 4) {
        if (nu - (2 * UA + (wave - 2) * U2) >= U2)
            mx_body<U2, false, false, DBG, GIVEN, UA, U2, UB, UC, VS>(a, smem);
        else
            mx_body<U2, true, false, DBG, GIVEN, UA, U2, UB, UC, VS>(a, smem);
    } else {
        if (nu - (2 * UA + 2 * U2 + (wave - 4) * UB) >= UB)
            mx_body<UB, false, false, DBG, GIVEN, UA, U2, UB, UC, VS>(a, smem);
        else
            mx_body<UB, true, false, DBG, GIVEN, UA, U2, UB, UC, VS>(a, smem);
    }
}

// Epilogue of a pass: the sixteen digit sums of a sample -> float64, plus the pass's locus constants, into chunk 0
// of the context's partial scores; the tally words go back to zero; a raised bounded-wait word is recorded.
// keep (or nullptr): the run's complete whole-row tallies are copied there (arrival count stripped) before the words are
// zeroed -- a cohort's kept tallies as a by-product of its first single-read pass.
__global__ __launch_bounds__(256) void mx_fold_kernel(const float *__restrict__ cpart, uint32_t n_sb, uint32_t Q, uint32_t P, uint32_t U,
                                                      uint64_t n, double inv_scale, const double *__restrict__ const_sum,
                                                      double *__restrict__ part0, int overwrite,
                                                      unsigned long long *__restrict__ tally, uint64_t n_tally,
                                                      unsigned long long *__restrict__ tally1, uint64_t n_tally1,
                                                      unsigned int *__restrict__ timeout,
                                                      unsigned long long *__restrict__ status,
                                                      unsigned long long *__restrict__ keep, uint64_t n_keep) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t nthreads = (uint64_t)gridDim.x * 256;
    for (uint64_t j = i; j < n_keep; j += nthreads) keep[j] = tally[j] & 0x00FFFFFFFFFFFFFFull;
    for (uint64_t j = i; j < n_tally; j += nthreads) tally[j] = 0ull;
    for (uint64_t j = i; j < n_tally1; j += nthreads) tally1[j] = 0ull;
    if (i == 0 && timeout) {
        if (*timeout) atomicOr(status, 1ull);
        *timeout = 0u;
    }
    __shared__ double s_const;
    if (threadIdx.x == 0) {  // the teams' locus constants, in slot order (the host zeroes the slots after this launch)
        double c = 0.0;
        for (uint32_t q = 0; q < 2 * Q; ++q) c += const_sum[q];
        s_const = c;
    }
    __syncthreads();
    if (i >= n) return;
    // (strips of U units: the layout's 64, or the first form's virtual strips)
    const uint32_t gu = (uint32_t)(i >> 5), s = (uint32_t)i & 31u;
    const uint64_t strip = U == 64u ? gu >> 6 : gu / U;
    const uint32_t unit = U == 64u ? gu & 63u : gu - (uint32_t)strip * U;
    const uint32_t e = s & 1u, ii = s >> 1;
    // The sixteen digit sums of a sample are exact integers (float32 below 2^24); they are recombined as exact integers too:
    // value = hi 2^28 + lo with the seven low hexadecimal digits in lo, the seven high ones in hi (each below 2^52 per
    // flush), summed over teams and flushes in int64, and rounded to float64 ONCE.  (Until round 5: Horner in float64 per
    // flush, then a float64 sum -- 1 ulp of the largest partial sum per step.  Exact sums make the result independent of
    // how the rows were split into teams -- kept tallies, two reads and the single read now give the same bits -- and let
    // the rows taken back below cancel to exactly zero.)
    long long hi = 0, lo = 0;
    bool isnan_ = false;
    for (uint32_t tm = 0; tm < Q && tm < n_sb; ++tm) {
        const uint32_t n_flush = ((n_sb - tm + Q - 1) / Q + kFlushSb - 1) / kFlushSb;
        for (uint32_t f = 0; f < n_flush; ++f) {
            const float *t =
                cpart + (((((uint64_t)f * Q + tm) * P + strip) * 64 + unit) * 2 + e) * 256 + ((ii >> 2) * 16) * 4 + (ii & 3);
            long long l = 0, h = 0;
#pragma unroll
            for (int d = 0; d < 7; ++d) {
                l += (long long)t[d * 4] * (1ll << (4 * d));
                h += (long long)t[(d + 7) * 4] * (1ll << (4 * d));
            }
            lo += l;
            hi += h;
            isnan_ = isnan_ || t[15 * 4] != 0.f;
        }
    }
    const double total = (double)hi * 268435456.0 + (double)lo;
    double r = total * inv_scale + s_const;
    if (isnan_) r = __longlong_as_double(0x7ff8000000000000ll);
    part0[i] = overwrite ? r : part0[i] + r;
}

// Whole-row tallies of a run whose strips do not fit one cooperative grid (plan.given): one workgroup = one
// superblock x a group of 16 strips (1 MiB, read once); a lane holds rows 2 lane, 2 lane + 1 of every unit its wave
// reads, exactly as in the fused kernel; the four waves meet in LDS and the workgroup adds one word per row.
__global__ __launch_bounds__(256) void mx_tally_kernel(const v4u *__restrict__ units, uint64_t n_sb_cohort, uint32_t sb0,
                                                       uint32_t P, uint32_t nu_last, unsigned long long *__restrict__ tally) {
    __shared__ uint32_t s_eff[128], s_mis[128];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint32_t sb = blockIdx.x, grp = blockIdx.y;
    if (tid < 128) s_eff[tid] = s_mis[tid] = 0u;
    __syncthreads();
    uint32_t xa = 0, ya = 0, za = 0, xb = 0, yb = 0, zb = 0;
    const uint32_t s_end = min(P, grp * 16 + 16);
    for (uint32_t strip = grp * 16; strip < s_end; ++strip) {
        const uint32_t nu = strip == P - 1 ? nu_last : 64u;
        const v4u *base = units + ((uint64_t)strip * 64 * n_sb_cohort + (uint64_t)(sb0 + sb) * nu) * 64 + lane;
        for (uint32_t u = wave; u < nu; u += 4) {
            const v4u w = __builtin_nontemporal_load(base + (uint64_t)u * 64);
            const uint32_t sx = w.x >> 1, sy = w.y >> 1, sz = w.z >> 1, sw = w.w >> 1;
            xa = bcnt_acc(w.y, bcnt_acc(w.x, xa));
            ya = bcnt_acc((w.x & 0xAAAAAAAAu) | (sy & 0x55555555u), ya);
            za = bcnt_acc((w.x & sx & 0x55555555u) | ((w.y & sy & 0x55555555u) << 1), za);
            xb = bcnt_acc(w.w, bcnt_acc(w.z, xb));
            yb = bcnt_acc((w.z & 0xAAAAAAAAu) | (sw & 0x55555555u), yb);
            zb = bcnt_acc((w.z & sz & 0x55555555u) | ((w.w & sw & 0x55555555u) << 1), zb);
        }
    }
    atomicAdd(&s_eff[2 * lane], xa + ya - 3u * za);
    atomicAdd(&s_mis[2 * lane], za);
    atomicAdd(&s_eff[2 * lane + 1], xb + yb - 3u * zb);
    atomicAdd(&s_mis[2 * lane + 1], zb);
    __syncthreads();
    if (tid < 128)
        atomicAdd(&tally[(uint64_t)sb * 128 + tid], ((unsigned long long)s_mis[tid] << 28) | (unsigned long long)s_eff[tid]);
}

// ---- packing: generator, plain rows <-> units -------------------------------------------------------------
static __device__ __forceinline__ uint64_t xmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// one thread = one row of one unit (32 samples); block = 128 rows x 2 units; grid = (unit pairs, superblocks)
__global__ __launch_bounds__(256) void synth_gt2x_kernel(unsigned long long *__restrict__ units, uint64_t n_units,
                                                         uint64_t n_sb, uint64_t n_samples, uint64_t sb0,
                                                         uint64_t gen_row0, uint64_t n_rows, uint64_t seed,
                                                         const uint32_t *__restrict__ t_het,
                                                         const uint32_t *__restrict__ t_hom,
                                                         const uint32_t *__restrict__ t_miss) {
    const uint64_t unit = (uint64_t)blockIdx.x * 2 + (threadIdx.x >> 7);
    const uint32_t rho = threadIdx.x & 127;
    const uint64_t sb = blockIdx.y;  // relative to sb0
    if (unit >= n_units) return;
    const uint64_t r = sb * 128 + rho;  // row relative to the first row written = index into the thresholds
    unsigned long long out = 0;
    if (r < n_rows) {
        const uint64_t key = xmix64(seed ^ ((gen_row0 + r) * 0xD1B54A32D192ED03ull));
        const uint32_t th = t_het[r], tm = t_hom[r], tmi = t_miss[r];
        for (int k = 0; k < 32; ++k) {
            const uint64_t s = unit * 32 + k;
            if (s < n_samples) {
                const uint64_t h = xmix64(key + s);
                const uint32_t gq = (uint32_t)h, ms = (uint32_t)(h >> 32);
                const unsigned long long c = ms < tmi ? 3ull : (gq < tm ? 2ull : (gq < th ? 1ull : 0ull));
                out |= c << (2 * k);
            }
        }
    }
    units[gt2x_unit_index(unit, sb0 + sb, n_units, n_sb) * 128 + rho] = out;
}

// plain rows (C-ABI order and codes, row stride src_stride_words) -> units; rows past n_rows of the last
// superblock become zero.  One thread = one row of one unit.
__global__ __launch_bounds__(256) void rows_to_gt2x_kernel(const uint32_t *__restrict__ src, uint64_t src_stride_words,
                                                           uint64_t n_words, uint64_t n_rows,
                                                           unsigned long long *__restrict__ units, uint64_t n_units,
                                                           uint64_t n_sb, uint64_t sb0) {
    const uint64_t unit = (uint64_t)blockIdx.x * 2 + (threadIdx.x >> 7);
    const uint32_t rho = threadIdx.x & 127;
    const uint64_t sb = blockIdx.y;
    if (unit >= n_units) return;
    const uint64_t r = sb * 128 + rho;
    unsigned long long out = 0;
    if (r < n_rows) {
        const uint32_t *p = src + r * src_stride_words + unit * 2;
        const unsigned long long lo = unit * 2 < n_words ? p[0] : 0u, hi = unit * 2 + 1 < n_words ? p[1] : 0u;
        const unsigned long long x = lo | (hi << 32);
        out = x ^ ((x >> 1) & 0x5555555555555555ull);  // NPS_CODE_* (2 = missing, 3 = dosage 2) -> 2 = dosage 2, 3 = missing
    }
    units[gt2x_unit_index(unit, sb0 + sb, n_units, n_sb) * 128 + rho] = out;
}

__global__ __launch_bounds__(256) void gt2x_to_rows_kernel(const unsigned long long *__restrict__ units, uint64_t n_units,
                                                           uint64_t n_sb, uint64_t row0, uint64_t n_rows,
                                                           uint32_t *__restrict__ dst, uint64_t dst_stride_words,
                                                           uint64_t n_words) {
    const uint64_t unit = (uint64_t)blockIdx.x * 2 + (threadIdx.x >> 7);
    const uint64_t r = (uint64_t)blockIdx.y * 128 + (threadIdx.x & 127);  // relative to row0
    if (unit >= n_units || r >= n_rows) return;
    const uint64_t row = row0 + r;
    unsigned long long x = units[gt2x_unit_index(unit, row >> 7, n_units, n_sb) * 128 + (row & 127)];
    x ^= (x >> 1) & 0x5555555555555555ull;  // the recoding is its own inverse
    uint32_t *p = dst + r * dst_stride_words + unit * 2;
    if (unit * 2 < n_words) p[0] = (uint32_t)x;
    if (unit * 2 + 1 < n_words) p[1] = (uint32_t)(x >> 32);
}

// a NPS_FMT_GT2 cohort (group-interleaved, plane-separated words, plain layout) -> units.  Workgroup = one superblock
// x 32 word columns (16 units): the 128 x 32 word tile comes in as 32 row groups x 512 contiguous bytes (a thread that
// fetches its own two words reads 32-byte pieces: 3.7 TB/s read + written), goes through LDS, and every unit leaves as
// 1 KiB of consecutive rows.
__global__ __launch_bounds__(256) void gt2_to_gt2x_kernel(const uint4 *__restrict__ src, uint64_t stride_words,
                                                          uint64_t n_words, uint64_t n_row_groups,
                                                          unsigned long long *__restrict__ units, uint64_t n_units,
                                                          uint64_t n_sb, uint64_t sb_off) {
    __shared__ uint32_t tile[128][33];  // [row][word column], padded
    const int t = threadIdx.x;
    const uint64_t sb = sb_off + blockIdx.y, c0 = (uint64_t)blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int e = t + 256 * i;  // (row group 0..31, column 0..31)
        const uint64_t rg = sb * 32 + (e >> 5), c = c0 + (e & 31);
        uint4 q = make_uint4(0, 0, 0, 0);
        if (rg < n_row_groups && c < n_words) q = src[rg * stride_words + c];
        const int r = 4 * (e >> 5);
        tile[r][e & 31] = q.x, tile[r + 1][e & 31] = q.y, tile[r + 2][e & 31] = q.z, tile[r + 3][e & 31] = q.w;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int e = t + 256 * i, u = e >> 7, rho = e & 127;
        const uint64_t unit = (uint64_t)blockIdx.x * 16 + u;
        if (unit >= n_units) continue;
        const unsigned long long lo = word_from_planes(tile[rho][2 * u]), hi = word_from_planes(tile[rho][2 * u + 1]);
        const unsigned long long x = lo | (hi << 32);
        units[gt2x_unit_index(unit, sb, n_units, n_sb) * 128 + rho] = x ^ ((x >> 1) & 0x5555555555555555ull);
    }
}

// ---- host side ------------------------------------------------------------------------------------------
static hipError_t grid_2d(uint64_t n_units, uint64_t n_sb, dim3 *grid) {
    if (n_sb > 65535) return hipErrorInvalidValue;  // callers split the superblock range
    *grid = dim3((uint32_t)((n_units + 1) / 2), (uint32_t)n_sb);
    return hipSuccess;
}

hipError_t launch_synth_gt2x(hipStream_t st, void *d_units, uint64_t n_samples, uint64_t n_rows_cohort, uint64_t row0,
                             uint64_t gen_row0, uint64_t n_rows, uint64_t seed, const uint32_t *d_t_het,
                             const uint32_t *d_t_hom, const uint32_t *d_t_miss) {
    if (n_rows == 0 || n_samples == 0) return hipSuccess;
    if (row0 & 127) return hipErrorInvalidValue;
    const MxGeom gm = mx_geom(n_samples, n_rows_cohort);
    dim3 grid;
    hipError_t e = grid_2d(gm.n_units, (n_rows + 127) / 128, &grid);
    if (e != hipSuccess) return e;
    (void)hipGetLastError();
    hipLaunchKernelGGL(synth_gt2x_kernel, grid, dim3(256), 0, st, (unsigned long long *)d_units, gm.n_units, gm.n_sb,
                       n_samples, row0 >> 7, gen_row0, n_rows, seed, d_t_het, d_t_hom, d_t_miss);
    return hipGetLastError();
}

hipError_t launch_rows_to_gt2x(hipStream_t st, const uint32_t *d_src, uint64_t src_stride_words, uint64_t n_samples,
                               uint64_t n_rows_cohort, uint64_t row0, uint64_t n_rows, void *d_units) {
    if (n_rows == 0 || n_samples == 0) return hipSuccess;
    if (row0 & 127) return hipErrorInvalidValue;
    const MxGeom gm = mx_geom(n_samples, n_rows_cohort);
    dim3 grid;
    hipError_t e = grid_2d(gm.n_units, (n_rows + 127) / 128, &grid);
    if (e != hipSuccess) return e;
    (void)hipGetLastError();
    hipLaunchKernelGGL(rows_to_gt2x_kernel, grid, dim3(256), 0, st, d_src, src_stride_words, words_for(n_samples), n_rows,
                       (unsigned long long *)d_units, gm.n_units, gm.n_sb, row0 >> 7);
    return hipGetLastError();
}

hipError_t launch_gt2x_to_rows(hipStream_t st, const void *d_units, uint64_t n_samples, uint64_t n_rows_cohort,
                               uint64_t row0, uint64_t n_rows, uint32_t *d_dst, uint64_t dst_stride_words) {
    if (n_rows == 0 || n_samples == 0) return hipSuccess;
    const MxGeom gm = mx_geom(n_samples, n_rows_cohort);
    dim3 grid;
    hipError_t e = grid_2d(gm.n_units, (n_rows + 127) / 128, &grid);
    if (e != hipSuccess) return e;
    (void)hipGetLastError();
    hipLaunchKernelGGL(gt2x_to_rows_kernel, grid, dim3(256), 0, st, (const unsigned long long *)d_units, gm.n_units,
                       gm.n_sb, row0, n_rows, d_dst, dst_stride_words, words_for(n_samples));
    return hipGetLastError();
}

hipError_t launch_gt2_to_gt2x(hipStream_t st, const uint32_t *d_src, uint64_t stride_words, uint64_t n_samples,
                              uint64_t n_rows, void *d_units) {
    if (n_rows == 0 || n_samples == 0) return hipSuccess;
    const MxGeom gm = mx_geom(n_samples, n_rows);
    (void)hipGetLastError();
    for (uint64_t sb = 0; sb < gm.n_sb; sb += 32768) {
        const uint64_t k = std::min<uint64_t>(32768, gm.n_sb - sb);
        hipLaunchKernelGGL(gt2_to_gt2x_kernel, dim3((uint32_t)((gm.n_units + 15) / 16), (uint32_t)k), dim3(256), 0, st,
                           reinterpret_cast<const uint4 *>(d_src), stride_words, words_for(n_samples), (n_rows + 3) / 4,
                           (unsigned long long *)d_units, gm.n_units, gm.n_sb, sb);
    }
    return hipGetLastError();
}

hipError_t mx_plan(int device, uint64_t n_samples, uint64_t n_rows, bool two_pass, MxPlan *plan) {
    *plan = MxPlan{};
    if (n_samples == 0 || n_rows == 0 || n_samples >= (1ull << 27)) return hipSuccess;
    static int cus_cached[64] = {0};
    int cus = device >= 0 && device < 64 ? cus_cached[device] : 0;
    if (!cus) {
        hipDeviceProp_t prop;
        hipError_t e = hipGetDeviceProperties(&prop, device);
        if (e != hipSuccess) return e;
        cus = prop.multiProcessorCount;
        if (device >= 0 && device < 64) cus_cached[device] = cus;
    }
    const MxGeom gm = mx_geom(n_samples, n_rows);
    if (gm.n_sb > 0x1fffffffull || gm.P > 65535u) return hipSuccess;
    plan->P = gm.P;
    plan->nu_last = gm.nu_last;
    plan->n_sb = (uint32_t)gm.n_sb;
    // One strip per compute unit, the whole grid resident (8-bit arrival count): the single-read kernel.  Fewer strips
    // than compute units: Q row teams per strip fill the chip (superblock k belongs to team k % Q).  More strips than
    // compute units (N > 2048 x CUs): the tallies come from their own pass and the accumulation runs as an ordinary
    // grid of P x Q independent workgroups, about four per compute unit for an even tail.
    plan->given = two_pass || gm.P > (uint32_t)cus || gm.P > 255;
    uint64_t q = (uint64_t)cus / gm.P;
    if (plan->given) {
        // independent workgroups, one resident per compute unit at a time: P x Q of them run in ceil(P Q / CUs) rounds.
        // Of the team counts that give between two and eight rounds, the one whose last round is fullest (147 strips:
        // Q = 7 would be 1029 workgroups = four rounds and five stragglers; Q = 12 is 1764 = seven rounds, 98 % full)
        const uint64_t lo = std::max<uint64_t>(1, ((uint64_t)2 * cus + gm.P - 1) / gm.P), hi = std::max<uint64_t>(lo, (uint64_t)8 * cus / gm.P);
        double best = -1.0;
        q = lo;
        for (uint64_t t = lo; t <= hi; ++t) {
            const uint64_t wg = (uint64_t)gm.P * t, rounds = (wg + cus - 1) / cus;
            const double fill = (double)wg / (double)(rounds * cus);
            if (fill > best + 1e-9) {
                best = fill;
                q = t;
            }
        }
#ifdef NPS_DIAGNOSTICS
        if (getenv("NPS_MXG_Q")) q = (uint64_t)std::max(1, atoi(getenv("NPS_MXG_Q")));
#endif
    }
    q = std::max<uint64_t>(1, std::min<uint64_t>(q, gm.n_sb));
    plan->Q = (uint32_t)q;
    const uint64_t n_t = (gm.n_sb + q - 1) / q;  // superblocks of the longest team
    plan->n_flush = (uint32_t)((n_t + kFlushSb - 1) / kFlushSb);
    // virtual strips of 62 units for the first form: only where one row team per strip is all there is (more than half the
    // compute units are strips already) and the finer cut still fits the resident grid
    plan->U = 64;
    plan->Pv = gm.P;
    plan->nu_last_v = gm.nu_last;
    if (!plan->given && q == 1) {
        const uint64_t total_units = (uint64_t)(gm.P - 1) * 64 + gm.nu_last, pv = (total_units + 61) / 62;
        if (pv > gm.P && pv <= (uint64_t)cus && pv <= 255) {
            plan->U = 62;
            plan->Pv = (uint32_t)pv;
            plan->nu_last_v = (uint32_t)(total_units - (pv - 1) * 62);
        }
    }
    plan->cpart_floats = (uint64_t)plan->n_flush * q * std::max(gm.P, plan->Pv) * 64 * 2 * 256;
    plan->ok = true;
    return hipSuccess;
}

hipError_t launch_mx_tally(hipStream_t st, const MxPlan &plan, const void *d_units, uint64_t n_sb_cohort, uint64_t sb0,
                           uint64_t n_samples, unsigned long long *d_tally) {
    (void)n_samples;
    (void)hipGetLastError();
    const uint32_t groups = (plan.P + 15) / 16;
    for (uint32_t s = 0; s < plan.n_sb; s += 1u << 30) {  // (grid.x holds 2^31 - 1 blocks)
        const uint32_t k = std::min<uint32_t>(1u << 30, plan.n_sb - s);
        hipLaunchKernelGGL(mx_tally_kernel, dim3(k, groups), dim3(256), 0, st, (const v4u *)d_units, n_sb_cohort,
                           (uint32_t)sb0 + s, plan.P, plan.nu_last, d_tally + (uint64_t)s * 128);
    }
    return hipGetLastError();
}

hipError_t launch_mx_prep(hipStream_t st, const nps_row_desc *d_desc, uint64_t n_rows, DevParams prm, int F, void *d_pre) {
    (void)hipGetLastError();
    const uint64_t n_pad = (n_rows + 127) / 128 * 128;
    hipLaunchKernelGGL(mx_prep_kernel, dim3((uint32_t)((n_pad + 255) / 256)), dim3(256), 0, st, d_desc, n_rows, n_pad, prm,
                       std::ldexp(1.0, F), (MxPre *)d_pre);
    return hipGetLastError();
}

#ifndef NPS_MX_FEW_STRIPS
#define NPS_MX_FEW_STRIPS 176
#endif
constexpr uint32_t kMxFewStrips = NPS_MX_FEW_STRIPS;  // layout strips up to which the 9|10|8|4 split is used (one row team)

hipError_t launch_fused_mx(hipStream_t st, const MxPlan &plan, const void *d_units, uint64_t n_sb_cohort, uint64_t sb0,
                           uint64_t n_samples, uint64_t n_rows, const nps_row_desc *d_desc, DevParams prm,
                           int64_t t_maxmis, int F, void *d_pre, unsigned long long *d_tally,
                           unsigned long long *d_tally1, nps_locus_stat *d_stats,
                           unsigned long long *d_nloci, double *d_const_sum, float *d_cpart, unsigned int *d_timeout) {
    {
        hipError_t pe = launch_mx_prep(st, d_desc, n_rows, prm, F, d_pre);
        if (pe != hipSuccess) return pe;
    }
    // one row team (more than 128 strips): the control waves carry two units instead of five (see fused_mx_kernel)
    if (plan.given) return hipErrorInvalidValue;  // (given tallies: launch_mx_given, nps_mxg.hip)
    const bool light_ctl = plan.Q == 1, vstrips = light_ctl && plan.U < 64;
    const void *fn = nullptr;
    if (vstrips) {
        switch (plan.U) {
        // 10|10|9|2 where the strips fill the chip; with fewer strips (up to ~176 of the layout's: 360 000 samples) the step is
        // the two data waves of SIMD 0 / 1 (nineteen units x ~300 cycles) while the control waves wait 1 600 cycles at the
        // barrier (profiles/r06_mx_timers.txt, 300 000 samples): 9|10|8|4 moves two units per SIMD pair over to them
        // (300 000 samples: 16.9 ms against 18.0-18.5; no gain from 400 000 on, profiles/r06_mx_variants.txt part 7)
        case 62:
            fn = plan.P <= kMxFewStrips ? (const void *)fused_mx_kernel<0, false, 9, 10, 8, 4, true, 62>
                                        : (const void *)fused_mx_kernel<0, false, 10, 10, 9, 2, true, 62>;
            break;
        default: return hipErrorInvalidValue;
        }
    } else {
        fn = light_ctl ? (const void *)fused_mx_kernel<0, false, 10, 10, 10, 2, false> : (const void *)fused_mx_kernel<0, false, kUD, kUD, kUD, kUC, false>;
    }
#ifdef NPS_DIAGNOSTICS
    // diagnostics builds only (tools/mkexp.sh -DNPS_DIAGNOSTICS): NPS_MX_DEBUG selects a kernel with parts left out (1 no tally
    // popcounts, 4 no hand-over, 5 neither), NPS_MX_SPLIT=UA,U2,UB,UC another split of a 62-unit strip over the eight waves
    static const int dbg = getenv("NPS_MX_DEBUG") ? atoi(getenv("NPS_MX_DEBUG")) : 0;
    if (vstrips && plan.U == 62) {
        switch (dbg) {
        case 1: fn = (const void *)fused_mx_kernel<1, false, 10, 10, 9, 2, true, 62>; break;
        case 4: fn = (const void *)fused_mx_kernel<4, false, 10, 10, 9, 2, true, 62>; break;
        case 5: fn = (const void *)fused_mx_kernel<5, false, 10, 10, 9, 2, true, 62>; break;
        default: break;
        }
        if (const char *sp = getenv("NPS_MX_SPLIT")) {
            int ua = 0, u2 = 0, ub = 0, uc = 0;
            if (sscanf(sp, "%d,%d,%d,%d", &ua, &u2, &ub, &uc) == 4) {
                const int key = ua * 1000 + u2 * 100 + ub * 10 + uc;
                switch (key) {
                case 10 * 1000 + 10 * 100 + 8 * 10 + 3: fn = (const void *)fused_mx_kernel<0, false, 10, 10, 8, 3, true, 62>; break;
                case 9 * 1000 + 10 * 100 + 9 * 10 + 3: fn = (const void *)fused_mx_kernel<0, false, 9, 10, 9, 3, true, 62>; break;
                case 9 * 1000 + 10 * 100 + 8 * 10 + 4: fn = (const void *)fused_mx_kernel<0, false, 9, 10, 8, 4, true, 62>; break;
                case 8 * 1000 + 10 * 100 + 8 * 10 + 5: fn = (const void *)fused_mx_kernel<0, false, 8, 10, 8, 5, true, 62>; break;
                case 8 * 1000 + 10 * 100 + 9 * 10 + 4: fn = (const void *)fused_mx_kernel<0, false, 8, 10, 9, 4, true, 62>; break;
                case 7 * 1000 + 10 * 100 + 8 * 10 + 6: fn = (const void *)fused_mx_kernel<0, false, 7, 10, 8, 6, true, 62>; break;
                default: fprintf(stderr, "NPS_MX_SPLIT %s is not compiled in\n", sp); return hipErrorInvalidValue;
                }
            }
        }
    }
#endif
    static const void *attr_set[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    {
        int slot = 0;
        while (slot < 8 && attr_set[slot] && attr_set[slot] != fn) ++slot;
        if (slot == 8) slot = 0;  // (more kernels than slots: set the attribute again)
        if (attr_set[slot] != fn) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes);
            if (e != hipSuccess) return e;
            attr_set[slot] = fn;
        }
    }
    MxArgs a;
    a.units = (const v4u *)d_units;
    a.n_sb_cohort = n_sb_cohort;
    a.sb0 = (uint32_t)sb0;
    a.n_sb = plan.n_sb;
    a.n_rows = n_rows;
    a.n_samples = n_samples;
    a.P = vstrips ? plan.Pv : plan.P;
    a.nu_last = vstrips ? plan.nu_last_v : plan.nu_last;
    a.U = vstrips ? plan.U : 64u;
    a.P_phys = plan.P;
    a.nu_last_phys = plan.nu_last;
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
    a.ctl_prio = plan.P <= 208 ? 1u : 0u;
    // strips per first-stage group (measured at 16 / 32 / 48 / 64, ms per 1M rows: 200 000 samples 9.48 / 9.27 / 9.07 /
    // 9.09, 400 000 20.1 / 19.6 / 19.6 / 19.7, 500 000 23.80 / 23.55 / 23.48 / 23.49; 8: slower, 128: slower)
    a.grp_strips = 48u;
#ifdef NPS_MX_GRP_ENV
    if (getenv("NPS_MX_GRP")) {  // (experiment builds) 0 = balanced groups of at most 64
        const int g = atoi(getenv("NPS_MX_GRP"));
        const uint32_t ng = (plan.P + 63) / 64;
        a.grp_strips = g > 0 ? (uint32_t)std::max(16, g) : (plan.P + ng - 1) / ng;
    }
    if (getenv("NPS_MX_PRIO")) a.ctl_prio = (uint32_t)atoi(getenv("NPS_MX_PRIO"));
#endif
    const dim3 grid((vstrips ? plan.Pv : plan.P) * plan.Q);
    void *args[] = {&a};
#ifdef NPS_MX_TIMERS
    {
        hipError_t e = hipLaunchCooperativeKernel(fn, grid, dim3(kMxThreads), args, kLdsBytes, st);
        if (e != hipSuccess) return e;
        (void)hipStreamSynchronize(st);
        unsigned long long h[16][16];
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_mx_timers), sizeof(h));
        // phases of a step, in order (control waves: 0 = look issue, early/mid table work; 6 = what is left of the look's
        // round trip after front(); 7 = tables)
        static const char *nm[16] = {"look-issue", "front", "barrier", "publish-begin", "accumulate", "publish-end", "look-wait",
                                     "tables", "loop/flush", "-", "-", "-", "-", "-", "-", "-"};
        for (int w = 0; w < kDW + 2; ++w) {
            fprintf(stderr, "mx timers wave %d (cycles per step):", w);
            for (int i = 0; i < 9; ++i) fprintf(stderr, "  %s %.0f", nm[i], (double)h[w][i] / plan.n_sb);
            fprintf(stderr, "\n");
        }
        return hipSuccess;
    }
#endif
    return hipLaunchCooperativeKernel(fn, grid, dim3(kMxThreads), args, kLdsBytes, st);
}

hipError_t launch_mx_fold(hipStream_t st, const MxPlan &plan, const float *d_cpart, uint64_t n_samples, int F,
                          const double *d_const_sum, double *d_part0, int overwrite, unsigned long long *d_tally,
                          uint64_t n_tally, unsigned long long *d_tally1, uint64_t n_tally1, unsigned int *d_timeout,
                          unsigned long long *d_status, bool vstrips, unsigned long long *d_keep, uint64_t n_keep) {
    (void)hipGetLastError();
    if (n_keep > n_tally) return hipErrorInvalidValue;
    const uint64_t blocks = std::max<uint64_t>(std::max<uint64_t>(1, (n_samples + 255) / 256), std::min<uint64_t>(4096, n_tally1 / 1024));
    hipLaunchKernelGGL(mx_fold_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, d_cpart, plan.n_sb, plan.Q,
                       vstrips ? plan.Pv : plan.P, vstrips ? plan.U : 64u, n_samples,
                       std::ldexp(1.0, -F), d_const_sum, d_part0, overwrite, d_tally, n_tally, d_tally1, n_tally1, d_timeout,
                       d_status, d_keep, n_keep);
    return hipGetLastError();
}

}  // namespace nps
