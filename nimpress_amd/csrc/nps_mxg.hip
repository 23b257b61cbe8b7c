// nps_mxg.hip -- the strip kernel for runs whose row tallies are GIVEN (round 5): a cohort that carries its tallies
// (nps_cohort_keep_tallies -- what NPS_MODE_AUTO arranges for cohort sizes whose resident grid does not cover the chip, and
// what many score files over one cohort want), or the second read of NPS_MODE_TWOPASS.
//
// With the tallies known beforehand nothing of nps_mx.hip's machinery is needed: no popcounts, no hand-over between the
// strips, no control waves, no cooperative launch.  The loop of nimpress.nim:561-583, 639-641 splits into
//
//   mx_ops_kernel    one thread per score row: the decisions of getImputedDosages from the row's whole-cohort tally
//                    (nimpress.nim:565-571), the locus constant or the sample imputation value (:417-481), the row's
//                    statistics, nloci -- and the row's three weight operands (beta digits; is-missing weights for the even
//                    and the odd sample of a pair) as 3 x 16 bytes in the layout LDS wants, 6 KiB per superblock;
//   mx_given_kernel  an ordinary grid of P strips x Q row teams; 8 identical waves x 8 units; per superblock: the 6 KiB of
//                    operands arrive by LDS-DMA one step ahead, every unit goes registers -> a wave-private 1 KiB LDS image
//                    -> ds_read_b64_tr_b4 -> four MFMAs (code x Wc, is_missing x Wm, even / odd sample).  ~23 vector
//                    instructions per unit instead of ~50: the pass is bound by the stream alone.  TWO register banks of
//                    loads in flight (128 KiB per compute unit).
//
// Layout, operand encodings, the digit sums handed to mx_fold_kernel: nps_mx.hip's.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "nps_kernels.h"
#include "nps_mx_common.h"

namespace nps {

#ifdef NPS_MX_TIMERS
__device__ unsigned long long g_mxg_timers[8][8];
#define GXT(i) do { if (timing) { const unsigned long long now_ = __builtin_readcyclecounter(); tph[i] += now_ - tlast; tlast = now_; } } while (0)
#else
#define GXT(i) do { } while (0)
#endif
constexpr int kGW = 8;                                   // waves of a workgroup, all alike
constexpr int kGU = 8;                                   // units per wave: 64 in all
constexpr uint32_t kGTransUnits = 2;                     // units of a wave on their way through LDS at a time (all eight at once,
                                                         // writes first, then the transposed reads: 21.6-21.8 ms against 20.3-20.8)
constexpr uint32_t kGTrans = 0;                          // [kGW][kGTransUnits][1 KiB]: the code image on its way to the transposed read
constexpr uint32_t kGTab = kGW * kGTransUnits * 1024u;   // [2][3 operands][128 rows][16 bytes]
constexpr uint32_t kGLdsBytes = kGTab + 2 * 6144u;
static_assert(kGW * kGU == 64, "units of a strip");

struct MxgArgs {
    const v4u *units;        // the cohort
    uint64_t n_sb_cohort;    // its superblocks
    uint32_t sb0, n_sb;      // this run: first superblock, superblocks
    uint32_t P, nu_last;     // strips, units of the last one
    uint32_t Q;              // row teams per strip: superblock k of the run belongs to team k % Q
    const v4u *ops;          // [n_sb][3][128] x 16 bytes from mx_ops_kernel; the fourth dword of a row = its superblock + 1
    float *cpart;            // [n_flush][Q][P][64][2][256]
    unsigned int *timeout;   // raised when a wave gives up waiting for its operand tables (mx_fold_kernel: sticky status)
};

// ---- per row: decisions, statistics, operands --------------------------------------------------------------------
// blocks of 128 threads = one superblock each; const_part: one double per block (summed in block order by the last block
// to finish: bit-reproducible), const_sum[0] receives the total
__global__ __launch_bounds__(128) void mx_ops_kernel(const nps_row_desc *__restrict__ desc, uint64_t n_rows,
                                                     const unsigned long long *__restrict__ tally, MxArgs a,
                                                     v4u *__restrict__ ops, double *__restrict__ const_part,
                                                     unsigned int *__restrict__ done_count) {
    const uint32_t sb = blockIdx.x, r = threadIdx.x;
    const uint64_t row = (uint64_t)sb * 128 + r;
    uint32_t wc[3] = {0u, 0u, 0u}, wme[3] = {0u, 0u, 0u}, wmo[3] = {0u, 0u, 0u};
    int used = 0;
    double cst = 0.0;
    if (row < n_rows) {
        const double beta = desc[row].beta, eaf = desc[row].eaf;
        const bool rie = desc[row].ref_is_effect != 0;
        const double inf = __builtin_huge_val(), nan = __longlong_as_double(0x7ff8000000000000ll);
        MxPre o;  // (what mx_prep_kernel makes for the other two kernels)
        o.c[0] = o.c[1] = o.c[2] = 0u;
        o.flags = 0u;
        o.w1 = o.wfb = 0;
        if (!(fabs(beta) < inf)) {
            o.flags = 1u;
        } else {
            o.w1 = __double2ll_rn(beta * a.scale);
            mx_codes(o.w1, 0u, o.c);
            const double imp = a.prm.imp_sample == NPS_SAMPLE_HOMREF ? (rie ? 2.0 : 0.0)
                               : (a.prm.imp_sample == NPS_SAMPLE_FAIL || a.prm.imp_sample == NPS_SAMPLE_INT_FAIL) ? nan
                                                                                                                  : eaf * 2.0;
            const double t = imp * beta;
            if (!(fabs(t) < inf)) {
                o.flags = 2u;
                o.wfb = 3 * o.w1;
            } else {
                o.wfb = __double2ll_rn(t * a.scale);
            }
        }
        MxPreX px;
        px.w1d = (double)o.w1;
        px.w3 = 3 * o.w1;
        px.w4 = 4 * o.w1;
        mx_row(a, tally[row], true, row, o, px, a.stats != nullptr, wc, wme, wmo, used, cst);
    }
    const uint32_t tag = sb + 1u;  // (the consumer's "has my DMA landed" mark: never zero, different in both table buffers)
    v4u *dst = ops + (uint64_t)sb * 384 + r;
    dst[0] = v4u{wc[0], wc[1], wc[2], tag};
    dst[128] = v4u{wme[0], wme[1], wme[2], tag};
    dst[256] = v4u{wmo[0], wmo[1], wmo[2], tag};
    // nloci and the locus constants of the block's rows, in lane order
    __shared__ double s_c[128];
    __shared__ unsigned int s_used[2];
    __shared__ bool s_last;
    s_c[r] = cst;
    const unsigned long long b = __ballot(used != 0);
    if ((r & 63u) == 0) s_used[r >> 6] = (unsigned int)__popcll(b);
    __syncthreads();
    if (r == 0) {
        double c = 0.0;
        for (int i = 0; i < 128; ++i) c += s_c[i];
        const_part[sb] = c;
        const unsigned int u = s_used[0] + s_used[1];
        if (u) atomicAdd(a.nloci, (unsigned long long)u);
        __threadfence();
        s_last = atomicAdd(done_count, 1u) + 1u == gridDim.x;
    }
    __syncthreads();
    if (s_last && r == 0) {  // every block's partial is visible: add them in block order
        __threadfence();
        double c = 0.0;
        for (uint32_t i = 0; i < gridDim.x; ++i) c += __hip_atomic_load(&const_part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.const_sum[0] = c;
        *done_count = 0u;
    }
}

// ---- the pass ----------------------------------------------------------------------------------------------------
static __device__ __forceinline__ void mxg_dma16(const void *gsrc, uint32_t lds_dst_any) {
    unsigned keep;
    const uint32_t lds_dst = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_dst_any);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

#ifndef NPS_MXG_BANKS
#define NPS_MXG_BANKS 2
#endif

template <bool GUARD>
static __device__ __forceinline__ void mxg_body(const MxgArgs &a, char *const smem) {
    constexpr int NU = kGU;
    constexpr int kBanks = NPS_MXG_BANKS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int g = lane >> 4, q = lane & 15;
    const uint32_t strip = blockIdx.x % a.P, team = blockIdx.x / a.P;
    const uint32_t n_t = team < a.n_sb ? (a.n_sb - team + a.Q - 1) / a.Q : 0u;  // superblocks of this team
    if (n_t == 0) return;
    const int nu = strip == a.P - 1 ? (int)a.nu_last : 64;
    const int u0 = wave * NU;
    const int n_my = GUARD ? max(0, min(NU, nu - u0)) : NU;  // wave-uniform
    constexpr bool full = !GUARD;

    char *const trans = smem + kGTrans + (size_t)wave * (kGTransUnits * 1024u);
    const int woff = mx_rowoff(2 * lane);
    const int r1off = mx_rowoff(32 * g + q), r2off = mx_rowoff(32 * g + 16 + q);
    const int fr0 = (32 * g + q) * 16, fr1 = (32 * g + 16 + q) * 16;
    const v4u *const base =
        a.units + ((uint64_t)strip * 64 * a.n_sb_cohort + (uint64_t)(a.sb0 + team) * nu + u0) * 64 + lane;
    const uint64_t sb_stride = (uint64_t)nu * 64 * a.Q;

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
    // the operands of superblock k of the team -> table buffer k & 1: waves 0..5 fetch 1 KiB each (64 rows x 16 bytes of
    // one of the three operands); the fourth dword of every row carries the superblock's number + 1
    const v4u *const ops_mine = a.ops + (uint64_t)wave * 64 + lane;
    const uint32_t lds_tab = (uint32_t)(uintptr_t)(smem + kGTab) + (uint32_t)wave * 1024u;
    auto dma_tables = [&](uint32_t k) {
        if (wave < 6 && k < n_t) mxg_dma16(ops_mine + (uint64_t)(team + k * a.Q) * 384, lds_tab + (k & 1u) * 6144u);
    };
    auto tables_landed = [&](uint32_t k) {  // this wave's piece of superblock k's operands is in LDS (nothing is counted)
        if (wave >= 6 || k >= n_t) return;
        const uint32_t want = team + k * a.Q + 1u;
        const char *p = smem + kGTab + (k & 1u) * 6144u + wave * 1024 + lane * 16 + 12;
        bool landed = false;
        for (uint32_t spins = 0; spins < (1u << 22); ++spins) {
            asm volatile("" ::: "memory");
            const uint32_t tag = *reinterpret_cast<const uint32_t *>(p);
            if (__all(tag == want)) {
                landed = true;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        // a wave that gives up would multiply by stale or partial operands: say so (the pass then fails with NPS_E_TIMEOUT
        // instead of returning wrong scores; nps_mx.hip's bounded waits do the same)
        if (!landed && lane == 0) __hip_atomic_store(a.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

    // scores += code * Wc + is_missing * Wm for the wave's units of superblock k (in `b`); its registers are refilled
    auto work = [&](uint32_t k, v4u(&b)[NU]) {
        if (k >= n_t || n_my == 0) return;
        const char *tab = smem + kGTab + (k & 1u) * 6144u;
        v8i Bc, Bme, Bmo;
        {
            const v3i c0 = tr6(tab + fr0), c1 = tr6(tab + fr1);
            const v3i e0 = tr6(tab + 2048 + fr0), e1 = tr6(tab + 2048 + fr1);
            const v3i o0 = tr6(tab + 4096 + fr0), o1 = tr6(tab + 4096 + fr1);
            Bc = v8i{c0[0], c0[1], c0[2], c1[0], c1[1], c1[2], 0, 0};
            Bme = v8i{e0[0], e0[1], e0[2], e1[0], e1[1], e1[2], 0, 0};
            Bmo = v8i{o0[0], o0[1], o0[2], o1[0], o1[1], o1[2], 0, 0};
        }
        auto one = [&](int u, bool refill) {
            char *const tp = trans + (u % (int)kGTransUnits) * 1024;
            *reinterpret_cast<v4u *>(tp + woff) = b[u];
            if (refill) b[u] = load_unit(k + kBanks, u);
            const v2i t01 = tr4(tp + r1off), t23 = tr4(tp + r2off);
            const uint32_t w[4] = {(uint32_t)t01[0], (uint32_t)t01[1], (uint32_t)t23[0], (uint32_t)t23[1]};
            v8i ce = {0, 0, 0, 0, 0, 0, 0, 0}, co = ce, me = ce, mo = ce;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const uint32_t s1 = w[r] >> 1;
                ce[r] = (int)(w[r] & 0x33333333u);       // even sample: 00hl = code / 2
                co[r] = (int)(s1 & 0x66666666u);         // odd sample: 0hl0 = 0, 1, 2, 4
                me[r] = (int)(w[r] & s1 & 0x11111111u);  // even sample missing: 0001 = 1/2
                mo[r] = (int)(w[r] & s1 & 0x44444444u);  // odd sample missing: 0100 = 2
            }
            C[u][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ce, Bc, C[u][0], 4, 2, 0, 128, 0, 130);
            C[u][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(co, Bc, C[u][1], 4, 2, 0, 127, 0, 130);
            C[u][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(me, Bme, C[u][0], 4, 2, 0, 128, 0, 130);
            C[u][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(mo, Bmo, C[u][1], 4, 2, 0, 126, 0, 130);
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
    };

    // ---- prologue
    // LDS keeps what the compute unit's previous workgroup left in it -- possibly this kernel's tables of an EARLIER launch,
    // whose rows carry the very superblock numbers this launch waits for (another score file's weights: seen once in round 5
    // as 125 samples of one strip off by 5e-4 relative).  No row of either buffer may carry a number before the first DMA
    // is issued: zero is nobody's (the numbers start at 1), and the stores are done before the DMA leaves.
    if (wave < 6) {
        *reinterpret_cast<uint32_t *>(smem + kGTab + wave * 1024 + lane * 16 + 12) = 0u;
        *reinterpret_cast<uint32_t *>(smem + kGTab + 6144u + wave * 1024 + lane * 16 + 12) = 0u;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    dma_tables(0);
#pragma unroll
    for (int b = 0; b < kBanks; ++b) load_sb((uint32_t)b, bank[b]);
    tables_landed(0);
    __syncthreads();
#ifdef NPS_MX_TIMERS
    const bool timing = strip == a.P / 2 && team == 0;
    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#endif
    auto step = [&](uint32_t k, v4u(&b)[NU]) {
        GXT(4);
        dma_tables(k + 1);  // (buffer (k + 1) & 1: every wave passed the barrier behind its last reads of superblock k - 1)
        GXT(0);
        work(k, b);
        GXT(1);
        if (((k + 1) & (kFlushSb - 1)) == 0 && k + 1 < n_t) store_c(k / kFlushSb, true);
        tables_landed(k + 1);
        GXT(2);
        __syncthreads();
        GXT(3);
    };
    // (the banks are register arrays: the loop is unrolled by their number so that every step names its bank)
    const uint32_t n_steps = (n_t + kBanks - 1) / kBanks * kBanks;
    for (uint32_t k = 0; k < n_steps; k += kBanks) {
#pragma unroll
        for (int b = 0; b < kBanks; ++b) step(k + (uint32_t)b, bank[b]);
    }
    store_c((n_t - 1) / kFlushSb, false);
#ifdef NPS_MX_TIMERS
    if (timing && lane == 0) {
        for (int i = 0; i < 8; ++i) g_mxg_timers[wave][i] = tph[i];
        g_mxg_timers[wave][7] = n_t;
    }
#endif
}

__global__ __launch_bounds__(kGW * 64, kGW / 4) void mx_given_kernel(const MxgArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6;
    const int nu = blockIdx.x % a.P == a.P - 1 ? (int)a.nu_last : 64;
    if (nu - wave * kGU >= kGU)
        mxg_body<false>(a, smem);
    else
        mxg_body<true>(a, smem);
}

// d_ops: 48 bytes per row of the run padded to 128; d_const_part: one double per superblock; d_done: one zeroed word;
// d_const_sum[0] receives the locus constants (the other 2 Q - 1 slots stay zero), as mx_fold_kernel expects them
hipError_t launch_mx_given(hipStream_t st, const MxPlan &plan, const void *d_units, uint64_t n_sb_cohort, uint64_t sb0,
                           uint64_t n_samples, uint64_t n_rows, const nps_row_desc *d_desc, DevParams prm,
                           int64_t t_maxmis, int F, const unsigned long long *d_tally, nps_locus_stat *d_stats,
                           unsigned long long *d_nloci, double *d_const_sum, float *d_cpart, void *d_ops,
                           double *d_const_part, unsigned int *d_done, unsigned int *d_timeout) {
    if (!plan.ok || !plan.given) return hipErrorInvalidValue;
    (void)hipGetLastError();
    MxArgs ra;  // (what mx_row reads)
    ra.units = nullptr;
    ra.n_sb_cohort = 0;
    ra.sb0 = 0;
    ra.n_sb = plan.n_sb;
    ra.n_rows = n_rows;
    ra.n_samples = n_samples;
    ra.P = plan.P;
    ra.nu_last = plan.nu_last;
    ra.Q = plan.Q;
    ra.desc = d_desc;
    ra.pre = nullptr;
    ra.prm = prm;
    ra.t_maxmis = t_maxmis;
    ra.scale = std::ldexp(1.0, F);
    ra.tally = nullptr;
    ra.tally1 = nullptr;
    ra.stats = d_stats;
    ra.nloci = d_nloci;
    ra.const_sum = d_const_sum;
    ra.cpart = nullptr;
    ra.timeout = nullptr;
    ra.ctl_prio = 0;
    ra.grp_strips = 0;
    hipLaunchKernelGGL(mx_ops_kernel, dim3(plan.n_sb), dim3(128), 0, st, d_desc, n_rows, d_tally, ra, (v4u *)d_ops,
                       d_const_part, d_done);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    static bool attr_set = false;
    if (!attr_set) {
        e = hipFuncSetAttribute((const void *)mx_given_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kGLdsBytes);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    MxgArgs a;
    a.units = (const v4u *)d_units;
    a.n_sb_cohort = n_sb_cohort;
    a.sb0 = (uint32_t)sb0;
    a.n_sb = plan.n_sb;
    a.P = plan.P;
    a.nu_last = plan.nu_last;
    a.Q = plan.Q;
    a.ops = (const v4u *)d_ops;
    a.cpart = d_cpart;
    a.timeout = d_timeout;
    hipLaunchKernelGGL(mx_given_kernel, dim3(plan.P * plan.Q), dim3(kGW * 64), kGLdsBytes, st, a);
#ifdef NPS_MX_TIMERS
    {
        (void)hipStreamSynchronize(st);
        unsigned long long h[8][8];
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_mxg_timers), sizeof(h));
        static const char *nm[5] = {"dma-issue", "work", "landing", "barrier", "loop"};
        for (int w = 0; w < kGW; ++w) {
            fprintf(stderr, "mxg timers wave %d (cycles per step, %llu steps):", w, h[w][7]);
            for (int i = 0; i < 5; ++i) fprintf(stderr, "  %s %.0f", nm[i], (double)h[w][i] / (double)std::max<unsigned long long>(h[w][7], 1));
            fprintf(stderr, "\n");
        }
    }
#endif
    return hipGetLastError();
}

}  // namespace nps
