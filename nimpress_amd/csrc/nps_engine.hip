// nps_engine.hip -- implementation of the C-ABI declared in include/nps.h.
//
// Host-side orchestration only: staging, batching, launches, bookkeeping of per-row stats.  All
// per-sample arithmetic happens in nps_kernels.hip (and nps_fused.hip).  There is no CPU compute
// path here: if HIP is unusable every entry point that would compute returns NPS_E_NODEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>
#include <string>
#include <mutex>
#include <vector>

#include "nps_kernels.h"

using namespace nps;

// ------------------------------------------------------------------------------------------
// errors
static thread_local std::string g_last_error;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return fail(_e == hipErrorOutOfMemory ? NPS_E_NOMEM : NPS_E_HIP, "%s failed: %s",  \
                        #expr, hipGetErrorString(_e));                                         \
    } while (0)

// ------------------------------------------------------------------------------------------
enum ProfClass { P_DECODE = 0, P_TALLY, P_PARAMS, P_ACCUM, P_FUSED, P_REDUCE, P_COUNT };

struct ProfSpan {
    hipEvent_t a, b;
    int cls;
};

struct nps_cohort {
    int device = 0;
    int format = NPS_FMT_GT2;
    uint64_t n_samples = 0, n_rows = 0;
    uint64_t stride_bytes = 0;  // per row (GT2: rows are interleaved in groups of 4, a group is 4*stride_bytes)
    void *d_data = nullptr;
    // nps_cohort_optimize (GT2): the rows are in the parity layout (nps_kernels.h: the high-bit plane of
    // slot 0 of every group holds the XOR of the four rows' planes)
    bool optimized = false;
    // NPS_FMT_GT2M: whole-row tallies (nmissing << 32 | neffect) produced by whatever packed the rows
    unsigned long long *d_row_tally = nullptr;
    // NPS_FMT_GT2X after nps_cohort_keep_tallies: the whole-row tallies (nmissing << 28 | neffect, the tally word of the
    // strip kernel without its arrival count), one per row of every superblock; valid until rows are rewritten
    unsigned long long *d_mx_row_tally = nullptr;
    // (atomic: contexts on several threads may score one cohort; whoever finds the flag clear takes tally_mutex, the writer
    //  publishes with release order AFTER the counted tallies are in device memory, readers load with acquire)
    std::atomic<bool> mx_row_tally_valid{false};
    std::mutex tally_mutex;  // NPS_MODE_AUTO may count them lazily from whichever context scores the cohort first
    std::atomic<uint32_t> expect_passes{0};  // nps_cohort_expect_passes: how often the caller will score this cohort (0: not said)
    // nps_cohort_push_*: rows decoded on the device straight into the cohort (a pinned ring the decode kernel reads
    // over PCIe, on a stream of the cohort's own); every call that reads the cohort waits for it (cohort_quiesce)
    hipStream_t push_stream = nullptr;
    void *h_push[2] = {nullptr, nullptr};
    hipEvent_t ev_push[2] = {nullptr, nullptr};
    size_t push_cap = 0;
    int push_next = 0;
    unsigned long long *d_push_tally = nullptr;  // scratch word for the decode kernel's tally (a GT2 cohort keeps none)
    // NPS_FMT_DS32: rows that hold a value outside [0, 2] (checked when rows are uploaded; the generator clips): while
    // there is one, the cohort is scored by the two-pass kernels (the single-read kernel's fixed-point tallies need the range)
    std::vector<unsigned char> ds_row_bad;
    uint64_t ds_bad_rows = 0;
};

static int cohort_quiesce(const nps_cohort *c) {
    if (c && c->push_stream) HIP_TRY(hipStreamSynchronize(c->push_stream));
    return NPS_OK;
}

struct PendingRow {
    int32_t batch_idx;  // >= 0: index into the open GT / DS batch's device stats; -1: host stat
    int32_t is_ds;      // which open batch batch_idx refers to
    nps_locus_stat host;
};

struct nps_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    uint64_t n = 0;         // samples
    uint64_t n_words = 0;   // ceil(n/16)
    uint64_t stride_words = 0;
    nps_params params{};

    // streaming batch
    uint32_t batch_cap = 0, batch_rows = 0;
    uint32_t *d_codes = nullptr;            // [batch_cap/4 groups][stride_words][4] (interleaved)
    unsigned long long *d_tally = nullptr;  // [batch_cap]
    nps_row_desc *d_desc = nullptr;         // [batch_cap]
    nps_row_desc *h_desc = nullptr;         // pinned [batch_cap]
    double *d_lut = nullptr;                // [batch_cap][4]
    nps_locus_stat *d_stats = nullptr;      // [batch_cap]
    nps_locus_stat *h_stats = nullptr;      // pinned [batch_cap]
    static constexpr int kRawSlots = 8;  // rows in flight between the caller's buffer and the device
    int32_t *h_raw[kRawSlots] = {};  // pinned staging ring for caller buffers
    void *h_arena = nullptr;  // ONE pinned allocation holding h_desc, h_stats and h_raw[]
    hipEvent_t ev_raw[kRawSlots] = {};
    int raw_next = 0;

    // FORMAT/DS streaming batch (allocated on the first nps_push_ds)
    uint32_t ds_cap = 0, ds_rows = 0;
    uint64_t ds_stride_f = 0;
    float *d_ds = nullptr;               // [ds_cap][ds_stride_f]
    nps_row_desc *d_ds_desc = nullptr;   // [ds_cap]
    DsTally *d_ds_tally = nullptr;
    DsRowP *d_ds_rowp = nullptr;
    nps_locus_stat *d_ds_stats = nullptr;
    void *h_ds_arena = nullptr;          // pinned: h_ds_desc, h_ds_stats, staging ring
    nps_row_desc *h_ds_desc = nullptr;
    nps_locus_stat *h_ds_stats = nullptr;
    float *h_ds_raw[2] = {nullptr, nullptr};
    hipEvent_t ev_ds_raw[2] = {nullptr, nullptr};
    int ds_raw_next = 0;
    // resident DS runs
    uint64_t res_ds_cap = 0;
    DsTally *d_rds_tally = nullptr;
    DsRowP *d_rds_rowp = nullptr;

    // raw GT staging for ploidy > 2: a ring of two pinned host buffers (grown on demand), read by the kernel
    size_t poly_cap = 0;
    void *h_poly[2] = {nullptr, nullptr};
    hipEvent_t ev_poly[2] = {nullptr, nullptr};
    int poly_next = 0;

    // accumulators
    AccumGeom geom{};         // streaming geometry (groups_per_chunk for a full batch)
    uint32_t n_chunks = 1;
    double *d_part = nullptr;  // [n_chunks][part_chunk_stride]
    // which chunks of d_part hold data: 0 = none yet (nothing has been zeroed either: the first
    // writer overwrites), 1 = chunk 0 only (fused epilogues), n_chunks = all (two-pass kernels)
    uint32_t chunks_used = 0;
    double *d_scores = nullptr;
    unsigned long long *d_nloci = nullptr;   // [0] used rows counted on the device, [1] sticky status bits
    unsigned long long *h_result = nullptr;  // pinned copy of that block
    bool broken = false;                     // a launch sequence failed half-way: nps_reset first
    double const_sum = 0.0;   // contributions of rows without genotype data (host decided)
    uint64_t host_nloci = 0;

    // resident runs (nps_score_cohort*): per-row buffers, grown on demand
    uint64_t res_cap = 0;
    unsigned long long *d_rtally = nullptr;
    double *d_rlut = nullptr;
    nps_locus_stat *d_rstats = nullptr;
    double *d_part_fused = nullptr;         // [Q][team stride] partial scores of the fused kernel
    uint64_t part_fused_cap = 0;            // doubles
    unsigned int *d_timeout = nullptr;      // bounded-wait flag of the fused kernels (cleared by fold_kernel)
    float *d_mx_cpart = nullptr;            // NPS_FMT_GT2X runs: digit sums handed from fused_mx_kernel to mx_fold_kernel
    uint64_t mx_cpart_cap = 0;              // floats
    double *d_mx_const = nullptr;           // ... and the locus constants of rows over --maxmis: [8 scratch][2 Q slots], zero between passes
    uint64_t mx_const_cap = 0;
    unsigned long long *d_mx_tally1 = nullptr;  // first-stage tally words (groups of 16 strips), zero between passes
    uint64_t mx_tally1_cap = 0;
    void *d_mx_ops = nullptr;                   // nps_mxg.hip (tallies given): 48 bytes of weight operands per row ...
    uint64_t mx_ops_cap = 0;                    // (in units of 48 bytes)
    double *d_mx_cblk = nullptr;                // ... and one partial sum of locus constants per superblock
    uint64_t mx_cblk_cap = 0;
    bool mx_plan_valid = false, mx_plan_two_pass = false;
    uint64_t mx_plan_m = 0;
    MxPlan mx_plan_cache{};
    bool rtally_clean = false;              // d_rtally is all zero (allocation, or the last fused epilogue)
    // shape -> persistent-grid plan of the last resident run (occupancy queries are slow)
    bool plan_valid = false;
    int plan_fmt = -1;
    uint64_t plan_m = 0;
    FusedPlan plan_cache{};
    bool res_pending = false;               // stats of the last resident run still on the device
    uint64_t res_m = 0;
    std::vector<int64_t> res_index;
    std::vector<nps_locus_stat> res_host_stats;

    // stats in push order
    std::vector<PendingRow> pending;        // rows of the open batch (+ no-data rows since)
    std::vector<nps_locus_stat> ready;      // completed, not yet returned by nps_flush
    size_t ready_cursor = 0;

    // profiling
    bool profiling = false;
    std::vector<ProfSpan> spans;
    nps_profile prof{};
};

static DevParams dev_params(const nps_params &p) {
    DevParams d;
    d.imp_locus = p.imp_locus;
    d.imp_missing = p.imp_missing;
    d.imp_sample = p.imp_sample;
    d.max_missing_rate = p.max_missing_rate;
    d.min_cs = (double)p.min_cs;
    return d;
}

static int check_params(const nps_params *p) {
    if (!p) return fail(NPS_E_INVAL, "params is NULL");
    if (p->imp_locus < 0 || p->imp_locus > NPS_LOCUS_IGNORE)
        return fail(NPS_E_INVAL, "imp_locus %d out of range", p->imp_locus);
    if (p->imp_missing < 0 || p->imp_missing > NPS_MISSING_IGNORE)
        return fail(NPS_E_INVAL, "imp_missing %d out of range", p->imp_missing);
    if (p->imp_sample < 0 || p->imp_sample > NPS_SAMPLE_INT_FAIL)
        return fail(NPS_E_INVAL, "imp_sample %d out of range", p->imp_sample);
    return NPS_OK;
}

// profiling helpers -------------------------------------------------------------------------
struct ProfScope {
    nps_ctx *c;
    int cls;
    hipEvent_t a = nullptr, b = nullptr;
    ProfScope(nps_ctx *ctx, int k) : c(ctx), cls(k) {
        if (c->profiling) {
            if (hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess)
                (void)hipEventRecord(a, c->stream);
            else
                a = b = nullptr;
        }
    }
    ~ProfScope() {
        if (a && b) {
            (void)hipEventRecord(b, c->stream);
            c->spans.push_back({a, b, cls});
        }
    }
};

static void resolve_spans(nps_ctx *c) {
    for (auto &s : c->spans) {
        float ms = 0.f;
        (void)hipEventSynchronize(s.b);
        (void)hipEventElapsedTime(&ms, s.a, s.b);
        double *acc[P_COUNT] = {&c->prof.ms_decode, &c->prof.ms_tally,  &c->prof.ms_params,
                                &c->prof.ms_accumulate, &c->prof.ms_fused, &c->prof.ms_reduce};
        uint64_t *cnt[P_COUNT] = {&c->prof.n_decode, &c->prof.n_tally,  &c->prof.n_params,
                                  &c->prof.n_accumulate, &c->prof.n_fused, &c->prof.n_reduce};
        *acc[s.cls] += ms;
        *cnt[s.cls] += 1;
        (void)hipEventDestroy(s.a);
        (void)hipEventDestroy(s.b);
    }
    c->spans.clear();
}

// ------------------------------------------------------------------------------------------
extern "C" int nps_abi_version(void) { return NPS_ABI_VERSION; }
extern "C" const char *nps_last_error(void) { return g_last_error.c_str(); }

extern "C" int nps_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int select_device(int device);
__global__ void warmup_kernel(unsigned int *p) {
    if (p) *p = 1u;
}
extern "C" int nps_warmup(int device) {
    int rc = select_device(device);
    if (rc) return rc;
    HIP_TRY(hipFree(nullptr));  // the primary context
    (void)hipGetLastError();
    hipLaunchKernelGGL(warmup_kernel, dim3(1), dim3(1), 0, nullptr, (unsigned int *)nullptr);  // the code object
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipDeviceSynchronize());
    return NPS_OK;
}

static int select_device(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(NPS_E_NODEVICE, "no HIP device available (%s); libnps has no CPU path",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n)
        return fail(NPS_E_NODEVICE, "device %d out of range (have %d)", device, n);
    HIP_TRY(hipSetDevice(device));
    return NPS_OK;
}

// geometry of the partial-score buffer: sample tiles x row chunks >= ~2048 blocks
static void choose_geometry(nps_ctx *c) {
    const uint32_t tiles = (uint32_t)std::max<uint64_t>(1, (c->n_words + 255) / 256);
    uint32_t chunks = (2048 + tiles - 1) / tiles;
    chunks = std::max(1u, std::min(chunks, 64u));
    c->n_chunks = chunks;
    c->geom.n_words = (uint32_t)c->n_words;
    c->geom.n_chunks = chunks;
    c->geom.part_chunk_stride = (uint64_t)tiles * 256 * 16;
}

static void free_ctx(nps_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (auto &s : c->spans) {
        (void)hipEventDestroy(s.a);
        (void)hipEventDestroy(s.b);
    }
    (void)hipFree(c->d_codes);
    (void)hipFree(c->d_tally);
    (void)hipFree(c->d_desc);
    (void)hipHostFree(c->h_arena);
    (void)hipFree(c->d_lut);
    (void)hipFree(c->d_stats);
    for (int k = 0; k < nps_ctx::kRawSlots; ++k)
        if (c->ev_raw[k]) (void)hipEventDestroy(c->ev_raw[k]);
    (void)hipFree(c->d_ds);
    for (int k = 0; k < 2; ++k) {
        (void)hipHostFree(c->h_poly[k]);
        if (c->ev_poly[k]) (void)hipEventDestroy(c->ev_poly[k]);
    }
    (void)hipFree(c->d_ds_desc);
    (void)hipFree(c->d_ds_tally);
    (void)hipFree(c->d_ds_rowp);
    (void)hipFree(c->d_ds_stats);
    (void)hipHostFree(c->h_ds_arena);
    for (int k = 0; k < 2; ++k)
        if (c->ev_ds_raw[k]) (void)hipEventDestroy(c->ev_ds_raw[k]);
    (void)hipFree(c->d_rds_tally);
    (void)hipFree(c->d_rds_rowp);
    (void)hipFree(c->d_part_fused);
    (void)hipFree(c->d_timeout);
    (void)hipFree(c->d_mx_cpart);
    (void)hipFree(c->d_mx_const);
    (void)hipFree(c->d_mx_tally1);
    (void)hipFree(c->d_mx_ops);
    (void)hipFree(c->d_mx_cblk);
    (void)hipFree(c->d_rtally);
    (void)hipFree(c->d_rlut);
    (void)hipFree(c->d_rstats);
    (void)hipFree(c->d_part);
    (void)hipFree(c->d_scores);
    (void)hipFree(c->d_nloci);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static int zero_state(nps_ctx *c) {
    // d_part is not touched: chunks_used = 0 makes the first writer overwrite it
    HIP_TRY(hipMemsetAsync(c->d_nloci, 0, 3 * sizeof(unsigned long long), c->stream));
    if (c->batch_rows)  // tallies of rows decoded into the open batch (the array is zero otherwise)
        HIP_TRY(hipMemsetAsync(c->d_tally, 0, sizeof(unsigned long long) * c->batch_rows, c->stream));
    c->chunks_used = 0;
    c->broken = false;
    c->const_sum = 0.0;
    c->host_nloci = 0;
    c->batch_rows = 0;
    c->ds_rows = 0;
    c->pending.clear();
    c->ready.clear();
    c->ready_cursor = 0;
    c->res_pending = false;  // unflushed stats of a resident run are dropped, never copied
    c->res_index.clear();
    c->res_host_stats.clear();
    return NPS_OK;
}

// the two-pass kernels add into every chunk of d_part: zero the chunks nothing has written yet
static int ensure_all_chunks(nps_ctx *c) {
    if (c->chunks_used < c->n_chunks) {
        HIP_TRY(hipMemsetAsync(c->d_part + (uint64_t)c->chunks_used * c->geom.part_chunk_stride, 0,
                               sizeof(double) * (c->n_chunks - c->chunks_used) * c->geom.part_chunk_stride,
                               c->stream));
        c->chunks_used = c->n_chunks;
    }
    return NPS_OK;
}

extern "C" int nps_create(nps_ctx **out, int device, uint64_t n_samples, const nps_params *params) {
    if (!out) return fail(NPS_E_INVAL, "out is NULL");
    *out = nullptr;
    int rc = check_params(params);
    if (rc) return rc;
    if (n_samples > 0x7fffffffull)
        return fail(NPS_E_UNSUPPORTED, "n_samples %llu exceeds 2^31-1",
                    (unsigned long long)n_samples);
    rc = select_device(device);
    if (rc) return rc;
    nps_ctx *c = new (std::nothrow) nps_ctx;
    if (!c) return fail(NPS_E_NOMEM, "out of host memory");
    c->device = device;
    c->n = n_samples;
    c->n_words = words_for(n_samples);
    c->stride_words = stride_words_for(n_samples);
    c->params = *params;
    choose_geometry(c);

    const uint64_t row_bytes = c->stride_words * 4;
    uint64_t cap = (64ull << 20) / row_bytes;
    cap = std::max<uint64_t>(16, std::min<uint64_t>(cap, 4096));
    cap = cap / 16 * 16;
    c->batch_cap = (uint32_t)cap;
    c->geom.groups_per_chunk =
        std::max(1u, ((c->batch_cap / 4) + c->n_chunks - 1) / c->n_chunks);

#define CTX_TRY(expr)                                                                    \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            int code = fail(_e == hipErrorOutOfMemory ? NPS_E_NOMEM : NPS_E_HIP,         \
                            "%s failed: %s", #expr, hipGetErrorString(_e));              \
            free_ctx(c);                                                                 \
            return code;                                                                 \
        }                                                                                \
    } while (0)
    CTX_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    CTX_TRY(hipMalloc(&c->d_codes, row_bytes * c->batch_cap));
    CTX_TRY(hipMemsetAsync(c->d_codes, 0, row_bytes * c->batch_cap, c->stream));
    CTX_TRY(hipMalloc(&c->d_tally, sizeof(unsigned long long) * c->batch_cap));
    CTX_TRY(hipMalloc(&c->d_desc, sizeof(nps_row_desc) * c->batch_cap));
    {
        // One pinned arena for all host staging, carved at 4 KiB boundaries, padded to 64 KiB.
        auto up = [](size_t v) { return (v + 4095) / 4096 * 4096; };
        const size_t sz_desc = up(sizeof(nps_row_desc) * c->batch_cap);
        const size_t sz_stats = up(sizeof(nps_locus_stat) * c->batch_cap);
        const size_t sz_raw = up(sizeof(int32_t) * 2 * std::max<uint64_t>(c->n, 1));
        size_t total = 4096 + sz_desc + sz_stats + nps_ctx::kRawSlots * sz_raw;
        total = (total + 65535) / 65536 * 65536;
        CTX_TRY(hipHostMalloc(&c->h_arena, total));
        char *p = (char *)c->h_arena;
        c->h_result = (unsigned long long *)p;
        c->h_result[0] = c->h_result[1] = c->h_result[2] = 0;
        p += 4096;
        c->h_desc = (nps_row_desc *)p;
        p += sz_desc;
        c->h_stats = (nps_locus_stat *)p;
        p += sz_stats;
        for (int k = 0; k < nps_ctx::kRawSlots; ++k) {
            c->h_raw[k] = (int32_t *)p;
            p += sz_raw;
        }
    }
    CTX_TRY(hipMalloc(&c->d_lut, sizeof(double) * 4 * c->batch_cap));
    CTX_TRY(hipMalloc(&c->d_stats, sizeof(nps_locus_stat) * c->batch_cap));
    for (int k = 0; k < nps_ctx::kRawSlots; ++k) {
        CTX_TRY(hipEventCreateWithFlags(&c->ev_raw[k], hipEventDisableTiming));
    }
    CTX_TRY(hipMalloc(&c->d_part, sizeof(double) * c->n_chunks * c->geom.part_chunk_stride));
    CTX_TRY(hipMalloc(&c->d_scores, sizeof(double) * std::max<uint64_t>(c->n, 1)));
    CTX_TRY(hipMalloc(&c->d_nloci, 256));
    CTX_TRY(hipMalloc(&c->d_timeout, 256));
    CTX_TRY(hipMemsetAsync(c->d_timeout, 0, 256, c->stream));
    CTX_TRY(hipMemsetAsync(c->d_tally, 0, sizeof(unsigned long long) * c->batch_cap, c->stream));
#undef CTX_TRY
    rc = zero_state(c);
    if (rc) {
        free_ctx(c);
        return rc;
    }
    *out = c;
    return NPS_OK;
}

extern "C" void nps_destroy(nps_ctx *ctx) { free_ctx(ctx); }
extern "C" uint64_t nps_n_samples(const nps_ctx *ctx) { return ctx ? ctx->n : 0; }
extern "C" int nps_device(const nps_ctx *ctx) { return ctx ? ctx->device : -1; }

extern "C" int nps_reset(nps_ctx *c, const nps_params *params) {
    if (!c) return fail(NPS_E_INVAL, "ctx is NULL");
    if (params) {
        int rc = check_params(params);
        if (rc) return rc;
        c->params = *params;
    }
    HIP_TRY(hipSetDevice(c->device));
    // no stream synchronisation: everything below is ordered on the context's stream, and no call
    // returns with a device-to-host copy still in flight
    return zero_state(c);
}

// ------------------------------------------------------------------------------------------
// rows without genotype data are decided entirely on the host: nimpress.nim:526-558 + 417-447
static void host_locus_row(nps_ctx *c, int kind, int rie, double beta, double eaf,
                           nps_locus_stat *st) {
    st->ngenotyped = 0;
    st->nmissing = 0;
    st->neffect = 0.0;
    int used;
    double dosage = 0.0;
    if (kind == NPS_ROW_ABSENT) {  // :536-551
        st->reason = NPS_REASON_ABSENT;
        if (c->params.imp_missing == NPS_MISSING_HOMREF) {
            used = 1;
            dosage = rie ? 2.0 : 0.0;
        } else {
            used = 0;
        }
    } else {  // UNCOVERED :526-531, FILTERED :553-558 -> imputeLocusDosages :417-447
        st->reason = kind == NPS_ROW_UNCOVERED ? NPS_REASON_UNCOVERED : NPS_REASON_FILTERED;
        switch (c->params.imp_locus) {
        case NPS_LOCUS_IGNORE: used = 0; break;
        case NPS_LOCUS_PS: used = 1; dosage = eaf * 2.0; break;
        case NPS_LOCUS_HOMREF: used = 1; dosage = rie ? 2.0 : 0.0; break;
        default: used = 1; dosage = std::numeric_limits<double>::quiet_NaN(); break;
        }
    }
    st->used = used;
    if (used) {
        volatile double term = dosage * beta;  // same product as nimpress.nim:640, not contracted
        c->const_sum += term;
        c->host_nloci += 1;
    }
}

static int check_usable(nps_ctx *c) {
    if (!c) return fail(NPS_E_INVAL, "ctx is NULL");
    if (c->broken)
        return fail(NPS_E_STATE, "a previous call on this context failed half-way through its launches; "
                                 "call nps_reset before using it again");
    return NPS_OK;
}

static int materialize_resident_stats(nps_ctx *c);

// run the open batch: params -> accumulate; then collect its stats
static int run_batch(nps_ctx *c) {
    if (c->res_pending && !c->pending.empty()) {  // keep stats in push order
        int rc = materialize_resident_stats(c);
        if (rc) return rc;
    }
    if (c->batch_rows == 0 && c->ds_rows == 0) {
        // only host rows pending: move them to ready
        for (auto &p : c->pending) c->ready.push_back(p.host);
        c->pending.clear();
        return NPS_OK;
    }
    const uint32_t rows = c->batch_rows;
    if (rows) {
        const uint32_t rows_pad = (rows + 3) / 4 * 4;
        HIP_TRY(hipMemcpyAsync(c->d_desc, c->h_desc, sizeof(nps_row_desc) * rows,
                               hipMemcpyHostToDevice, c->stream));
        {
            ProfScope ps(c, P_PARAMS);
            HIP_TRY(launch_row_params(c->stream, c->d_tally, c->d_desc, rows, rows_pad, c->n,
                                      dev_params(c->params), c->d_lut, c->d_stats, c->d_nloci));
        }
        if (c->n) {
            int rc = ensure_all_chunks(c);
            if (rc) return rc;
            ProfScope ps(c, P_ACCUM);
            HIP_TRY(launch_accumulate(c->stream, c->d_codes, c->stride_words, rows, c->d_lut, c->geom,
                                      c->d_part));
        }
        HIP_TRY(hipMemcpyAsync(c->h_stats, c->d_stats, sizeof(nps_locus_stat) * rows,
                               hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemsetAsync(c->d_tally, 0, sizeof(unsigned long long) * rows, c->stream));
    }
    const uint32_t drows = c->ds_rows;
    if (drows) {
        HIP_TRY(hipMemcpyAsync(c->d_ds_desc, c->h_ds_desc, sizeof(nps_row_desc) * drows,
                               hipMemcpyHostToDevice, c->stream));
        {
            ProfScope ps(c, P_TALLY);
            HIP_TRY(launch_ds_tally(c->stream, c->d_ds, c->ds_stride_f, c->n, c->d_ds_desc, drows,
                                    c->d_ds_tally));
        }
        {
            ProfScope ps(c, P_PARAMS);
            HIP_TRY(launch_ds_params(c->stream, c->d_ds_tally, c->d_ds_desc, drows, c->n,
                                     dev_params(c->params), c->d_ds_rowp, c->d_ds_stats, c->d_nloci));
        }
        {
            int rc = ensure_all_chunks(c);
            if (rc) return rc;
            ProfScope ps(c, P_ACCUM);
            HIP_TRY(launch_ds_accumulate(c->stream, c->d_ds, c->ds_stride_f, c->n, c->d_ds_rowp, drows,
                                         c->d_part, c->n_chunks, c->geom.part_chunk_stride));
        }
        HIP_TRY(hipMemcpyAsync(c->h_ds_stats, c->d_ds_stats, sizeof(nps_locus_stat) * drows,
                               hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (auto &p : c->pending)
        c->ready.push_back(p.batch_idx < 0 ? p.host
                                           : (p.is_ds ? c->h_ds_stats[p.batch_idx] : c->h_stats[p.batch_idx]));
    c->pending.clear();
    c->batch_rows = 0;
    c->ds_rows = 0;
    return NPS_OK;
}

static int begin_data_row(nps_ctx *c, int ref_is_effect, double beta, double eaf, uint32_t *slot) {
    if (c->batch_rows == c->batch_cap) {
        int rc = run_batch(c);
        if (rc) return rc;
    }
    *slot = c->batch_rows;
    nps_row_desc &d = c->h_desc[*slot];
    d.beta = beta;
    d.eaf = eaf;
    d.kind = NPS_ROW_PRESENT;
    d.ref_is_effect = ref_is_effect ? 1 : 0;
    return NPS_OK;
}

static void commit_data_row(nps_ctx *c, uint32_t slot) {
    PendingRow p;
    p.batch_idx = (int32_t)slot;
    p.is_ds = 0;
    memset(&p.host, 0, sizeof p.host);
    c->pending.push_back(p);
    c->batch_rows = slot + 1;
}

static int ensure_ds(nps_ctx *c);

// ploidy > 2: the dosage can exceed 2, which the 2-bit codes cannot hold (nimpress is documented as
// diploid-specific, README.md:158, but its loop counts any number of alleles, nim:385-390).  The
// record is decoded on the device into a float dosage row and scored through the DS path.
static int push_gt_polyploid(nps_ctx *c, const void *gts, int elem_bytes, int ploidy, int eaidx,
                             int ref_is_effect, double beta, double eaf) {
    int rc = ensure_ds(c);
    if (rc) return rc;
    if (c->ds_rows == c->ds_cap) {
        rc = run_batch(c);
        if (rc) return rc;
    }
    const uint32_t slot = c->ds_rows;
    nps_row_desc &d = c->h_ds_desc[slot];
    d.beta = beta;
    d.eaf = eaf;
    d.kind = NPS_ROW_PRESENT;
    // the decoded row already counts the effect allele: flag bit 1 tells the DS kernels not to apply
    // the 2 - DS transform, bit 0 still selects the homref imputation value (nim:435-438)
    d.ref_is_effect = (ref_is_effect ? 1 : 0) | 2;
    if (c->n) {
        const size_t bytes = (size_t)elem_bytes * (size_t)ploidy * c->n;
        if (bytes > c->poly_cap) {  // the staging grows to the widest record seen (rare: once per context)
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->poly_cap = 0;
            for (int k = 0; k < 2; ++k) {
                (void)hipHostFree(c->h_poly[k]);
                c->h_poly[k] = nullptr;
                if (!c->ev_poly[k]) HIP_TRY(hipEventCreateWithFlags(&c->ev_poly[k], hipEventDisableTiming));
            }
            HIP_TRY(hipHostMalloc(&c->h_poly[0], bytes));
            HIP_TRY(hipHostMalloc(&c->h_poly[1], bytes));
            c->poly_cap = bytes;
        }
        // as for diploid rows: the caller may reuse `gts` on return, so it is copied into a pinned ring
        // slot first (which the decode kernel reads where it lies); no stream synchronisation per row
        const int k = c->poly_next;
        c->poly_next ^= 1;
        HIP_TRY(hipEventSynchronize(c->ev_poly[k]));
        memcpy(c->h_poly[k], gts, bytes);
        {
            ProfScope ps(c, P_DECODE);
            HIP_TRY(launch_decode_gt_to_ds(c->stream, c->h_poly[k], elem_bytes, c->n, ploidy, eaidx,
                                           c->d_ds + (uint64_t)slot * c->ds_stride_f));
        }
        HIP_TRY(hipEventRecord(c->ev_poly[k], c->stream));
    }
    PendingRow p;
    p.batch_idx = (int32_t)slot;
    p.is_ds = 1;
    memset(&p.host, 0, sizeof p.host);
    c->pending.push_back(p);
    c->ds_rows = slot + 1;
    return NPS_OK;
}

static int push_gt_typed(nps_ctx *c, const void *gts, int elem_bytes, int ploidy, int eaidx,
                         int ref_is_effect, double beta, double eaf) {
    if (int urc = check_usable(c)) return urc;
    if (c->n && !gts) return fail(NPS_E_INVAL, "gts is NULL");
    if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4)
        return fail(NPS_E_INVAL, "elem_bytes %d (1, 2 or 4)", elem_bytes);
    if (ploidy < 1) return fail(NPS_E_INVAL, "ploidy %d < 1", ploidy);
    if (ploidy > 8) return fail(NPS_E_UNSUPPORTED, "ploidy %d > 8", ploidy);
    if (eaidx < 0) return fail(NPS_E_INVAL, "eaidx %d < 0 (nimpress.nim:380 doAssert)", eaidx);
    HIP_TRY(hipSetDevice(c->device));
    if (ploidy > 2)
        return push_gt_polyploid(c, gts, elem_bytes, ploidy, eaidx, ref_is_effect, beta, eaf);
    uint32_t slot;
    int rc = begin_data_row(c, ref_is_effect, beta, eaf, &slot);
    if (rc) return rc;
    if (c->n) {
        // the caller may reuse `gts` on return: copy it into a pinned ring slot first
        const int k = c->raw_next;
        c->raw_next = (k + 1) % nps_ctx::kRawSlots;
        HIP_TRY(hipEventSynchronize(c->ev_raw[k]));
        const size_t bytes = (size_t)elem_bytes * (size_t)ploidy * c->n;
        memcpy(c->h_raw[k], gts, bytes);
        // the decode kernel reads the pinned slot itself, over PCIe: one launch per row instead of a DMA
        // copy and a launch that wait for each other (copy engine <-> compute queue, ~50 us per row)
        {
            ProfScope ps(c, P_DECODE);
            HIP_TRY(launch_decode_gt(c->stream, c->h_raw[k], elem_bytes, c->n, ploidy, eaidx,
                                     c->d_codes + (uint64_t)(slot >> 2) * c->stride_words * 4, slot & 3,
                                     c->d_tally + slot));
        }
        HIP_TRY(hipEventRecord(c->ev_raw[k], c->stream));
    }
    commit_data_row(c, slot);
    return NPS_OK;
}

extern "C" int nps_push_gt(nps_ctx *c, const int32_t *gts, int ploidy, int eaidx, int ref_is_effect,
                           double beta, double eaf) {
    return push_gt_typed(c, gts, 4, ploidy, eaidx, ref_is_effect, beta, eaf);
}

extern "C" int nps_push_gt_raw(nps_ctx *c, const void *gt, int elem_bytes, int ploidy, int eaidx,
                               int ref_is_effect, double beta, double eaf) {
    return push_gt_typed(c, gt, elem_bytes, ploidy, eaidx, ref_is_effect, beta, eaf);
}

extern "C" int nps_push_packed(nps_ctx *c, const uint32_t *row, int ref_is_effect, double beta,
                               double eaf) {
    if (int urc = check_usable(c)) return urc;
    if (c->n && !row) return fail(NPS_E_INVAL, "row is NULL");
    HIP_TRY(hipSetDevice(c->device));
    uint32_t slot;
    int rc = begin_data_row(c, ref_is_effect, beta, eaf, &slot);
    if (rc) return rc;
    if (c->n) {
        const int k = c->raw_next;
        c->raw_next = (k + 1) % nps_ctx::kRawSlots;
        HIP_TRY(hipEventSynchronize(c->ev_raw[k]));
        memcpy(c->h_raw[k], row, sizeof(uint32_t) * c->n_words);
        {
            ProfScope ps(c, P_TALLY);
            HIP_TRY(launch_tally_scatter_row(c->stream, reinterpret_cast<const uint32_t *>(c->h_raw[k]), c->n, -1,
                                             c->d_codes + (uint64_t)(slot >> 2) * c->stride_words * 4,
                                             slot & 3, c->d_tally + slot));
        }
        HIP_TRY(hipEventRecord(c->ev_raw[k], c->stream));
    }
    commit_data_row(c, slot);
    return NPS_OK;
}

extern "C" int nps_push_bed(nps_ctx *c, const uint8_t *bed_row, int effect_is_a1, int ref_is_effect,
                            double beta, double eaf) {
    if (int urc = check_usable(c)) return urc;
    if (c->n && !bed_row) return fail(NPS_E_INVAL, "bed_row is NULL");
    if (effect_is_a1 < 0 || effect_is_a1 > NPS_MAP_PGEN_REF) return fail(NPS_E_INVAL, "bad code map %d", effect_is_a1);
    HIP_TRY(hipSetDevice(c->device));
    uint32_t slot;
    int rc = begin_data_row(c, ref_is_effect, beta, eaf, &slot);
    if (rc) return rc;
    if (c->n) {
        const int k = c->raw_next;
        c->raw_next = (k + 1) % nps_ctx::kRawSlots;
        HIP_TRY(hipEventSynchronize(c->ev_raw[k]));
        const size_t bytes = (size_t)((c->n + 3) / 4), padded = sizeof(uint32_t) * c->n_words;
        memcpy(c->h_raw[k], bed_row, bytes);
        memset((char *)c->h_raw[k] + bytes, 0, padded - bytes);
        {
            ProfScope ps(c, P_TALLY);
            HIP_TRY(launch_tally_scatter_row(c->stream, reinterpret_cast<const uint32_t *>(c->h_raw[k]), c->n,
                                             effect_is_a1,
                                             c->d_codes + (uint64_t)(slot >> 2) * c->stride_words * 4,
                                             slot & 3, c->d_tally + slot));
        }
        HIP_TRY(hipEventRecord(c->ev_raw[k], c->stream));
    }
    commit_data_row(c, slot);
    return NPS_OK;
}

// lazily allocate the DS streaming batch
static int ensure_ds(nps_ctx *c) {
    if (c->d_ds) return NPS_OK;
    c->ds_stride_f = ds_stride_floats(c->n);
    const uint64_t row_bytes = c->ds_stride_f * 4;
    uint64_t cap = (64ull << 20) / row_bytes;
    cap = std::max<uint64_t>(4, std::min<uint64_t>(cap, 1024));
    c->ds_cap = (uint32_t)cap;
    HIP_TRY(hipMalloc(&c->d_ds, row_bytes * cap));
    HIP_TRY(hipMemsetAsync(c->d_ds, 0, row_bytes * cap, c->stream));
    HIP_TRY(hipMalloc(&c->d_ds_desc, sizeof(nps_row_desc) * cap));
    HIP_TRY(hipMalloc(&c->d_ds_tally, sizeof(DsTally) * cap));
    HIP_TRY(hipMalloc(&c->d_ds_rowp, sizeof(DsRowP) * cap));
    HIP_TRY(hipMalloc(&c->d_ds_stats, sizeof(nps_locus_stat) * cap));
    auto up = [](size_t v) { return (v + 4095) / 4096 * 4096; };
    const size_t sz_desc = up(sizeof(nps_row_desc) * cap), sz_stats = up(sizeof(nps_locus_stat) * cap);
    const size_t sz_raw = up(sizeof(float) * std::max<uint64_t>(c->n, 1));
    size_t total = (sz_desc + sz_stats + 2 * sz_raw + 65535) / 65536 * 65536;
    HIP_TRY(hipHostMalloc(&c->h_ds_arena, total));
    char *p = (char *)c->h_ds_arena;
    c->h_ds_desc = (nps_row_desc *)p;
    p += sz_desc;
    c->h_ds_stats = (nps_locus_stat *)p;
    p += sz_stats;
    for (int k = 0; k < 2; ++k) {
        c->h_ds_raw[k] = (float *)p;
        p += sz_raw;
        HIP_TRY(hipEventCreateWithFlags(&c->ev_ds_raw[k], hipEventDisableTiming));
    }
    return NPS_OK;
}

extern "C" int nps_push_ds(nps_ctx *c, const float *ds, int ref_is_effect, double beta, double eaf) {
    if (int urc = check_usable(c)) return urc;
    if (c->n && !ds) return fail(NPS_E_INVAL, "ds is NULL");
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_ds(c);
    if (rc) return rc;
    if (c->ds_rows == c->ds_cap) {
        rc = run_batch(c);
        if (rc) return rc;
    }
    const uint32_t slot = c->ds_rows;
    nps_row_desc &d = c->h_ds_desc[slot];
    d.beta = beta;
    d.eaf = eaf;
    d.kind = NPS_ROW_PRESENT;
    d.ref_is_effect = ref_is_effect ? 1 : 0;
    if (c->n) {
        const int k = c->ds_raw_next;
        c->ds_raw_next = (k + 1) % 2;
        HIP_TRY(hipEventSynchronize(c->ev_ds_raw[k]));
        memcpy(c->h_ds_raw[k], ds, sizeof(float) * c->n);
        HIP_TRY(hipMemcpyAsync(c->d_ds + (uint64_t)slot * c->ds_stride_f, c->h_ds_raw[k],
                               sizeof(float) * c->n, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipEventRecord(c->ev_ds_raw[k], c->stream));
    }
    PendingRow p;
    p.batch_idx = (int32_t)slot;
    p.is_ds = 1;
    memset(&p.host, 0, sizeof p.host);
    c->pending.push_back(p);
    c->ds_rows = slot + 1;
    return NPS_OK;
}

extern "C" int nps_push_locus(nps_ctx *c, int kind, int ref_is_effect, double beta, double eaf) {
    if (int urc = check_usable(c)) return urc;
    if (kind != NPS_ROW_UNCOVERED && kind != NPS_ROW_ABSENT && kind != NPS_ROW_FILTERED)
        return fail(NPS_E_INVAL, "kind %d is not a no-data row kind", kind);
    PendingRow p;
    p.batch_idx = -1;
    p.is_ds = 0;
    host_locus_row(c, kind, ref_is_effect, beta, eaf, &p.host);
    c->pending.push_back(p);
    return NPS_OK;
}

// the device's result block (used rows, status bits) -> pinned host copy; the caller synchronises
static int fetch_result(nps_ctx *c) {
    HIP_TRY(hipMemcpyAsync(c->h_result, c->d_nloci, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                           c->stream));
    return NPS_OK;
}

// after a stream synchronisation that followed fetch_result: did a bounded wait inside a fused kernel
// expire?  The bit is sticky until nps_reset (the scores of this context are invalid).
static int check_status(nps_ctx *c) {
    if (c->h_result[1] & 1ull)
        return fail(NPS_E_TIMEOUT, "fused kernel: a bounded inter-workgroup wait expired; the scores "
                                   "of this context are invalid (nps_reset and retry in NPS_MODE_TWOPASS)");
    return NPS_OK;
}

extern "C" int nps_flush(nps_ctx *c, nps_locus_stat *stats_out, size_t cap, size_t *n_out) {
    int rc = check_usable(c);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    if (stats_out) rc = materialize_resident_stats(c);
    if (rc) return rc;
    rc = run_batch(c);
    if (rc) return rc;
    rc = fetch_result(c);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));
    rc = check_status(c);
    if (rc) return rc;
    size_t n = 0;
    if (stats_out) {
        while (c->ready_cursor < c->ready.size() && n < cap) stats_out[n++] = c->ready[c->ready_cursor++];
        if (c->ready_cursor == c->ready.size()) {
            c->ready.clear();
            c->ready_cursor = 0;
        }
    }
    if (n_out) *n_out = n;
    return NPS_OK;
}

// nimpress.nim:643-649 on the device, one enqueue and ONE synchronisation: the finish kernel reads
// the device's count of used rows itself, the count and the status bits come back in the same copy
// as (optionally) the scores.
static int finish_common(nps_ctx *c, double offset, double *d_dst, double *h_scores_out,
                         uint64_t *nloci_out, bool normalise = true) {
    int rc = run_batch(c);
    if (rc) return rc;
    if (c->n) {
        ProfScope ps(c, P_REDUCE);
        HIP_TRY(launch_finish(c->stream, c->d_part, c->chunks_used, c->geom.part_chunk_stride, c->n,
                              c->const_sum, c->d_nloci, c->host_nloci, normalise ? 1 : 0, offset, d_dst));
    }
    rc = fetch_result(c);
    if (rc) return rc;
    if (h_scores_out && c->n)
        HIP_TRY(hipMemcpyAsync(h_scores_out, d_dst, sizeof(double) * c->n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    rc = check_status(c);
    if (rc) return rc;
    if (nloci_out) *nloci_out = c->host_nloci + c->h_result[0];
    return NPS_OK;
}

extern "C" int nps_finish(nps_ctx *c, double offset, double *scores_out, uint64_t *nloci_out) {
    int rc = check_usable(c);
    if (rc) return rc;
    if (c->n && !scores_out) return fail(NPS_E_INVAL, "scores_out is NULL");
    HIP_TRY(hipSetDevice(c->device));
    return finish_common(c, offset, c->d_scores, scores_out, nloci_out);
}

extern "C" int nps_finish_device(nps_ctx *c, double offset, double *d_scores_out,
                                 uint64_t *nloci_out) {
    int rc = check_usable(c);
    if (rc) return rc;
    if (c->n && !d_scores_out) return fail(NPS_E_INVAL, "d_scores_out is NULL");
    HIP_TRY(hipSetDevice(c->device));
    return finish_common(c, offset, d_scores_out, nullptr, nloci_out);
}

extern "C" int nps_partial_device(nps_ctx *c, double *d_sums_out, uint64_t *nloci_out) {
    int rc = check_usable(c);
    if (rc) return rc;
    if (c->n && !d_sums_out) return fail(NPS_E_INVAL, "d_sums_out is NULL");
    HIP_TRY(hipSetDevice(c->device));
    return finish_common(c, 0.0, d_sums_out, nullptr, nloci_out, false);
}

extern "C" int nps_normalize_device(nps_ctx *c, double *d_sums, uint64_t nloci, double offset) {
    if (!c) return fail(NPS_E_INVAL, "ctx is NULL");
    if (c->n && !d_sums) return fail(NPS_E_INVAL, "d_sums_inout is NULL");
    HIP_TRY(hipSetDevice(c->device));
    if (c->n) {
        // one "chunk" = the reduced sums themselves, in place (every thread reads and writes its own i)
        ProfScope ps(c, P_REDUCE);
        HIP_TRY(launch_finish(c->stream, d_sums, 1, c->n, c->n, 0.0, nullptr, nloci, 1, offset, d_sums));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    return NPS_OK;
}

// ------------------------------------------------------------------------------------------
// resident cohort
extern "C" int nps_cohort_create(nps_cohort **out, int device, uint64_t n_samples, uint64_t n_rows,
                                 int format) {
    if (!out) return fail(NPS_E_INVAL, "out is NULL");
    *out = nullptr;
    if (format != NPS_FMT_GT2 && format != NPS_FMT_DS32 && format != NPS_FMT_GT2M && format != NPS_FMT_GT2X &&
        format != NPS_FMT_GT_AUTO && format != NPS_FMT_DS16)
        return fail(NPS_E_INVAL, "unknown cohort format %d", format);
    if (n_samples > 0x7fffffffull) return fail(NPS_E_UNSUPPORTED, "n_samples too large");
    int rc = select_device(device);
    if (rc) return rc;
    if (format == NPS_FMT_GT_AUTO) {
        // The strip layout at every size (round 5).  Where the single-read kernel's grid -- P strips x floor(CUs / P) row
        // teams, all resident -- covers the chip (up to 180 000 samples and 366 593 .. 522 240 on 256 compute units) the
        // tallies are counted in the pass.  Elsewhere (147 strips at 300 000 samples: 147 of 256 compute units; more
        // strips than compute units beyond 522 240) nps_score_cohort[_def] under NPS_MODE_AUTO counts the cohort's
        // tallies ONCE, keeps them with the cohort (nps_cohort_keep_tallies) and scores with the tallies given: an
        // ordinary grid, one read per pass from then on, time independent of the genotypes.  (Until round 4 such sizes got
        // the row layout, whose table-lookup kernel runs at 0.47 .. 0.63 of the roofline depending on the genotypes.)
        format = n_samples < (1ull << 27) ? NPS_FMT_GT2X : NPS_FMT_GT2;
    }
    nps_cohort *c = new (std::nothrow) nps_cohort;
    if (!c) return fail(NPS_E_NOMEM, "out of host memory");
    c->device = device;
    c->format = format;
    c->n_samples = n_samples;
    c->n_rows = n_rows;
    c->stride_bytes = format == NPS_FMT_DS32   ? ds_stride_floats(n_samples) * 4
                      : format == NPS_FMT_DS16 ? ds_stride_floats(n_samples) * 2  // (a multiple of 128 bytes)
                                               : stride_words_for(n_samples) * 4;
    const uint64_t rows_alloc = format == NPS_FMT_GT2 ? (n_rows + 3) / 4 * 4 : n_rows;
    uint64_t bytes = std::max<uint64_t>(c->stride_bytes * rows_alloc, 256);
    if (format == NPS_FMT_GT2M) {
        c->stride_bytes = 0;  // not row-major: 1 KiB units of 128 rows x 32 samples
        bytes = std::max<uint64_t>(gt2m_bytes(n_samples, n_rows), 256);
        const uint64_t tb = sizeof(unsigned long long) * std::max<uint64_t>(gt2m_superblocks(n_rows) * 128, 1);
        if (hipMalloc(&c->d_row_tally, tb) != hipSuccess || hipMemset(c->d_row_tally, 0, tb) != hipSuccess) {
            (void)hipFree(c->d_row_tally);
            delete c;
            return fail(NPS_E_NOMEM, "hipMalloc of the row tallies failed");
        }
    }
    if (format == NPS_FMT_GT2X) {
        c->stride_bytes = 0;  // not row-major: strips x superblocks x 1 KiB units
        bytes = std::max<uint64_t>(gt2x_bytes(n_samples, n_rows), 256);
    }
    hipError_t e = hipMalloc(&c->d_data, bytes);
    if (e != hipSuccess) {
        (void)hipFree(c->d_row_tally);
        delete c;
        return fail(NPS_E_NOMEM, "hipMalloc(%llu bytes) for the cohort failed: %s",
                    (unsigned long long)bytes, hipGetErrorString(e));
    }
    e = hipMemset(c->d_data, 0, bytes);
    if (e != hipSuccess) {
        (void)hipFree(c->d_data);
        (void)hipFree(c->d_row_tally);
        delete c;
        return fail(NPS_E_HIP, "hipMemset failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return NPS_OK;
}

extern "C" uint64_t nps_cohort_row_stride(const nps_cohort *c) { return c ? c->stride_bytes : 0; }
extern "C" uint64_t nps_cohort_n_rows(const nps_cohort *c) { return c ? c->n_rows : 0; }
extern "C" int nps_cohort_format(const nps_cohort *c) { return c ? c->format : -1; }

extern "C" void nps_cohort_destroy(nps_cohort *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(c->d_data);
    (void)hipFree(c->d_row_tally);
    (void)hipFree(c->d_mx_row_tally);
    (void)hipFree(c->d_push_tally);
    for (int k = 0; k < 2; ++k) {
        (void)hipHostFree(c->h_push[k]);
        if (c->ev_push[k]) (void)hipEventDestroy(c->ev_push[k]);
    }
    if (c->push_stream) (void)hipStreamDestroy(c->push_stream);
    delete c;
}

// back to the plain layout (before rows are written into an optimised cohort); the transform is its own inverse
static int cohort_unoptimize(nps_cohort *c) {
    if (!c->optimized) return NPS_OK;
    hipError_t e = launch_cohort_parity(nullptr, (uint32_t *)c->d_data, c->stride_bytes / 4, c->n_samples, c->n_rows);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail(NPS_E_HIP, "cohort re-ordering failed: %s", hipGetErrorString(e));
    c->optimized = false;
    return NPS_OK;
}

extern "C" int nps_cohort_optimize(nps_cohort *c) {
    if (!c) return fail(NPS_E_INVAL, "cohort is NULL");
    if (c->format != NPS_FMT_GT2 || c->optimized || c->n_rows == 0 || c->n_samples == 0) return NPS_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());  // no scoring kernel may still be reading the rows rewritten here
    hipError_t e = launch_cohort_parity(nullptr, (uint32_t *)c->d_data, c->stride_bytes / 4, c->n_samples, c->n_rows);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail(NPS_E_HIP, "cohort optimisation failed: %s", hipGetErrorString(e));
    c->optimized = true;
    return NPS_OK;
}

static int check_range(const nps_cohort *c, uint64_t row0, uint64_t nrows) {
    if (!c) return fail(NPS_E_INVAL, "cohort is NULL");
    if (row0 > c->n_rows || nrows > c->n_rows - row0)
        return fail(NPS_E_INVAL, "rows [%llu,+%llu) outside cohort of %llu rows",
                    (unsigned long long)row0, (unsigned long long)nrows,
                    (unsigned long long)c->n_rows);
    return NPS_OK;
}

// GT2 rows <-> the group-interleaved device layout, through a host buffer of whole groups
static int gt2_transfer(const nps_cohort *c, uint64_t row0, uint64_t nrows, void *host_rows,
                        size_t host_stride, bool to_device) {
    const uint64_t n_words = words_for(c->n_samples), sw = c->stride_bytes / 4;
    if (row0 & 3) return fail(NPS_E_INVAL, "row0 must be a multiple of 4 for 2-bit cohorts");
    const uint64_t chunk_groups = std::max<uint64_t>(1, (64ull << 20) / (sw * 16));
    std::vector<uint32_t> buf;
    for (uint64_t r = 0; r < nrows; r += chunk_groups * 4) {
        const uint64_t k = std::min<uint64_t>(chunk_groups * 4, nrows - r);  // rows in this chunk
        const uint64_t groups = (k + 3) / 4;
        char *dev = (char *)c->d_data + ((row0 + r) >> 2) * sw * 16;
        buf.assign(groups * sw * 4, 0u);
        if (!to_device) {
            HIP_TRY(hipMemcpy(buf.data(), dev, groups * sw * 16, hipMemcpyDeviceToHost));
            if (c->optimized)  // parity layout -> plain, on the copy
                for (uint64_t x = 0; x < groups * sw; ++x) {
                    uint32_t *q = buf.data() + x * 4;
                    q[0] = parity_fix(q[0], q[1], q[2], q[3]);
                }
        }
        for (uint64_t j = 0; j < k; ++j) {
            uint32_t *hrow = (uint32_t *)((char *)host_rows + (r + j) * host_stride);
            uint32_t *g = buf.data() + (j >> 2) * sw * 4 + (j & 3);
            if (to_device)
                for (uint64_t w = 0; w < n_words; ++w) g[w * 4] = word_to_planes(hrow[w]);
            else
                for (uint64_t w = 0; w < n_words; ++w) hrow[w] = word_from_planes(g[w * 4]);
        }
        // rows of a last partial group that are not part of this upload become zero
        if (to_device)
            HIP_TRY(hipMemcpy(dev, buf.data(), groups * sw * 16, hipMemcpyHostToDevice));
    }
    return NPS_OK;
}

// rows of a host buffer -> staging on the device -> interleave (and, for .bed rows, recode) kernel.
// width = bytes of one source row; bed_mode: nullptr (native codes) or per-row effect-is-A1 flags.
static int gt2_upload(nps_cohort *c, uint64_t row0, uint64_t nrows, const void *host_rows,
                      size_t host_stride, size_t width, const uint8_t *bed_mode) {
    if (row0 & 3) return fail(NPS_E_INVAL, "row0 must be a multiple of 4 for 2-bit cohorts");
    const uint64_t sw = c->stride_bytes / 4;  // word columns per group in the cohort
    const uint64_t src_stride_words = sw;     // staging rows padded the same way (zero filled)
    uint64_t chunk = std::max<uint64_t>(4, (256ull << 20) / (src_stride_words * 4)) / 4 * 4;
    chunk = std::min<uint64_t>(chunk, 4ull * 65535);
    chunk = std::min<uint64_t>(chunk, (nrows + 3) / 4 * 4);
    uint32_t *d_stage = nullptr;
    uint8_t *d_mode = nullptr;
    HIP_TRY(hipMalloc(&d_stage, chunk * src_stride_words * 4));
    hipError_t e = hipSuccess;
    if (bed_mode) e = hipMalloc(&d_mode, chunk);
    for (uint64_t r = 0; e == hipSuccess && r < nrows; r += chunk) {
        const uint64_t k = std::min(chunk, nrows - r);
        e = hipMemsetAsync(d_stage, 0, chunk * src_stride_words * 4, nullptr);
        if (e == hipSuccess)
            e = hipMemcpy2D(d_stage, src_stride_words * 4, (const char *)host_rows + r * host_stride,
                            host_stride, width, k, hipMemcpyHostToDevice);
        if (e == hipSuccess && bed_mode) e = hipMemcpy(d_mode, bed_mode + r, k, hipMemcpyHostToDevice);
        if (e == hipSuccess)
            e = launch_interleave_rows(nullptr, d_stage, src_stride_words, k, c->n_samples,
                                       bed_mode ? d_mode : nullptr,
                                       (uint32_t *)((char *)c->d_data + ((row0 + r) >> 2) * sw * 16), sw);
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    (void)hipFree(d_stage);
    (void)hipFree(d_mode);
    if (e != hipSuccess) return fail(NPS_E_HIP, "cohort upload failed: %s", hipGetErrorString(e));
    return NPS_OK;
}

// NPS_FMT_GT2X: plain rows (C-ABI order and codes) <-> units, through a device staging buffer of whole superblocks
static int gt2x_transfer(const nps_cohort *c, uint64_t row0, uint64_t nrows, void *host_rows, size_t host_stride,
                         bool to_device) {
    if (to_device && (row0 & 127))
        return fail(NPS_E_INVAL, "row0 must be a multiple of 128 for NPS_FMT_GT2X cohorts");
    const uint64_t n_words = words_for(c->n_samples), sw = (n_words + 1) / 2 * 2;
    const uint64_t chunk = std::max<uint64_t>(128, (256ull << 20) / (sw * 4) / 128 * 128);
    uint32_t *d_stage = nullptr;
    HIP_TRY(hipMalloc(&d_stage, std::min(chunk, (nrows + 127) / 128 * 128) * sw * 4));
    hipError_t e = hipSuccess;
    for (uint64_t r = 0; e == hipSuccess && r < nrows; r += chunk) {
        const uint64_t k = std::min(chunk, nrows - r);
        if (to_device) {
            e = hipMemcpy2D(d_stage, sw * 4, (const char *)host_rows + r * host_stride, host_stride, n_words * 4, k,
                            hipMemcpyHostToDevice);
            if (e == hipSuccess)
                e = launch_rows_to_gt2x(nullptr, d_stage, sw, c->n_samples, c->n_rows, row0 + r, k, c->d_data);
            if (e == hipSuccess) e = hipDeviceSynchronize();
        } else {
            e = launch_gt2x_to_rows(nullptr, c->d_data, c->n_samples, c->n_rows, row0 + r, k, d_stage, sw);
            if (e == hipSuccess)
                e = hipMemcpy2D((char *)host_rows + r * host_stride, host_stride, d_stage, sw * 4, n_words * 4, k,
                                hipMemcpyDeviceToHost);
        }
    }
    (void)hipFree(d_stage);
    if (e != hipSuccess) return fail(NPS_E_HIP, "cohort transfer failed: %s", hipGetErrorString(e));
    return NPS_OK;
}

extern "C" int nps_cohort_upload_bed(nps_cohort *c, uint64_t row0, uint64_t nrows, const uint8_t *bed_rows,
                                     size_t row_stride_bytes, const uint8_t *effect_is_a1) {
    int rc = check_range(c, row0, nrows);
    if (rc) return rc;
    if (c->format != NPS_FMT_GT2) return fail(NPS_E_INVAL, ".bed rows need a 2-bit (NPS_FMT_GT2) cohort");
    const size_t width = (size_t)((c->n_samples + 3) / 4);
    if (nrows == 0 || width == 0) return NPS_OK;
    if (!bed_rows || !effect_is_a1 || row_stride_bytes < width)
        return fail(NPS_E_INVAL, "bad .bed buffer / stride / flags");
    for (uint64_t r = 0; r < nrows; ++r)
        if (effect_is_a1[r] > NPS_MAP_PGEN_REF) return fail(NPS_E_INVAL, "row %llu: bad code map %d", (unsigned long long)r, (int)effect_is_a1[r]);
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());  // no scoring kernel may still be reading the rows replaced here
    c->mx_row_tally_valid = false;
    rc = cohort_unoptimize(c);
    if (rc) return rc;
    return gt2_upload(c, row0, nrows, bed_rows, row_stride_bytes, width, effect_is_a1);
}

// one row of a NPS_FMT_GT2 cohort from the buffer a VCF/BCF record holds, decoded on the device (the resident form of
// nps_push_gt_raw / nps_push_bed: same kernels, the cohort is the destination)
static int cohort_push_prepare(nps_cohort *c, uint64_t row, size_t bytes, int *slot) {
    if (!c) return fail(NPS_E_INVAL, "cohort is NULL");
    if (c->format != NPS_FMT_GT2) return fail(NPS_E_UNSUPPORTED, "rows are pushed into NPS_FMT_GT2 cohorts only");
    if (row >= c->n_rows) return fail(NPS_E_INVAL, "row %llu outside cohort of %llu rows", (unsigned long long)row,
                                      (unsigned long long)c->n_rows);
    HIP_TRY(hipSetDevice(c->device));
    if (!c->push_stream) {
        HIP_TRY(hipDeviceSynchronize());  // whatever wrote or read the cohort before
        int rc = cohort_unoptimize(c);
        if (rc) return rc;
        HIP_TRY(hipStreamCreateWithFlags(&c->push_stream, hipStreamNonBlocking));
        HIP_TRY(hipMalloc(&c->d_push_tally, 256));
        for (int k = 0; k < 2; ++k) HIP_TRY(hipEventCreateWithFlags(&c->ev_push[k], hipEventDisableTiming));
    }
    if (c->optimized) {
        HIP_TRY(hipDeviceSynchronize());
        int rc = cohort_unoptimize(c);
        if (rc) return rc;
    }
    if (bytes > c->push_cap) {
        HIP_TRY(hipStreamSynchronize(c->push_stream));
        for (int k = 0; k < 2; ++k) {
            (void)hipHostFree(c->h_push[k]);
            c->h_push[k] = nullptr;
        }
        c->push_cap = 0;
        HIP_TRY(hipHostMalloc(&c->h_push[0], bytes));
        HIP_TRY(hipHostMalloc(&c->h_push[1], bytes));
        c->push_cap = bytes;
    }
    *slot = c->push_next;
    c->push_next ^= 1;
    HIP_TRY(hipEventSynchronize(c->ev_push[*slot]));
    return NPS_OK;
}

extern "C" int nps_cohort_push_gt_raw(nps_cohort *c, uint64_t row, const void *gt, int elem_bytes, int ploidy,
                                      int eaidx) {
    if (elem_bytes != 1 && elem_bytes != 2 && elem_bytes != 4)
        return fail(NPS_E_INVAL, "elem_bytes %d (1, 2 or 4)", elem_bytes);
    if (ploidy < 1) return fail(NPS_E_INVAL, "ploidy %d < 1", ploidy);
    if (ploidy > 2) return fail(NPS_E_UNSUPPORTED, "ploidy %d > 2 does not fit the 2-bit cohort", ploidy);
    if (eaidx < 0) return fail(NPS_E_INVAL, "eaidx %d < 0 (nimpress.nim:380 doAssert)", eaidx);
    if (c && c->n_samples && !gt) return fail(NPS_E_INVAL, "gt is NULL");
    const size_t bytes = c ? (size_t)elem_bytes * (size_t)ploidy * c->n_samples : 0;
    int k = 0;
    int rc = cohort_push_prepare(c, row, std::max<size_t>(bytes, 16), &k);
    if (rc) return rc;
    if (c->n_samples == 0) return NPS_OK;
    memcpy(c->h_push[k], gt, bytes);
    const uint64_t sw = c->stride_bytes / 4;
    HIP_TRY(launch_decode_gt(c->push_stream, c->h_push[k], elem_bytes, c->n_samples, ploidy, eaidx,
                             (uint32_t *)c->d_data + (row >> 2) * sw * 4, (int)(row & 3), c->d_push_tally));
    HIP_TRY(hipEventRecord(c->ev_push[k], c->push_stream));
    return NPS_OK;
}

extern "C" int nps_cohort_push_bed(nps_cohort *c, uint64_t row, const uint8_t *bed_row, int effect_is_a1) {
    if (c && c->n_samples && !bed_row) return fail(NPS_E_INVAL, "bed_row is NULL");
    if (effect_is_a1 < 0 || effect_is_a1 > NPS_MAP_PGEN_REF) return fail(NPS_E_INVAL, "bad code map %d", effect_is_a1);
    const size_t bytes = c ? (size_t)((c->n_samples + 3) / 4) : 0;
    int k = 0;
    int rc = cohort_push_prepare(c, row, std::max<size_t>((bytes + 3) / 4 * 4 + 16, 16), &k);
    if (rc) return rc;
    if (c->n_samples == 0) return NPS_OK;
    memset(c->h_push[k], 0, (bytes + 3) / 4 * 4 + 16);
    memcpy(c->h_push[k], bed_row, bytes);
    const uint64_t sw = c->stride_bytes / 4;
    HIP_TRY(launch_tally_scatter_row(c->push_stream, reinterpret_cast<const uint32_t *>(c->h_push[k]), c->n_samples,
                                     effect_is_a1, (uint32_t *)c->d_data + (row >> 2) * sw * 4, (int)(row & 3),
                                     c->d_push_tally));
    HIP_TRY(hipEventRecord(c->ev_push[k], c->push_stream));
    return NPS_OK;
}

// NPS_FMT_DS16 <-> float32 rows of the host, through a float32 staging buffer of at most 256 MiB.  Upload: a row that holds
// a value which is not the value of a code (a decimal with at most four places in [0, 2], as float32) is REFUSED -- the
// format is lossless or it is not used; the rows of the range are undefined after a refusal.
static int ds16_transfer(nps_cohort *c, uint64_t row0, uint64_t nrows, void *host_rows, size_t host_stride, bool upload) {
    const uint64_t stride_f = ds_stride_floats(c->n_samples), stride_e = c->stride_bytes / 2;
    const uint64_t chunk = std::max<uint64_t>(1, std::min<uint64_t>(nrows, (256ull << 20) / (stride_f * 4)));
    float *d_stage = nullptr;
    unsigned char *d_bad = nullptr;
    HIP_TRY(hipMalloc(&d_stage, chunk * stride_f * 4));
    if (hipMalloc(&d_bad, chunk) != hipSuccess) {
        (void)hipFree(d_stage);
        return fail(NPS_E_NOMEM, "hipMalloc failed");
    }
    std::vector<unsigned char> bad(chunk);
    hipError_t e = hipSuccess;
    long long first_bad = -1;
    for (uint64_t r = 0; r < nrows && e == hipSuccess && first_bad < 0; r += chunk) {
        const uint64_t k = std::min(chunk, nrows - r);
        uint16_t *rows = (uint16_t *)c->d_data + (row0 + r) * stride_e;
        char *host = (char *)host_rows + r * host_stride;
        if (upload) {
            e = hipMemcpy2D(d_stage, stride_f * 4, host, host_stride, c->n_samples * 4, k, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = launch_ds16_pack(nullptr, d_stage, stride_f, c->n_samples, k, rows, stride_e, d_bad);
            if (e == hipSuccess) e = hipMemcpy(bad.data(), d_bad, k, hipMemcpyDeviceToHost);
            for (uint64_t j = 0; j < k && e == hipSuccess; ++j)
                if (bad[j]) {
                    first_bad = (long long)(row0 + r + j);
                    break;
                }
        } else {
            e = launch_ds16_unpack(nullptr, rows, stride_e, c->n_samples, k, d_stage, stride_f);
            if (e == hipSuccess)
                e = hipMemcpy2D(host, host_stride, d_stage, stride_f * 4, c->n_samples * 4, k, hipMemcpyDeviceToHost);
        }
    }
    (void)hipFree(d_stage);
    (void)hipFree(d_bad);
    if (e != hipSuccess) return fail(NPS_E_HIP, "NPS_FMT_DS16 transfer failed: %s", hipGetErrorString(e));
    if (first_bad >= 0)
        return fail(NPS_E_UNSUPPORTED, "row %lld holds a FORMAT/DS value that is not a decimal with at most four places in "
                    "[0, 2]: NPS_FMT_DS16 stores such values only (losslessly); use a NPS_FMT_DS32 cohort", first_bad);
    return NPS_OK;
}

extern "C" int nps_cohort_upload(nps_cohort *c, uint64_t row0, uint64_t nrows, const void *host_rows,
                                 size_t host_stride) {
    int rc = check_range(c, row0, nrows);
    if (rc) return rc;
    if (c->format == NPS_FMT_GT2M)
        return fail(NPS_E_UNSUPPORTED, "NPS_FMT_GT2M cohorts are filled by nps_cohort_convert (from a "
                                       "NPS_FMT_GT2 cohort) or by the synthetic generator");
    const bool ds_any = c->format == NPS_FMT_DS32 || c->format == NPS_FMT_DS16;  // (both take and give float32 rows)
    const size_t width = ds_any ? c->n_samples * 4 : words_for(c->n_samples) * 4;
    if (nrows == 0 || width == 0) return NPS_OK;
    if (!host_rows || host_stride < width) return fail(NPS_E_INVAL, "bad host buffer / stride");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());  // no scoring kernel may still be reading the rows replaced here
    c->mx_row_tally_valid = false;    // (kept tallies describe the rows as they were: nps_cohort_keep_tallies again)
    if (c->format == NPS_FMT_DS16) return ds16_transfer(c, row0, nrows, const_cast<void *>(host_rows), host_stride, true);
    if (c->format == NPS_FMT_GT2) {
        rc = cohort_unoptimize(c);
        if (rc) return rc;
        return gt2_upload(c, row0, nrows, host_rows, host_stride, width, nullptr);
    }
    if (c->format == NPS_FMT_GT2X) return gt2x_transfer(c, row0, nrows, const_cast<void *>(host_rows), host_stride, true);
    HIP_TRY(hipMemcpy2D((char *)c->d_data + row0 * c->stride_bytes, c->stride_bytes, host_rows,
                        host_stride, width, nrows, hipMemcpyHostToDevice));
    // the range of a dosage, row by row, where the rows now lie (one read at upload time, none when scoring)
    unsigned char *d_bad = nullptr;
    HIP_TRY(hipMalloc(&d_bad, nrows));
    std::vector<unsigned char> bad(nrows);
    hipError_t e = launch_ds_range_check(nullptr, (const float *)c->d_data + row0 * (c->stride_bytes / 4), c->stride_bytes / 4,
                                         c->n_samples, nrows, d_bad);
    if (e == hipSuccess) e = hipMemcpy(bad.data(), d_bad, nrows, hipMemcpyDeviceToHost);
    (void)hipFree(d_bad);
    if (e != hipSuccess) return fail(NPS_E_HIP, "range check of the uploaded dosages failed: %s", hipGetErrorString(e));
    if (c->ds_row_bad.size() != c->n_rows) c->ds_row_bad.assign(c->n_rows, 0);
    for (uint64_t r = 0; r < nrows; ++r) {
        c->ds_bad_rows += (uint64_t)bad[r] - (uint64_t)c->ds_row_bad[row0 + r];
        c->ds_row_bad[row0 + r] = bad[r];
    }
    return NPS_OK;
}

extern "C" int nps_cohort_download(const nps_cohort *c, uint64_t row0, uint64_t nrows,
                                   void *host_rows, size_t host_stride) {
    int rc = check_range(c, row0, nrows);
    if (rc) return rc;
    if (c->format == NPS_FMT_GT2M) return fail(NPS_E_UNSUPPORTED, "NPS_FMT_GT2M cohorts cannot be downloaded");
    const size_t width = c->format == NPS_FMT_DS32 || c->format == NPS_FMT_DS16 ? c->n_samples * 4 : words_for(c->n_samples) * 4;
    if (nrows == 0 || width == 0) return NPS_OK;
    if (!host_rows || host_stride < width) return fail(NPS_E_INVAL, "bad host buffer / stride");
    HIP_TRY(hipSetDevice(c->device));
    { int qrc = cohort_quiesce(c); if (qrc) return qrc; }
    if (c->format == NPS_FMT_DS16) return ds16_transfer(const_cast<nps_cohort *>(c), row0, nrows, host_rows, host_stride, false);
    if (c->format == NPS_FMT_GT2) return gt2_transfer(c, row0, nrows, host_rows, host_stride, false);
    if (c->format == NPS_FMT_GT2X) return gt2x_transfer(c, row0, nrows, host_rows, host_stride, false);
    HIP_TRY(hipMemcpy2D(host_rows, host_stride, (const char *)c->d_data + row0 * c->stride_bytes,
                        c->stride_bytes, width, nrows, hipMemcpyDeviceToHost));
    return NPS_OK;
}

extern "C" int nps_cohort_synth_rows(nps_cohort *c, uint64_t row0, uint64_t nrows, uint64_t gen_row0,
                                     uint64_t seed, const uint32_t *t_het, const uint32_t *t_hom,
                                     const uint32_t *t_miss) {
    int rc = check_range(c, row0, nrows);
    if (rc) return rc;
    if (nrows == 0) return NPS_OK;
    if (!t_het || !t_hom || !t_miss) return fail(NPS_E_INVAL, "threshold arrays are NULL");
    if (c->format == NPS_FMT_GT2 && (row0 & 3))
        return fail(NPS_E_INVAL, "row0 must be a multiple of 4 for 2-bit cohorts");
    if ((c->format == NPS_FMT_GT2M || c->format == NPS_FMT_GT2X) && (row0 & 127))
        return fail(NPS_E_INVAL, "row0 must be a multiple of 128 for NPS_FMT_GT2M / NPS_FMT_GT2X cohorts");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());  // no scoring kernel may still be reading the rows replaced here
    c->mx_row_tally_valid = false;
    rc = cohort_unoptimize(c);
    if (rc) return rc;
    uint32_t *d_t = nullptr;
    HIP_TRY(hipMalloc(&d_t, sizeof(uint32_t) * 3 * nrows));
    hipError_t e = hipMemcpy(d_t, t_het, sizeof(uint32_t) * nrows, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_t + nrows, t_hom, sizeof(uint32_t) * nrows, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_t + 2 * nrows, t_miss, sizeof(uint32_t) * nrows, hipMemcpyHostToDevice);
    const uint64_t step = 32768;
    for (uint64_t r = 0; e == hipSuccess && r < nrows; r += step) {
        const uint64_t k = std::min(step, nrows - r);
        if (c->format == NPS_FMT_GT2M)
            e = launch_synth_gt2m(nullptr, c->d_data, c->n_samples, row0 + r, gen_row0 + r, k, seed, d_t + r,
                                  d_t + nrows + r, d_t + 2 * nrows + r, c->d_row_tally);
        else if (c->format == NPS_FMT_GT2X)
            e = launch_synth_gt2x(nullptr, c->d_data, c->n_samples, c->n_rows, row0 + r, gen_row0 + r, k, seed, d_t + r,
                                  d_t + nrows + r, d_t + 2 * nrows + r);
        else if (c->format == NPS_FMT_DS16)
            e = launch_synth_ds16(nullptr, (uint16_t *)c->d_data, c->stride_bytes / 2, c->n_samples, row0 + r, gen_row0 + r, k,
                                  seed, d_t + r, d_t + nrows + r, d_t + 2 * nrows + r);
        else if (c->format == NPS_FMT_DS32)
            e = launch_synth_ds(nullptr, (float *)c->d_data, c->stride_bytes / 4, c->n_samples,
                                row0 + r, gen_row0 + r, k, seed, d_t + r, d_t + nrows + r, d_t + 2 * nrows + r);
        else
            e = launch_synth_gt(nullptr, (uint32_t *)c->d_data, c->stride_bytes / 4, c->n_samples,
                                row0 + r, gen_row0 + r, k, seed, d_t + r, d_t + nrows + r, d_t + 2 * nrows + r);
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(d_t);
    if (e != hipSuccess) return fail(NPS_E_HIP, "synthetic fill failed: %s", hipGetErrorString(e));
    if (c->format == NPS_FMT_DS32 && c->ds_bad_rows)  // the generator clips to [0, 2]: these rows are in range now
        for (uint64_t r = row0; r < row0 + nrows && r < c->ds_row_bad.size(); ++r) {
            c->ds_bad_rows -= c->ds_row_bad[r];
            c->ds_row_bad[r] = 0;
        }
    return NPS_OK;
}

extern "C" int nps_cohort_synth(nps_cohort *c, uint64_t row0, uint64_t nrows, uint64_t seed,
                                const uint32_t *t_het, const uint32_t *t_hom, const uint32_t *t_miss) {
    return nps_cohort_synth_rows(c, row0, nrows, row0, seed, t_het, t_hom, t_miss);
}

// ------------------------------------------------------------------------------------------
// score definitions kept on the device
struct nps_scoredef {
    int device = 0;
    uint64_t n_desc = 0;
    uint64_t m = 0;                       // PRESENT rows (consume cohort rows)
    std::vector<nps_row_desc> host_rows;  // rows without genotype data, in order
    std::vector<int64_t> data_index;      // per desc: >= 0 index among PRESENT rows, -1 host row
    nps_row_desc *d_desc = nullptr;       // [m] PRESENT rows, compact
    // NPS_FMT_GT2X runs: the largest |beta| (4 + max(2, 2 |eaf|)) over the PRESENT rows with finite numbers: the
    // fixed-point scale 2^F of fused_mx_kernel keeps every weight below 2^56
    double mx_bound = 0.0;
    // ... and when the definition's weights span more than 2^30 (a sample that carries only its small-beta rows would see
    // the quantisation of the largest one): magnitude bands of 2^30, each a copy of the PRESENT rows with beta = 0
    // outside the band, scored one pass per band with the band's own scale (empty: one band, d_desc itself)
    struct MxBand {
        double bound = 0.0;
        nps_row_desc *d_desc = nullptr;
    };
    std::vector<MxBand> mx_bands;
};
constexpr int kMxBandBits = 30, kMxMaxBands = 8;

extern "C" int nps_scoredef_create(nps_scoredef **out, int device, const nps_row_desc *rows,
                                   uint64_t n_desc) {
    if (!out) return fail(NPS_E_INVAL, "out is NULL");
    *out = nullptr;
    if (n_desc && !rows) return fail(NPS_E_INVAL, "rows is NULL");
    int rc = select_device(device);
    if (rc) return rc;
    nps_scoredef *d = new (std::nothrow) nps_scoredef;
    if (!d) return fail(NPS_E_NOMEM, "out of host memory");
    d->device = device;
    d->n_desc = n_desc;
    std::vector<nps_row_desc> data;
    data.reserve(n_desc);
    d->data_index.resize(n_desc);
    for (uint64_t j = 0; j < n_desc; ++j) {
        const nps_row_desc &r = rows[j];
        if (r.kind == NPS_ROW_PRESENT) {
            d->data_index[j] = (int64_t)data.size();
            data.push_back(r);
            if (std::isfinite(r.beta)) {
                const double ie = std::isfinite(r.eaf) ? std::max(2.0, 2.0 * std::fabs(r.eaf)) : 2.0;
                d->mx_bound = std::max(d->mx_bound, std::fabs(r.beta) * (4.0 + ie));
            }
        } else if (r.kind == NPS_ROW_UNCOVERED || r.kind == NPS_ROW_ABSENT ||
                   r.kind == NPS_ROW_FILTERED) {
            d->data_index[j] = -1;
            d->host_rows.push_back(r);
        } else {
            delete d;
            return fail(NPS_E_INVAL, "row %llu: bad kind %d", (unsigned long long)j, r.kind);
        }
    }
    d->m = data.size();
    if (d->m > 0xfffffff0ull) {
        delete d;
        return fail(NPS_E_UNSUPPORTED, "too many rows");
    }
    if (d->m) {
        hipError_t e = hipMalloc(&d->d_desc, sizeof(nps_row_desc) * d->m);
        if (e == hipSuccess)
            e = hipMemcpy(d->d_desc, data.data(), sizeof(nps_row_desc) * d->m, hipMemcpyHostToDevice);
        // magnitude bands for the fixed-point kernel (north star: 1e-6 RELATIVE for every sample, also one whose only
        // rows are the definition's smallest): band b holds the rows with bound 2^-30(b+1) < v <= bound 2^-30b
        std::vector<int> band(d->m, 0);
        int n_bands = 1;
        if (e == hipSuccess && d->mx_bound > 0.0) {
            for (uint64_t j = 0; j < d->m; ++j) {
                const nps_row_desc &r = data[j];
                if (!std::isfinite(r.beta) || r.beta == 0.0) continue;
                const double ie = std::isfinite(r.eaf) ? std::max(2.0, 2.0 * std::fabs(r.eaf)) : 2.0;
                int e_top = 0, e_v = 0;
                (void)std::frexp(d->mx_bound, &e_top);
                (void)std::frexp(std::fabs(r.beta) * (4.0 + ie), &e_v);
                band[j] = std::min(kMxMaxBands - 1, std::max(0, (e_top - e_v) / kMxBandBits));
                n_bands = std::max(n_bands, band[j] + 1);
            }
        }
        if (e == hipSuccess && n_bands > 1) {
            std::vector<nps_row_desc> copy(d->m);
            for (int b = 0; b < n_bands && e == hipSuccess; ++b) {
                nps_scoredef::MxBand mb;
                bool any = b == 0;
                for (uint64_t j = 0; j < d->m; ++j) {
                    copy[j] = data[j];
                    if (band[j] != b && std::isfinite(data[j].beta)) copy[j].beta = 0.0;
                    if (band[j] == b && std::isfinite(data[j].beta) && data[j].beta != 0.0) {
                        const double ie = std::isfinite(data[j].eaf) ? std::max(2.0, 2.0 * std::fabs(data[j].eaf)) : 2.0;
                        mb.bound = std::max(mb.bound, std::fabs(data[j].beta) * (4.0 + ie));
                        any = true;
                    }
                }
                if (!any) continue;  // an empty band costs no pass
                e = hipMalloc(&mb.d_desc, sizeof(nps_row_desc) * d->m);
                if (e == hipSuccess)
                    e = hipMemcpy(mb.d_desc, copy.data(), sizeof(nps_row_desc) * d->m, hipMemcpyHostToDevice);
                d->mx_bands.push_back(mb);
            }
        }
        if (e != hipSuccess) {
            (void)hipFree(d->d_desc);
            for (auto &mb : d->mx_bands) (void)hipFree(mb.d_desc);
            delete d;
            return fail(e == hipErrorOutOfMemory ? NPS_E_NOMEM : NPS_E_HIP,
                        "uploading the score definition failed: %s", hipGetErrorString(e));
        }
    }
    *out = d;
    return NPS_OK;
}

extern "C" uint64_t nps_scoredef_n_present(const nps_scoredef *d) { return d ? d->m : 0; }

extern "C" void nps_scoredef_destroy(nps_scoredef *d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    (void)hipFree(d->d_desc);
    for (auto &mb : d->mx_bands) (void)hipFree(mb.d_desc);
    delete d;
}

// ------------------------------------------------------------------------------------------
// resident scoring.  Two-pass mode: per block of rows, tally -> params -> accumulate.
#ifdef NPS_DIAGNOSTICS
// run-time switches exist in diagnostics builds only (tools/mkexp.sh -DNPS_DIAGNOSTICS): the release
// library reads no environment variable on its launch path
static uint64_t env_u64(const char *name, uint64_t dflt) {
    const char *s = getenv(name);
    if (!s || !*s) return dflt;
    char *end = nullptr;
    unsigned long long v = strtoull(s, &end, 10);
    return end && *end == 0 ? (uint64_t)v : dflt;
}
#else
static inline uint64_t env_u64(const char *, uint64_t dflt) { return dflt; }
#endif

// stats of the last resident run stay on the device until somebody asks for them
static int materialize_resident_stats(nps_ctx *c) {
    if (!c->res_pending) return NPS_OK;
    std::vector<nps_locus_stat> dev(c->res_m);
    if (c->res_m) {
        HIP_TRY(hipMemcpyAsync(dev.data(), c->d_rstats, sizeof(nps_locus_stat) * c->res_m,
                               hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    size_t h = 0;
    for (size_t j = 0; j < c->res_index.size(); ++j) {
        if (c->res_index[j] < 0) {
            c->ready.push_back(c->res_host_stats[h++]);
            continue;
        }
        c->ready.push_back(dev[(size_t)c->res_index[j]]);
    }
    c->res_pending = false;
    c->res_index.clear();
    c->res_host_stats.clear();
    return NPS_OK;
}

// (re)allocation helper: *p holds at least `need` elements of `elem` bytes afterwards
// the reference's test `nmissing / N > --maxmis` (double division, nimpress.nim:565) is monotone in nmissing: the largest
// count that is NOT over the rate (-1: none), found with that very expression -- the kernels compare integers
static int64_t maxmis_threshold(uint64_t n, double rate) {
    if (n == 0 || (double)0 / (double)n > rate) return -1;
    uint64_t lo = 0, hi = n;  // pred(lo) holds
    while (lo < hi) {
        const uint64_t mid = lo + (hi - lo + 1) / 2;
        if (!((double)mid / (double)n > rate))
            lo = mid;
        else
            hi = mid - 1;
    }
    return (int64_t)lo;
}

static int grow(nps_ctx *c, void **p, uint64_t *cap, uint64_t need, size_t elem) {
    if (need <= *cap) return NPS_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    HIP_TRY(hipMalloc(p, elem * need));
    *cap = need;
    return NPS_OK;
}

static int ensure_resident_buffers(nps_ctx *c, uint64_t m_pad) {
    if (m_pad <= c->res_cap) return NPS_OK;
    HIP_TRY(hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_rtally);
    (void)hipFree(c->d_rlut);
    (void)hipFree(c->d_rstats);
    c->d_rtally = nullptr;
    c->d_rlut = nullptr;
    c->d_rstats = nullptr;
    c->res_cap = 0;
    c->rtally_clean = false;
    HIP_TRY(hipMalloc(&c->d_rtally, sizeof(unsigned long long) * m_pad));
    HIP_TRY(hipMalloc(&c->d_rlut, sizeof(double) * 4 * m_pad));
    HIP_TRY(hipMalloc(&c->d_rstats, sizeof(nps_locus_stat) * m_pad));
    c->res_cap = m_pad;
    return NPS_OK;
}

// Everything that can be refused is checked BEFORE the context changes: a call that returns an error
// from its validation leaves the context exactly as it was, so it can be repeated (e.g. in another
// mode).  The rows without genotype data are applied, and the run is registered for nps_flush, only
// once all launches have been queued; a failure between the first and the last launch marks the
// context broken (NPS_E_STATE until nps_reset).
extern "C" int nps_score_cohort_def(nps_ctx *c, const nps_cohort *co, uint64_t cohort_row0,
                                    const nps_scoredef *def, int mode) {
    int rc = check_usable(c);
    if (rc) return rc;
    if (!co || !def) return fail(NPS_E_INVAL, "cohort or scoredef is NULL");
    if (co->device != c->device || def->device != c->device)
        return fail(NPS_E_INVAL, "cohort / scoredef / context on different devices");
    if (co->n_samples != c->n)
        return fail(NPS_E_INVAL, "cohort has %llu samples, context %llu",
                    (unsigned long long)co->n_samples, (unsigned long long)c->n);
    if (mode != NPS_MODE_AUTO && mode != NPS_MODE_TWOPASS && mode != NPS_MODE_FUSED)
        return fail(NPS_E_INVAL, "bad mode %d", mode);
    const uint64_t m = def->m;
    rc = check_range(co, cohort_row0, m);
    if (rc) return rc;
    if (co->format == NPS_FMT_GT2M)
        return fail(NPS_E_UNSUPPORTED, "NPS_FMT_GT2M cohorts are scored with nps_score_cohort_multi");
    rc = cohort_quiesce(co);  // rows pushed into the cohort (nps_cohort_push_*) are complete
    if (rc) return rc;
    const bool is_ds16 = co->format == NPS_FMT_DS16;
    const bool is_ds = co->format == NPS_FMT_DS32 || is_ds16;
    const bool is_mx = co->format == NPS_FMT_GT2X;
    if (is_ds16 && mode == NPS_MODE_TWOPASS)
        return fail(NPS_E_UNSUPPORTED, "NPS_FMT_DS16 cohorts are scored by the single-read kernel only");
    if (is_mx && (cohort_row0 & 127))
        return fail(NPS_E_INVAL, "cohort_row0 must be a multiple of 128 for NPS_FMT_GT2X cohorts");
    if (!is_ds && (cohort_row0 & 3))
        return fail(NPS_E_INVAL, "cohort_row0 must be a multiple of 4 (rows are stored in groups of 4)");
    HIP_TRY(hipSetDevice(c->device));
    MxPlan mxp;
    bool kept_tallies = false, harvest = false;
    if (is_mx && m && c->n) {
        // a cohort that carries its tallies (nps_cohort_keep_tallies) is scored with them given under NPS_MODE_AUTO: the
        // "two-pass" plan (independent workgroups) without its tally pass
        kept_tallies = co->mx_row_tally_valid.load(std::memory_order_acquire) && mode == NPS_MODE_AUTO;
        if (!kept_tallies && mode == NPS_MODE_AUTO) {
            // Does the single-read kernel's resident grid cover the chip at this size, and will the cohort be scored again?
            //   * a resident grid exists (P <= compute units) and the run covers the whole cohort: the pass counts the tallies
            //     anyway -- where later passes want them given (the grid covers less than nine tenths of the chip, or the
            //     caller said nps_cohort_expect_passes >= 2) its epilogue KEEPS them with the cohort (`harvest`): the first
            //     run is one read, every later one runs with the tallies given (round 6; until then the first run of such
            //     a size was a tally pass + a given-tallies pass: two reads);
            //   * no resident grid (more strips than compute units), or a partial run of at least a quarter of the cohort:
            //     count the cohort's tallies once (one more read) and keep them (the cohort's own cache: rewriting rows
            //     drops it); shorter runs tally just their own rows (two reads of those rows).
            MxPlan p1;
            HIP_TRY(mx_plan(c->device, c->n, m, false, &p1));
            int cus = 0;
            HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
            // (nine tenths, by measurement: at 400 000 samples -- 196 strips, 77 % of the chip -- the in-pass kernel runs at
            //  0.64-0.67 of the roofline and the same cohort with its tallies given at 0.73-0.75; until round 6, when keeping
            //  the tallies still cost a pass of its own, the line was drawn at seven tenths)
            const bool covers = p1.ok && !p1.given && (uint64_t)p1.P * p1.Q * 10 >= (uint64_t)cus * 9;
            // (only where a strip has ONE row team -- more than 128 strips, 262 144 samples: with several teams per strip the
            //  given-tallies kernel is no faster than the pass that counts them -- 250 000 samples 11.3 against 11.4 ms,
            //  200 000 equal -- and slower below: 100 000 samples 5.05 against 4.27 ms, profiles/r06_harvest.txt)
            const bool one_team = p1.ok && (p1.given || p1.Q == 1);
            const bool want_kept = one_team && (!covers || co->expect_passes.load(std::memory_order_relaxed) >= 2);
            const bool whole = cohort_row0 == 0 && m == co->n_rows;
            if (p1.ok && want_kept) {
                if (!p1.given && whole && m >= 1024) {
                    harvest = true;
                } else if ((p1.given || (!covers && m >= 16384)) && m * 4 >= co->n_rows) {
                    nps_cohort *mco = const_cast<nps_cohort *>(co);
                    std::lock_guard<std::mutex> lk(mco->tally_mutex);
                    if (!mco->mx_row_tally_valid.load(std::memory_order_acquire)) {
                        rc = nps_cohort_keep_tallies(mco);
                        if (rc) return rc;
                        HIP_TRY(hipSetDevice(c->device));
                    }
                    kept_tallies = true;
                }
            }
        }
        const bool two_pass = mode == NPS_MODE_TWOPASS || kept_tallies;
        if (c->mx_plan_valid && c->mx_plan_m == m && c->mx_plan_two_pass == two_pass) {
            mxp = c->mx_plan_cache;
        } else {
            HIP_TRY(mx_plan(c->device, c->n, m, two_pass, &mxp));
            c->mx_plan_cache = mxp;
            c->mx_plan_m = m;
            c->mx_plan_two_pass = two_pass;
            c->mx_plan_valid = true;
        }
        if (!mxp.ok)
            return fail(NPS_E_UNSUPPORTED, "shape (%llu samples, %llu rows) is beyond the NPS_FMT_GT2X kernels (2^27 samples)",
                        (unsigned long long)c->n, (unsigned long long)m);
        // more strips than compute units: the single-read kernel cannot hold the grid resident; AUTO takes the
        // tally + accumulate pair (two reads), an explicit NPS_MODE_FUSED is refused
        if (mxp.given && mode == NPS_MODE_FUSED)
            return fail(NPS_E_UNSUPPORTED, "shape (%llu samples, %llu rows) does not fit the persistent grid of the "
                        "single-read NPS_FMT_GT2X kernel (one 2048-sample strip per compute unit); NPS_MODE_AUTO "
                        "scores it in two reads", (unsigned long long)c->n, (unsigned long long)m);
    }
    FusedPlan plan;
    if (!is_mx && mode != NPS_MODE_TWOPASS && m && c->n) {
        if (c->plan_valid && c->plan_fmt == co->format && c->plan_m == m) {
            plan = c->plan_cache;
        } else {
            const int want = (int)env_u64("NPS_FUSED_THREADS", 0), max_q = (int)env_u64("NPS_FUSED_MAXQ", 0);
            if (is_ds)
                HIP_TRY(ds_fused_plan(c->device, c->n, m, want, max_q, &plan, is_ds16 ? 2 : 4));
            else
                HIP_TRY(fused_plan(c->device, c->n, m, want, max_q, &plan));
            c->plan_cache = plan;
            c->plan_fmt = co->format;
            c->plan_m = m;
            c->plan_valid = true;
        }
        if (env_u64("NPS_DISABLE_FUSED", 0) == 1 && mode == NPS_MODE_AUTO) plan.ok = false;
        if (is_ds && co->ds_bad_rows) {  // a dosage outside [0, 2] somewhere in the cohort: the two-pass kernels take any value
            if (mode == NPS_MODE_FUSED)
                return fail(NPS_E_INVAL, "%llu row(s) of the cohort hold FORMAT/DS values outside [0, 2]: the single-read DS "
                            "kernel hands dosage sums over in fixed point and needs that range (NPS_MODE_AUTO / "
                            "NPS_MODE_TWOPASS score such a cohort in two reads)", (unsigned long long)co->ds_bad_rows);
            plan.ok = false;
        }
        if (!plan.ok && (mode == NPS_MODE_FUSED || is_ds16))
            return fail(NPS_E_UNSUPPORTED, "shape (%llu samples, %llu rows) does not fit the fused "
                        "persistent grid%s", (unsigned long long)c->n, (unsigned long long)m,
                        is_ds16 ? " (NPS_FMT_DS16 has no two-read kernels: use NPS_FMT_DS32)" : "");
    }

    rc = run_batch(c);  // keep push order: finish whatever was streamed before
    if (rc) return rc;
    rc = materialize_resident_stats(c);
    if (rc) return rc;

    // the run is registered only after its launches are queued (commit below)
    auto commit = [&]() {
        c->res_host_stats.clear();
        c->res_host_stats.reserve(def->host_rows.size());
        for (const nps_row_desc &r : def->host_rows) {  // host decided; nimpress.nim:526-558
            nps_locus_stat st;
            host_locus_row(c, r.kind, r.ref_is_effect, r.beta, r.eaf, &st);
            c->res_host_stats.push_back(st);
        }
        c->res_index = def->data_index;
        c->res_m = m;
        c->res_pending = true;
    };
    if (m == 0) {
        commit();
        return NPS_OK;
    }

    // buffers (a failed allocation leaves the context usable: nothing has been queued yet)
    const uint64_t m_pad = is_mx ? (m + 127) / 128 * 128 : (m + 15) / 16 * 16;
    const bool fused = plan.ok && c->n;
    // (the single-read DS kernel keeps a PAIR of tally words per row)
    const uint64_t n_tally = is_ds && fused ? 2 * m_pad : m_pad;
    rc = ensure_resident_buffers(c, n_tally);
    if (rc) return rc;
    if (is_mx && c->n) {
        rc = grow(c, (void **)&c->d_mx_cpart, &c->mx_cpart_cap, mxp.cpart_floats, sizeof(float));
        if (rc) return rc;
        if (8 + 2ull * mxp.Q > c->mx_const_cap) {
            rc = grow(c, (void **)&c->d_mx_const, &c->mx_const_cap, 8 + 2ull * mxp.Q, sizeof(double));
            if (rc) return rc;
            HIP_TRY(hipMemsetAsync(c->d_mx_const, 0, sizeof(double) * c->mx_const_cap, c->stream));
        }
        if (harvest) {  // (the cohort's kept tallies: allocated once, by whoever harvests first)
            nps_cohort *mco = const_cast<nps_cohort *>(co);
            std::lock_guard<std::mutex> lk(mco->tally_mutex);
            if (!mco->d_mx_row_tally) {
                if (hipMalloc(&mco->d_mx_row_tally, sizeof(unsigned long long) * gt2x_superblocks(co->n_rows) * 128) != hipSuccess) {
                    (void)hipGetLastError();
                    mco->d_mx_row_tally = nullptr;
                    harvest = false;  // (no room: the pass runs as it always did)
                }
            }
        }
        if (mxp.given) {
            rc = grow(c, (void **)&c->d_mx_ops, &c->mx_ops_cap, m_pad, 48);
            if (rc == NPS_OK) rc = grow(c, (void **)&c->d_mx_cblk, &c->mx_cblk_cap, m_pad / 128, sizeof(double));
            if (rc) return rc;
        }
        const uint64_t need1 = (uint64_t)((mxp.P + 15) / 16) * m_pad;
        if (need1 > c->mx_tally1_cap) {
            rc = grow(c, (void **)&c->d_mx_tally1, &c->mx_tally1_cap, need1, sizeof(unsigned long long));
            if (rc) return rc;
            HIP_TRY(hipMemsetAsync(c->d_mx_tally1, 0, sizeof(unsigned long long) * c->mx_tally1_cap, c->stream));
        }
    }
    if (fused) {
        rc = grow(c, (void **)&c->d_part_fused, &c->part_fused_cap,
                  (uint64_t)plan.Q * plan.part_team_stride, sizeof(double));
        if (rc) return rc;
    }
    if (is_ds) {
        if (m_pad > c->res_ds_cap) {
            uint64_t cap_a = c->res_ds_cap, cap_b = c->res_ds_cap;
            rc = grow(c, (void **)&c->d_rds_tally, &cap_a, m_pad, sizeof(DsTally));
            if (rc == NPS_OK) rc = grow(c, (void **)&c->d_rds_rowp, &cap_b, m_pad, sizeof(DsRowP));
            c->res_ds_cap = std::min(cap_a, cap_b);
            if (rc) return rc;
        }
    }

    // ---- launches.  From here on an error leaves queued work behind: the context is marked broken.
    struct Guard {
        nps_ctx *c;
        bool armed = false, ok = false;  // armed: a launch that changes the context's sums is queued
        ~Guard() { if (armed && !ok) c->broken = true; }
    } guard{c};
    auto done = [&]() {
        commit();
        guard.ok = true;
        return NPS_OK;
    };
    // the fused kernels' tally words must be zero on entry; their epilogue (fold_kernel) leaves them so
    auto tally_ready = [&]() -> int {
        if (!c->rtally_clean)
            HIP_TRY(hipMemsetAsync(c->d_rtally, 0, sizeof(unsigned long long) * c->res_cap, c->stream));
        c->rtally_clean = false;
        return NPS_OK;
    };
    auto epilogue = [&]() -> int {
        ProfScope ps(c, P_REDUCE);
        HIP_TRY(launch_fold(c->stream, c->d_part_fused, plan.Q, plan.part_team_stride, c->n, c->d_part,
                            c->chunks_used == 0 ? 1 : 0, c->d_rtally, n_tally, c->d_timeout,
                            c->d_nloci + 1));
        c->chunks_used = std::max(c->chunks_used, 1u);
        c->rtally_clean = true;  // all words were zero before the run, [0, m_pad) are zero again
        return NPS_OK;
    };

    if (is_mx) {
        if (c->n == 0) return done();
        const int64_t t_maxmis = maxmis_threshold(c->n, c->params.max_missing_rate);
        // one pass per magnitude band of the definition (normally one): band 0 counts nloci and writes the statistics
        struct Run {
            const nps_row_desc *d_desc;
            double bound;
        };
        std::vector<Run> runs;
        if (def->mx_bands.empty())
            runs.push_back(Run{def->d_desc, def->mx_bound});
        else
            for (const auto &mb : def->mx_bands) runs.push_back(Run{mb.d_desc, mb.bound});
        unsigned long long *scratch_nloci = reinterpret_cast<unsigned long long *>(c->d_mx_const);  // (the 8 scratch doubles)
        double *const_slots = c->d_mx_const + 8;
        for (size_t b = 0; b < runs.size(); ++b) {
            // fixed-point scale: every weight of the band below 2^56 (fourteen hexadecimal digits)
            int F = 56;
            if (runs[b].bound > 0.0) {
                int e2 = 0;
                (void)std::frexp(runs[b].bound, &e2);  // bound < 2^e2
                F = std::min(1000, std::max(-1000, 56 - e2));
            }
            rc = tally_ready();
            if (rc) return rc;
            if (mxp.given && !kept_tallies) {
                ProfScope ps(c, P_TALLY);
                HIP_TRY(launch_mx_tally(c->stream, mxp, co->d_data, gt2x_superblocks(co->n_rows), cohort_row0 >> 7, c->n,
                                        c->d_rtally));
            }
            hipError_t fe;
            {
                ProfScope ps(c, mxp.given ? P_ACCUM : P_FUSED);
                if (mxp.given)
                    fe = launch_mx_given(c->stream, mxp, co->d_data, gt2x_superblocks(co->n_rows), cohort_row0 >> 7, c->n, m,
                                         runs[b].d_desc, dev_params(c->params), t_maxmis, F,
                                         kept_tallies ? co->d_mx_row_tally + cohort_row0 : c->d_rtally,
                                         b == 0 ? c->d_rstats : nullptr, b == 0 ? c->d_nloci : scratch_nloci, const_slots,
                                         c->d_mx_cpart, c->d_mx_ops, c->d_mx_cblk, c->d_timeout + 17, c->d_timeout);
                else
                    fe = launch_fused_mx(c->stream, mxp, co->d_data, gt2x_superblocks(co->n_rows), cohort_row0 >> 7, c->n, m,
                                         runs[b].d_desc, dev_params(c->params), t_maxmis, F, c->d_rlut,
                                         kept_tallies ? co->d_mx_row_tally + cohort_row0 : c->d_rtally, c->d_mx_tally1,
                                         b == 0 ? c->d_rstats : nullptr, b == 0 ? c->d_nloci : scratch_nloci,
                                         const_slots, c->d_mx_cpart, c->d_timeout);
            }
            if (fe != hipSuccess) {
                (void)hipGetLastError();  // the runtime refused the cooperative grid: nothing ran
                c->rtally_clean = !mxp.given || kept_tallies;
                return fail(NPS_E_HIP, "NPS_FMT_GT2X kernel launch failed: %s", hipGetErrorString(fe));
            }
            guard.armed = true;
            {
                ProfScope ps(c, P_REDUCE);
                HIP_TRY(launch_mx_fold(c->stream, mxp, c->d_mx_cpart, c->n, F, const_slots, c->d_part,
                                       c->chunks_used == 0 ? 1 : 0, c->d_rtally, m_pad, c->d_mx_tally1,
                                       (uint64_t)((mxp.P + 15) / 16) * m_pad, c->d_timeout, c->d_nloci + 1,
                                       !mxp.given && mxp.U < 64 /* launch_fused_mx cut its own strips */,
                                       harvest && b == 0 && !mxp.given ? co->d_mx_row_tally : nullptr,
                                       harvest && b == 0 && !mxp.given ? m_pad : 0));
                HIP_TRY(hipMemsetAsync(const_slots, 0, sizeof(double) * 2 * mxp.Q, c->stream));
            }
            c->chunks_used = std::max(c->chunks_used, 1u);
            c->rtally_clean = true;
        }
        if (harvest && !mxp.given) {
            // the kept tallies are published only once they ARE in device memory (another context may score this cohort
            // from another thread and stream): one wait, on the cohort's first pass only
            HIP_TRY(hipStreamSynchronize(c->stream));
            const_cast<nps_cohort *>(co)->mx_row_tally_valid.store(true, std::memory_order_release);
        }
        return done();
    }

    if (is_ds) {
        const uint64_t stride_f = co->stride_bytes / 4;  // (NPS_FMT_DS32 only below the fused branch)
        const float *ds = (const float *)((const char *)co->d_data + cohort_row0 * co->stride_bytes);
        if (fused) {
            rc = tally_ready();
            if (rc) return rc;
            hipError_t fe;
            {
                ProfScope ps(c, P_FUSED);
                fe = launch_ds_fused(c->stream, plan, ds, co->stride_bytes, is_ds16 ? 2 : 4, c->n, m, def->d_desc,
                                     dev_params(c->params), maxmis_threshold(c->n, c->params.max_missing_rate), c->d_rtally, c->d_rstats,
                                     c->d_nloci, c->d_part_fused, c->d_timeout);
            }
            if (fe == hipSuccess) {
                guard.armed = true;
                rc = epilogue();
                if (rc) return rc;
                return done();
            }
            (void)hipGetLastError();  // the runtime refused the cooperative grid: nothing ran
            c->rtally_clean = true;   // (zeroed above, untouched)
            if (mode == NPS_MODE_FUSED || is_ds16 || fe != hipErrorCooperativeLaunchTooLarge)
                return fail(NPS_E_HIP, "fused DS kernel launch failed: %s", hipGetErrorString(fe));
        }
        if (is_ds16) return fail(NPS_E_UNSUPPORTED, "NPS_FMT_DS16 cohorts are scored by the single-read kernel only");
        // launches of >= 2048 rows keep every CU busy in both kernels (one workgroup per row in the
        // tally; samples x row chunks in the accumulation)
        uint64_t block_rows = env_u64("NPS_BLOCK_ROWS", 0);
        if (block_rows == 0) block_rows = std::max<uint64_t>(2048, (256ull << 20) / co->stride_bytes);
        rc = ensure_all_chunks(c);
        if (rc) return rc;
        guard.armed = true;
        for (uint64_t r0 = 0; r0 < m; r0 += block_rows) {
            const uint64_t k = std::min(block_rows, m - r0);
            {
                ProfScope ps(c, P_TALLY);
                HIP_TRY(launch_ds_tally(c->stream, ds + r0 * stride_f, stride_f, c->n, def->d_desc + r0,
                                        k, c->d_rds_tally + r0));
            }
            {
                ProfScope ps(c, P_PARAMS);
                HIP_TRY(launch_ds_params(c->stream, c->d_rds_tally + r0, def->d_desc + r0, k, c->n,
                                         dev_params(c->params), c->d_rds_rowp + r0, c->d_rstats + r0,
                                         c->d_nloci));
            }
            {
                ProfScope ps(c, P_ACCUM);
                HIP_TRY(launch_ds_accumulate(c->stream, ds + r0 * stride_f, stride_f, c->n,
                                             c->d_rds_rowp + r0, k, c->d_part, c->n_chunks,
                                             c->geom.part_chunk_stride));
            }
        }
        return done();
    }

    const uint64_t stride_words = co->stride_bytes / 4;
    const uint32_t *codes = (const uint32_t *)co->d_data + (cohort_row0 >> 2) * stride_words * 4;
    const nps_row_desc *d_desc = def->d_desc;
    const int parity = co->optimized ? 1 : 0;  // the kernels undo the parity layout for the tally
    if (fused) {
        rc = tally_ready();
        if (rc) return rc;
        hipError_t fe;
        {
            ProfScope ps(c, P_FUSED);
            fe = launch_fused(c->stream, plan, codes, stride_words, c->n, m, d_desc,
                              dev_params(c->params), c->d_rtally, c->d_rstats, c->d_nloci,
                              c->d_part_fused, c->d_timeout, parity);
        }
        if (fe == hipSuccess) {
            guard.armed = true;
            rc = epilogue();
            if (rc) return rc;
#ifdef NPS_DIAGNOSTICS
            if (getenv("NPS_TELEMETRY")) {  // diagnostics of the control wave (cycles, summed)
                unsigned long long t[8] = {0};
                (void)hipStreamSynchronize(c->stream);
                (void)hipMemcpy(t, (char *)c->d_timeout + 16, sizeof t, hipMemcpyDeviceToHost);
                (void)hipMemset((char *)c->d_timeout + 16, 0, sizeof t);
                const double wg = (double)plan.P * plan.Q, st = t[4] ? (double)t[4] : 1.0;
                fprintf(stderr, "[nps] fused P=%u Q=%u T=%u: per step per WG: spins %.2f, poll wait "
                        "%.0f cyc, control chain %.0f cyc, barrier wait %.0f cyc (steps/WG %.0f)\n",
                        plan.P, plan.Q, plan.threads, t[0] / st, t[1] / st, t[2] / st, t[3] / st, st / wg);
            }
#endif
            return done();
        }
        // the runtime refused the cooperative grid (it would not be fully resident): nothing ran
        (void)hipGetLastError();
        c->rtally_clean = true;
        if (mode == NPS_MODE_FUSED || fe != hipErrorCooperativeLaunchTooLarge)
            return fail(NPS_E_HIP, "fused kernel launch failed: %s", hipGetErrorString(fe));
    }
    // rows per tally/accumulate pair; default ~96 MB so the second read can hit the 256 MB
    // Infinity Cache
    uint64_t block_rows = env_u64("NPS_BLOCK_ROWS", 0);
    if (block_rows == 0) block_rows = std::max<uint64_t>(64, (96ull << 20) / co->stride_bytes);
    block_rows = (block_rows + 15) / 16 * 16;
    rc = ensure_all_chunks(c);
    if (rc) return rc;
    guard.armed = true;
    c->rtally_clean = false;  // the two-pass tally kernel leaves its counts in d_rtally
    for (uint64_t r0 = 0; r0 < m; r0 += block_rows) {
        const uint64_t k = std::min(block_rows, m - r0);
        const uint64_t k_pad = (k + 3) / 4 * 4;  // only the last block can be ragged
        {
            ProfScope ps(c, P_TALLY);
            HIP_TRY(launch_tally_packed(c->stream, codes + (r0 >> 2) * stride_words * 4, stride_words,
                                        c->n, k, c->d_rtally + r0, parity));
        }
        {
            ProfScope ps(c, P_PARAMS);
            HIP_TRY(launch_row_params(c->stream, c->d_rtally + r0, d_desc + r0, k, k_pad, c->n,
                                      dev_params(c->params), c->d_rlut + r0 * 4, c->d_rstats + r0,
                                      c->d_nloci));
        }
        if (c->n) {
            AccumGeom g = c->geom;
            const uint32_t groups = (uint32_t)(k_pad / 4);
            g.groups_per_chunk = std::max(1u, (groups + g.n_chunks - 1) / g.n_chunks);
            ProfScope ps(c, P_ACCUM);
            HIP_TRY(launch_accumulate(c->stream, codes + (r0 >> 2) * stride_words * 4, stride_words, k,
                                      c->d_rlut + r0 * 4, g, c->d_part, parity));
        }
    }
    return done();
}

extern "C" int nps_score_cohort(nps_ctx *c, const nps_cohort *co, uint64_t cohort_row0,
                                const nps_row_desc *rows, uint64_t n_desc, int mode) {
    if (!c || !co) return fail(NPS_E_INVAL, "ctx or cohort is NULL");
    nps_scoredef *def = nullptr;
    int rc = nps_scoredef_create(&def, c->device, rows, n_desc);
    if (rc) return rc;
    rc = nps_score_cohort_def(c, co, cohort_row0, def, mode);
    // the launches read def->d_desc: complete them before the temporary definition goes away
    if (rc == NPS_OK) {
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(NPS_E_HIP, "resident scoring failed: %s", hipGetErrorString(e));
    }
    nps_scoredef_destroy(def);
    return rc;
}

// ------------------------------------------------------------------------------------------
// several score definitions in one pass over a NPS_FMT_GT2M cohort (nps_multi.hip)
extern "C" int nps_cohort_convert(nps_cohort *dst, const nps_cohort *src) {
    if (!dst || !src) return fail(NPS_E_INVAL, "cohort is NULL");
    if ((dst->format != NPS_FMT_GT2M && dst->format != NPS_FMT_GT2X) || src->format != NPS_FMT_GT2)
        return fail(NPS_E_INVAL, "nps_cohort_convert goes from a NPS_FMT_GT2 to a NPS_FMT_GT2M or NPS_FMT_GT2X cohort");
    if (dst->device != src->device || dst->n_samples != src->n_samples || dst->n_rows != src->n_rows)
        return fail(NPS_E_INVAL, "source and destination differ in device, samples or rows");
    if (src->optimized)
        return fail(NPS_E_STATE, "the source cohort is in the nps_cohort_optimize layout; convert it before optimising");
    HIP_TRY(hipSetDevice(dst->device));
    { int qrc = cohort_quiesce(src); if (qrc) return qrc; }
    HIP_TRY(hipDeviceSynchronize());
    if (src->n_rows == 0 || src->n_samples == 0) return NPS_OK;
    dst->mx_row_tally_valid = false;
    if (dst->format == NPS_FMT_GT2X) {
        HIP_TRY(launch_gt2_to_gt2x(nullptr, (const uint32_t *)src->d_data, src->stride_bytes / 4, src->n_samples,
                                   src->n_rows, dst->d_data));
        HIP_TRY(hipDeviceSynchronize());
        return NPS_OK;
    }
    // units and, from the same tiles, the whole-row tallies (tallyAlleles nimpress.nim:32-47)
    HIP_TRY(launch_convert_gt2m(nullptr, (const uint32_t *)src->d_data, src->stride_bytes / 4, src->n_samples,
                                src->n_rows, dst->d_data, dst->d_row_tally));
    HIP_TRY(hipDeviceSynchronize());
    return NPS_OK;
}

extern "C" int nps_cohort_row_tallies(const nps_cohort *c, uint64_t row0, uint64_t nrows, uint64_t *nmissing_out,
                                      uint64_t *neffect_out) {
    int rc = check_range(c, row0, nrows);
    if (rc) return rc;
    const bool kept = c->format == NPS_FMT_GT2X && c->mx_row_tally_valid;
    if (c->format != NPS_FMT_GT2M && !kept)
        return fail(NPS_E_UNSUPPORTED, "row tallies are kept with NPS_FMT_GT2M cohorts, and with NPS_FMT_GT2X cohorts after "
                                       "nps_cohort_keep_tallies");
    if (nrows == 0) return NPS_OK;
    HIP_TRY(hipSetDevice(c->device));
    std::vector<unsigned long long> t(nrows);
    HIP_TRY(hipMemcpy(t.data(), (kept ? c->d_mx_row_tally : c->d_row_tally) + row0, sizeof(unsigned long long) * nrows,
                      hipMemcpyDeviceToHost));
    for (uint64_t j = 0; j < nrows; ++j) {
        if (nmissing_out) nmissing_out[j] = kept ? (t[j] >> 28) & 0xfffffffull : t[j] >> 32;
        if (neffect_out) neffect_out[j] = kept ? t[j] & 0xfffffffull : t[j] & 0xffffffffull;
    }
    return NPS_OK;
}

// Experiment B of the round-4 verdict: a NPS_FMT_GT2X cohort that carries its whole-row tallies (tallyAlleles,
// nimpress.nim:32-47, of every row over all samples), counted ONCE by mx_tally_kernel.  nps_score_cohort[_def] with
// NPS_MODE_AUTO then scores the cohort with the tallies given: one read of the matrix, no popcounts, no hand-over
// between the strips, an ordinary grid.  For many score files over one cohort (BASELINE configs[3]: tally once, score
// eight times).  Rewriting rows (upload, synth, convert) drops the tallies.
extern "C" int nps_cohort_keep_tallies(nps_cohort *c) {
    if (!c) return fail(NPS_E_INVAL, "cohort is NULL");
    if (c->format != NPS_FMT_GT2X)
        return fail(NPS_E_UNSUPPORTED, "nps_cohort_keep_tallies is for NPS_FMT_GT2X cohorts (NPS_FMT_GT2M carries its tallies "
                                       "from the packer; the other layouts count while they read)");
    if (c->n_rows == 0 || c->n_samples == 0) {
        c->mx_row_tally_valid = true;
        return NPS_OK;
    }
    if (c->n_samples >= (1ull << 27)) return fail(NPS_E_UNSUPPORTED, "more than 2^27 samples");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());
    const uint64_t n_sb = gt2x_superblocks(c->n_rows);
    if (!c->d_mx_row_tally) HIP_TRY(hipMalloc(&c->d_mx_row_tally, sizeof(unsigned long long) * n_sb * 128));
    HIP_TRY(hipMemset(c->d_mx_row_tally, 0, sizeof(unsigned long long) * n_sb * 128));
    MxPlan mp;
    HIP_TRY(mx_plan(c->device, c->n_samples, c->n_rows, true, &mp));
    if (!mp.ok) return fail(NPS_E_UNSUPPORTED, "shape beyond the NPS_FMT_GT2X kernels");
    HIP_TRY(launch_mx_tally(nullptr, mp, c->d_data, n_sb, 0, c->n_samples, c->d_mx_row_tally));
    HIP_TRY(hipDeviceSynchronize());
    c->mx_row_tally_valid = true;
    return NPS_OK;
}
extern "C" int nps_cohort_expect_passes(nps_cohort *c, uint32_t n_passes) {
    if (!c) return fail(NPS_E_INVAL, "cohort is NULL");
    c->expect_passes.store(n_passes, std::memory_order_relaxed);
    return NPS_OK;
}
extern "C" int nps_cohort_has_tallies(const nps_cohort *c) {
    return c && (c->format == NPS_FMT_GT2M || (c->format == NPS_FMT_GT2X && c->mx_row_tally_valid)) ? 1 : 0;
}

struct nps_multidef {
    int device = 0;
    int S = 0;
    uint64_t n_desc = 0;
    nps_row_desc *d_desc = nullptr;  // [S][n_desc]
    int *d_F = nullptr;              // fixed-point exponent per score: weight * 2^F is an integer below 2^(8 ND - 9)
    int ND = 7;                      // base-256 digits per weight
};

extern "C" int nps_multidef_create_bits(nps_multidef **out, int device, const nps_row_desc *rows, int n_scores,
                                        uint64_t n_desc, int weight_bits) {
    if (!out) return fail(NPS_E_INVAL, "out is NULL");
    *out = nullptr;
    if (weight_bits != 0 && weight_bits != 41 && weight_bits != 49)
        return fail(NPS_E_INVAL, "weight_bits must be 41 or 49 (0 = default 49), not %d", weight_bits);
    const int ND = weight_bits == 41 ? 6 : 7;
    if (n_scores < 1 || n_scores > NPS_MULTI_MAX_SCORES)
        return fail(NPS_E_INVAL, "n_scores %d outside 1..%d", n_scores, NPS_MULTI_MAX_SCORES);
    if (n_desc && !rows) return fail(NPS_E_INVAL, "rows is NULL");
    if (n_desc > 0xfffffff0ull) return fail(NPS_E_UNSUPPORTED, "too many rows");
    int F[NPS_MULTI_MAX_SCORES];
    for (int s = 0; s < n_scores; ++s) {
        double maxb = 0.0, maxe = 0.0, minb = HUGE_VAL;
        for (uint64_t j = 0; j < n_desc; ++j) {
            const nps_row_desc &r = rows[(uint64_t)s * n_desc + j];
            if (r.kind < NPS_ROW_PRESENT || r.kind > NPS_ROW_NOT_IN_SCORE)
                return fail(NPS_E_INVAL, "score %d row %llu: bad kind %d", s, (unsigned long long)j, r.kind);
            if (r.kind == NPS_ROW_NOT_IN_SCORE) continue;
            if (!std::isfinite(r.beta))
                return fail(NPS_E_UNSUPPORTED, "score %d row %llu: beta is not finite (use the single-score path)",
                            s, (unsigned long long)j);
            maxb = std::max(maxb, std::fabs(r.beta));
            if (r.beta != 0.0) minb = std::min(minb, std::fabs(r.beta));
            if (std::isfinite(r.eaf)) maxe = std::max(maxe, std::fabs(r.eaf));
        }
        // One fixed-point scale per score: a weight is rounded to 2^-(8 ND - 9) of the largest one (x 5 for the imputation
        // range), so a sample that carries only the score's SMALL-beta rows keeps 1e-6 relative only while
        // 2.5 max|beta| / min|beta| <= 1e-6 x 2^(8 ND - 9): 2^25 for seven digits, 2^17 for six.  Beyond that the
        // definition is refused here -- the single-score path (nps_scoredef_create + nps_score_cohort_def) scores such a
        // definition in magnitude bands -- instead of silently losing those samples.
        const double span_limit = std::ldexp(1.0, ND == 7 ? 25 : 17);
        if (maxb > 0.0 && maxb / minb > span_limit)
            return fail(NPS_E_UNSUPPORTED, "score %d: |beta| spans %.3g (%.3g .. %.3g), more than the 2^%d one %d-bit "
                        "fixed-point scale holds within 1e-6 relative; score this definition with the single-score path "
                        "(nps_scoredef_create / nps_score_cohort_def: magnitude bands), the others in one pass",
                        s, maxb / minb, minb, maxb, ND == 7 ? 25 : 17, 8 * ND - 7);
        // largest weight a row can have: |beta| * max(|imputed - 3|, |locus constant|)
        const double bound = maxb * (3.0 + std::max(2.0, 2.0 * maxe));
        int e = 0;
        if (bound > 0.0) (void)std::frexp(bound, &e);  // bound < 2^e
        F[s] = 8 * ND - 9 - e;  // the kernel's coefficients (up to 160 x weight) stay below 2^(8 ND - 1)
    }
    int rc = select_device(device);
    if (rc) return rc;
    nps_multidef *d = new (std::nothrow) nps_multidef;
    if (!d) return fail(NPS_E_NOMEM, "out of host memory");
    d->device = device;
    d->S = n_scores;
    d->ND = ND;
    d->n_desc = n_desc;
    hipError_t e = hipMalloc(&d->d_F, sizeof(int) * NPS_MULTI_MAX_SCORES);
    if (e == hipSuccess) e = hipMemcpy(d->d_F, F, sizeof(int) * n_scores, hipMemcpyHostToDevice);
    if (e == hipSuccess && n_desc) {
        e = hipMalloc(&d->d_desc, sizeof(nps_row_desc) * n_desc * n_scores);
        if (e == hipSuccess)
            e = hipMemcpy(d->d_desc, rows, sizeof(nps_row_desc) * n_desc * n_scores, hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        (void)hipFree(d->d_F);
        (void)hipFree(d->d_desc);
        delete d;
        return fail(e == hipErrorOutOfMemory ? NPS_E_NOMEM : NPS_E_HIP, "uploading the score definitions failed: %s",
                    hipGetErrorString(e));
    }
    *out = d;
    return NPS_OK;
}

extern "C" int nps_multidef_create(nps_multidef **out, int device, const nps_row_desc *rows, int n_scores,
                                   uint64_t n_desc) {
    return nps_multidef_create_bits(out, device, rows, n_scores, n_desc, 0);
}

extern "C" void nps_multidef_destroy(nps_multidef *d) {
    if (!d) return;
    (void)hipSetDevice(d->device);
    (void)hipDeviceSynchronize();
    (void)hipFree(d->d_desc);
    (void)hipFree(d->d_F);
    delete d;
}

struct nps_multi {
    int device = 0, S = 0, cus = 0;
    hipStream_t stream = nullptr;
    uint64_t n = 0;
    nps_params params{};
    void *d_state = nullptr;      // MultiState[S]
    double *d_part = nullptr;     // [S][n] float64 running sums
    bool have_sums = false;
    void *d_table = nullptr;
    uint64_t table_cap = 0;
    int32_t *d_partial = nullptr;
    uint64_t partial_cap = 0;
    double *d_offsets = nullptr, *d_scores = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    double ms[3] = {0.0, 0.0, 0.0};  // params, product, fold of all calls since the last reset
    bool timed = false;
    int coarse_missing = 0;          // nps_multi_set_missing_weight_bits: leading base-256 digits kept (4 = 32 bits, 5 = 40; 0 = all)
    bool broken = false;             // a HIP call failed between the first and the last launch of a pass
};

// add the device time of the last call (if its events have not been read yet) to the running totals
static void multi_drain_timing(nps_multi *m) {
    if (!m->timed) return;
    float a = 0.f, b = 0.f, c = 0.f;
    if (hipEventSynchronize(m->ev[3]) == hipSuccess) {
        (void)hipEventElapsedTime(&a, m->ev[0], m->ev[1]);
        (void)hipEventElapsedTime(&b, m->ev[1], m->ev[2]);
        (void)hipEventElapsedTime(&c, m->ev[2], m->ev[3]);
        m->ms[0] += a;
        m->ms[1] += b;
        m->ms[2] += c;
    }
    m->timed = false;
}

static void free_multi(nps_multi *m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->stream) (void)hipStreamSynchronize(m->stream);
    (void)hipFree(m->d_state);
    (void)hipFree(m->d_part);
    (void)hipFree(m->d_table);
    (void)hipFree(m->d_partial);
    (void)hipFree(m->d_offsets);
    (void)hipFree(m->d_scores);
    for (hipEvent_t e : m->ev)
        if (e) (void)hipEventDestroy(e);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}

extern "C" int nps_multi_create(nps_multi **out, int device, uint64_t n_samples, const nps_params *params,
                                int n_scores) {
    if (!out) return fail(NPS_E_INVAL, "out is NULL");
    *out = nullptr;
    int rc = check_params(params);
    if (rc) return rc;
    if (n_scores < 1 || n_scores > NPS_MULTI_MAX_SCORES)
        return fail(NPS_E_INVAL, "n_scores %d outside 1..%d", n_scores, NPS_MULTI_MAX_SCORES);
    if (n_samples > 0x7fffffffull) return fail(NPS_E_UNSUPPORTED, "n_samples too large");
    rc = select_device(device);
    if (rc) return rc;
    nps_multi *m = new (std::nothrow) nps_multi;
    if (!m) return fail(NPS_E_NOMEM, "out of host memory");
    m->device = device;
    m->S = n_scores;
    m->n = n_samples;
    m->params = *params;
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    m->cus = e == hipSuccess ? prop.multiProcessorCount : 256;
    const size_t sb = multi_state_bytes() * NPS_MULTI_MAX_SCORES;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(&m->d_state, sb);
    if (e == hipSuccess) e = hipMemset(m->d_state, 0, sb);
    if (e == hipSuccess) e = hipMalloc(&m->d_part, sizeof(double) * std::max<uint64_t>(n_samples, 1) * n_scores);
    if (e == hipSuccess) e = hipMalloc(&m->d_scores, sizeof(double) * std::max<uint64_t>(n_samples, 1) * n_scores);
    if (e == hipSuccess) e = hipMalloc(&m->d_offsets, sizeof(double) * NPS_MULTI_MAX_SCORES);
    for (int k = 0; k < 4 && e == hipSuccess; ++k) e = hipEventCreate(&m->ev[k]);
    if (e != hipSuccess) {
        free_multi(m);
        return fail(e == hipErrorOutOfMemory ? NPS_E_NOMEM : NPS_E_HIP, "nps_multi_create: %s", hipGetErrorString(e));
    }
    *out = m;
    return NPS_OK;
}

extern "C" void nps_multi_destroy(nps_multi *m) { free_multi(m); }

extern "C" int nps_multi_set_missing_weight_bits(nps_multi *m, int bits) {
    if (!m) return fail(NPS_E_INVAL, "ctx is NULL");
    if (bits != 0 && bits != 32 && bits != 40 && bits != 56)
        return fail(NPS_E_INVAL, "bits must be 32, 40 or 56 (0 = default 56), not %d", bits);
    m->coarse_missing = bits == 32 ? 4 : bits == 40 ? 5 : 0;
    return NPS_OK;
}

extern "C" int nps_multi_reset(nps_multi *m, const nps_params *params) {
    if (!m) return fail(NPS_E_INVAL, "ctx is NULL");
    if (params) {
        int rc = check_params(params);
        if (rc) return rc;
        m->params = *params;
    }
    HIP_TRY(hipSetDevice(m->device));
    multi_drain_timing(m);
    HIP_TRY(hipMemsetAsync(m->d_state, 0, multi_state_bytes() * NPS_MULTI_MAX_SCORES, m->stream));
    m->broken = false;
    m->have_sums = false;
    m->ms[0] = m->ms[1] = m->ms[2] = 0.0;
    return NPS_OK;
}

extern "C" int nps_score_cohort_multi(nps_multi *m, const nps_cohort *co, uint64_t cohort_row0,
                                      const nps_multidef *def) {
    if (!m || !co || !def) return fail(NPS_E_INVAL, "ctx, cohort or definitions is NULL");
    if (m->broken) return fail(NPS_E_STATE, "an earlier pass failed on the device: nps_multi_reset first");
    if (co->format != NPS_FMT_GT2M) return fail(NPS_E_INVAL, "nps_score_cohort_multi needs a NPS_FMT_GT2M cohort");
    if (co->device != m->device || def->device != m->device)
        return fail(NPS_E_INVAL, "cohort / definitions / context on different devices");
    if (co->n_samples != m->n)
        return fail(NPS_E_INVAL, "cohort has %llu samples, context %llu", (unsigned long long)co->n_samples,
                    (unsigned long long)m->n);
    if (def->S != m->S) return fail(NPS_E_INVAL, "definitions hold %d scores, context %d", def->S, m->S);
    if (cohort_row0 & 127) return fail(NPS_E_INVAL, "cohort_row0 must be a multiple of 128");
    int rc = check_range(co, cohort_row0, def->n_desc);
    if (rc) return rc;
    if (def->n_desc == 0) return NPS_OK;
    HIP_TRY(hipSetDevice(m->device));
    const MultiPlan pl = multi_plan(m->n, def->n_desc, m->S, def->ND, m->coarse_missing, m->cus);
    if (pl.table_bytes() + pl.flag_bytes() > m->table_cap) {
        HIP_TRY(hipStreamSynchronize(m->stream));
        (void)hipFree(m->d_table);
        m->d_table = nullptr;
        m->table_cap = 0;
        HIP_TRY(hipMalloc(&m->d_table, pl.table_bytes() + pl.flag_bytes()));
        m->table_cap = pl.table_bytes() + pl.flag_bytes();
    }
    if (pl.partial_elems() > m->partial_cap) {
        HIP_TRY(hipStreamSynchronize(m->stream));
        (void)hipFree(m->d_partial);
        m->d_partial = nullptr;
        m->partial_cap = 0;
        HIP_TRY(hipMalloc(&m->d_partial, sizeof(int32_t) * std::max<uint64_t>(pl.partial_elems(), 1)));
        m->partial_cap = pl.partial_elems();
    }
    // (the timing events are about to be re-recorded: read the previous pass's times only if it has completed already --
    //  a caller that queues pass after pass is never blocked here; its per-pass times are then not accumulated)
    if (m->timed && hipEventQuery(m->ev[3]) == hipSuccess) multi_drain_timing(m);
    m->timed = false;
    // Everything that can be refused has been checked and allocated.  From here a HIP failure leaves the running
    // sums and counts of the context undefined: it is marked broken (NPS_E_STATE until nps_multi_reset).
    // (multi_params_kernel writes every fragment of the table, zeros for unused columns and padding rows)
    auto run = [&]() -> hipError_t {
        hipError_t e = hipEventRecord(m->ev[0], m->stream);
        if (e != hipSuccess) return e;
        e = launch_multi_params(m->stream, co->d_row_tally + cohort_row0, def->d_desc, def->n_desc, m->S, pl, m->n,
                                dev_params(m->params), def->d_F, m->d_table, m->d_state, m->coarse_missing);
        if (e != hipSuccess) return e;
        e = hipEventRecord(m->ev[1], m->stream);
        if (e != hipSuccess) return e;
        if (m->n) {
            e = launch_multi_mfma(m->stream, pl, co->d_data, cohort_row0 / 128, m->d_table, m->d_partial, m->d_state,
                                  co->d_row_tally + cohort_row0, def->n_desc,
                                  reinterpret_cast<uint32_t *>(static_cast<char *>(m->d_table) + pl.table_bytes()));
            if (e != hipSuccess) return e;
        }
        e = hipEventRecord(m->ev[2], m->stream);
        if (e != hipSuccess) return e;
        e = launch_multi_fold(m->stream, pl, m->d_partial, m->n, m->S, def->d_F, m->d_part, m->have_sums ? 0 : 1,
                              m->d_state);
        if (e != hipSuccess) return e;
        return hipEventRecord(m->ev[3], m->stream);
    };
    const hipError_t e = run();
    if (e != hipSuccess) {
        m->broken = true;
        return fail(NPS_E_HIP, "nps_score_cohort_multi: %s (context needs nps_multi_reset)", hipGetErrorString(e));
    }
    m->have_sums = true;
    m->timed = true;
    return NPS_OK;
}

static int multi_finish_common(nps_multi *m, const double *offsets, double *d_dst, double *h_scores_out,
                               uint64_t *nloci_out, int normalise = 1) {
    if (!offsets && normalise) return fail(NPS_E_INVAL, "offsets is NULL");
    if (m->broken) return fail(NPS_E_STATE, "an earlier pass failed on the device: nps_multi_reset first");
    HIP_TRY(hipSetDevice(m->device));
    if (normalise)
        HIP_TRY(hipMemcpyAsync(m->d_offsets, offsets, sizeof(double) * m->S, hipMemcpyHostToDevice, m->stream));
    HIP_TRY(launch_multi_finish(m->stream, m->d_part, m->n, m->S, m->d_state, m->d_offsets, m->have_sums ? 1 : 0,
                                d_dst, normalise));
    std::vector<char> st(multi_state_bytes() * m->S);
    HIP_TRY(hipMemcpyAsync(st.data(), m->d_state, st.size(), hipMemcpyDeviceToHost, m->stream));
    if (h_scores_out && m->n)
        HIP_TRY(hipMemcpyAsync(h_scores_out, d_dst, sizeof(double) * m->n * m->S, hipMemcpyDeviceToHost, m->stream));
    HIP_TRY(hipStreamSynchronize(m->stream));
    if (nloci_out)
        for (int s = 0; s < m->S; ++s)
            memcpy(&nloci_out[s], st.data() + multi_state_bytes() * s, sizeof(uint64_t));  // first field
    return NPS_OK;
}

extern "C" int nps_multi_finish(nps_multi *m, const double *offsets, double *scores_out, uint64_t *nloci_out) {
    if (!m) return fail(NPS_E_INVAL, "ctx is NULL");
    if (m->n && !scores_out) return fail(NPS_E_INVAL, "scores_out is NULL");
    return multi_finish_common(m, offsets, m->d_scores, scores_out, nloci_out);
}

extern "C" int nps_multi_finish_device(nps_multi *m, const double *offsets, double *d_scores_out,
                                       uint64_t *nloci_out) {
    if (!m) return fail(NPS_E_INVAL, "ctx is NULL");
    if (m->n && !d_scores_out) return fail(NPS_E_INVAL, "d_scores_out is NULL");
    return multi_finish_common(m, offsets, d_scores_out, nullptr, nloci_out);
}

extern "C" int nps_multi_n_scores(const nps_multi *m) { return m ? m->S : 0; }
extern "C" uint64_t nps_multi_n_samples(const nps_multi *m) { return m ? m->n : 0; }
extern "C" int nps_multi_device(const nps_multi *m) { return m ? m->device : -1; }

extern "C" int nps_multi_partial_device(nps_multi *m, double *d_sums_out, uint64_t *nloci_out) {
    if (!m) return fail(NPS_E_INVAL, "ctx is NULL");
    if (m->n && !d_sums_out) return fail(NPS_E_INVAL, "d_sums_out is NULL");
    return multi_finish_common(m, nullptr, d_sums_out, nullptr, nloci_out, 0);
}

extern "C" int nps_multi_partial(nps_multi *m, double *sums_out, uint64_t *nloci_out) {
    if (!m) return fail(NPS_E_INVAL, "ctx is NULL");
    if (m->n && !sums_out) return fail(NPS_E_INVAL, "sums_out is NULL");
    return multi_finish_common(m, nullptr, m->d_scores, sums_out, nloci_out, 0);
}

extern "C" int nps_multi_timing(nps_multi *m, double *ms_params, double *ms_product, double *ms_fold) {
    if (!m) return fail(NPS_E_INVAL, "ctx is NULL");
    HIP_TRY(hipSetDevice(m->device));
    multi_drain_timing(m);
    if (ms_params) *ms_params = m->ms[0];
    if (ms_product) *ms_product = m->ms[1];
    if (ms_fold) *ms_fold = m->ms[2];
    return NPS_OK;
}

// ------------------------------------------------------------------------------------------
extern "C" int nps_profile_enable(nps_ctx *c, int on) {
    if (!c) return fail(NPS_E_INVAL, "ctx is NULL");
    c->profiling = on != 0;
    return NPS_OK;
}

extern "C" int nps_profile_get(nps_ctx *c, nps_profile *out, int reset) {
    if (!c || !out) return fail(NPS_E_INVAL, "NULL argument");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    resolve_spans(c);
    *out = c->prof;
    if (reset) c->prof = nps_profile{};
    return NPS_OK;
}

extern "C" void *nps_stream(nps_ctx *c) { return c ? (void *)c->stream : nullptr; }

extern "C" int nps_fused_geometry(nps_ctx *c, int format, uint64_t n_rows, uint32_t *slices,
                                  uint32_t *teams, uint32_t *samples_per_slice) {
    if (!c) return fail(NPS_E_INVAL, "ctx is NULL");
    const int ds_elem = format == NPS_FMT_DS16 ? 2 : 4;
    if (format == NPS_FMT_DS16) format = NPS_FMT_DS32;  // (the same kernel and slices; four rows per batch)
    if (format != NPS_FMT_GT2 && format != NPS_FMT_DS32 && format != NPS_FMT_GT2X)
        return fail(NPS_E_INVAL, "unknown format %d", format);
    HIP_TRY(hipSetDevice(c->device));
    if (format == NPS_FMT_GT2X) {
        MxPlan mp;
        HIP_TRY(mx_plan(c->device, c->n, n_rows, false, &mp));
        // (more strips than compute units: the grid of the accumulation with given tallies -- kept with the cohort, or from
        //  the tally pass; its slices are still the 2048-sample strips)
        if (mp.ok && mp.given) HIP_TRY(mx_plan(c->device, c->n, n_rows, true, &mp));
        // (the single-read kernel may cut its own strips of 62 units = 1 984 samples from the unit sequence: what the grid IS)
        const bool vs = mp.ok && !mp.given && mp.U < 64;
        if (slices) *slices = mp.ok ? (vs ? mp.Pv : mp.P) : 0;
        if (teams) *teams = mp.ok ? mp.Q : 0;
        if (samples_per_slice) *samples_per_slice = mp.ok ? (vs ? 32 * mp.U : 2048) : 0;
        return NPS_OK;
    }
    FusedPlan plan;
    if (format == NPS_FMT_DS32)
        HIP_TRY(ds_fused_plan(c->device, c->n, n_rows, 0, 0, &plan, ds_elem));
    else
        HIP_TRY(fused_plan(c->device, c->n, n_rows, 0, 0, &plan));
    if (slices) *slices = plan.ok ? plan.P : 0;
    if (teams) *teams = plan.ok ? plan.Q : 0;
    if (samples_per_slice)
        *samples_per_slice = !plan.ok ? 0 : (format == NPS_FMT_DS32 ? (plan.threads - 64) * 8 : (plan.threads - 64) * 16);
    return NPS_OK;
}
