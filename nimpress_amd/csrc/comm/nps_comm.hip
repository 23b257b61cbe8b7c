// nps_comm.hip -- libnps_rccl.so: the exchange step of the multi-GPU layouts for single-process hosts (include/nps_comm.h).
// Binds libnps's public C-ABI (include/nps.h) and RCCL; libnps.so itself has no RCCL dependency.
//
// One process drives n devices: ncclCommInitAll, one HIP stream per device, every collective inside
// ncclGroupStart/End (a single thread may not block on one rank's call while the others are not posted).
// xGMI on an MI355X node is point-to-point between every GPU pair, the payloads are <= samples x scores x 8 bytes:
// latency-bound, nothing to pipeline.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdarg>
#include <cstdio>
#include <new>
#include <string>
#include <vector>

#include "nps_comm.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return fail(NPS_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));           \
    } while (0)
#define NCCL_TRY(expr)                                                                                         \
    do {                                                                                                       \
        ncclResult_t r_ = (expr);                                                                              \
        if (r_ != ncclSuccess) return fail(NPS_E_HIP, "%s failed: %s", #expr, ncclGetErrorString(r_));         \
    } while (0)
#define NPS_TRY(expr, what)                                                                                    \
    do {                                                                                                       \
        int rc_ = (expr);                                                                                      \
        if (rc_ != NPS_OK) return fail(rc_, "%s: %s", what, nps_last_error());                                 \
    } while (0)

// nimpress.nim:643-649 for the S x n matrix of a row-sharded multi-score run: m[s][i] = m[s][i] / (2 nloci[s]) + offset[s]
__global__ __launch_bounds__(256) void normalize_matrix_kernel(double *__restrict__ m, uint64_t n, const double *__restrict__ nloci2,
                                                               const double *__restrict__ offsets) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const int s = blockIdx.y;
    if (i >= n) return;
    double v = m[(uint64_t)s * n + i];
    v /= nloci2[s];  // (nloci.toFloat * 2.0, :645; 0 loci: x / 0 as in the reference)
    v += offsets[s];
    m[(uint64_t)s * n + i] = v;
}

}  // namespace

struct nps_comm {
    int n = 0;
    std::vector<int> dev;
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> stream;
    std::vector<double *> d_scal;  // per device: [2][NPS_MULTI_MAX_SCORES] doubles (2 nloci, offsets) for the matrix normalisation
};

extern "C" const char *nps_comm_last_error(void) { return g_err.c_str(); }

extern "C" int nps_comm_init_all(nps_comm **out, int n_devices, const int *devices) {
    if (!out) return fail(NPS_E_INVAL, "out is NULL");
    *out = nullptr;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) {
        (void)hipGetLastError();
        return fail(NPS_E_NODEVICE, "no HIP device visible (nps_comm has no CPU path)");
    }
    if (n_devices < 1 || n_devices > visible) return fail(NPS_E_INVAL, "n_devices %d outside 1..%d", n_devices, visible);
    nps_comm *c = new (std::nothrow) nps_comm;
    if (!c) return fail(NPS_E_NOMEM, "out of host memory");
    c->n = n_devices;
    c->dev.resize(n_devices);
    for (int r = 0; r < n_devices; ++r) {
        c->dev[r] = devices ? devices[r] : r;
        if (c->dev[r] < 0 || c->dev[r] >= visible) {
            delete c;
            return fail(NPS_E_INVAL, "device %d of rank %d is not visible (0..%d)", devices ? devices[r] : r, r, visible - 1);
        }
        for (int q = 0; q < r; ++q)
            if (c->dev[q] == c->dev[r]) {
                delete c;
                return fail(NPS_E_INVAL, "device %d listed twice (one RCCL rank per GPU)", c->dev[r]);
            }
    }
    c->comm.assign(n_devices, nullptr);
    c->stream.assign(n_devices, nullptr);
    c->d_scal.assign(n_devices, nullptr);
    ncclResult_t nr = ncclCommInitAll(c->comm.data(), n_devices, c->dev.data());
    if (nr != ncclSuccess) {
        delete c;
        return fail(NPS_E_HIP, "ncclCommInitAll over %d device(s) failed: %s", n_devices, ncclGetErrorString(nr));
    }
    for (int r = 0; r < n_devices; ++r) {
        hipError_t e = hipSetDevice(c->dev[r]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream[r], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc(&c->d_scal[r], sizeof(double) * 2 * NPS_MULTI_MAX_SCORES);
        if (e != hipSuccess) {
            nps_comm_destroy(c);
            return fail(NPS_E_HIP, "setting up device %d failed: %s", c->dev[r], hipGetErrorString(e));
        }
    }
    *out = c;
    return NPS_OK;
}

extern "C" int nps_comm_size(const nps_comm *c) { return c ? c->n : 0; }
extern "C" int nps_comm_device(const nps_comm *c, int rank) { return c && rank >= 0 && rank < c->n ? c->dev[rank] : -1; }

extern "C" void nps_comm_destroy(nps_comm *c) {
    if (!c) return;
    for (int r = 0; r < c->n; ++r) {
        (void)hipSetDevice(c->dev[r]);
        if (c->stream[r]) {
            (void)hipStreamSynchronize(c->stream[r]);
            (void)hipStreamDestroy(c->stream[r]);
        }
        (void)hipFree(c->d_scal[r]);
        if (c->comm[r]) (void)ncclCommDestroy(c->comm[r]);
    }
    delete c;
}

static int check_ctxs(const nps_comm *c, nps_ctx *const *ctxs, uint64_t *n_out) {
    if (!c || !ctxs) return fail(NPS_E_INVAL, "communicator / contexts are NULL");
    uint64_t n = 0;
    for (int r = 0; r < c->n; ++r) {
        if (!ctxs[r]) return fail(NPS_E_INVAL, "context of rank %d is NULL", r);
        if (nps_device(ctxs[r]) != c->dev[r])
            return fail(NPS_E_INVAL, "context of rank %d lives on device %d, the communicator's rank %d is device %d", r,
                        nps_device(ctxs[r]), r, c->dev[r]);
        if (r == 0) n = nps_n_samples(ctxs[r]);
        if (nps_n_samples(ctxs[r]) != n)
            return fail(NPS_E_INVAL, "contexts differ in n_samples (%llu on rank 0, %llu on rank %d)", (unsigned long long)n,
                        (unsigned long long)nps_n_samples(ctxs[r]), r);
    }
    *n_out = n;
    return NPS_OK;
}

extern "C" int nps_comm_allgather_scores(nps_comm *c, nps_ctx *const *ctxs, const double *offsets, double *const *d_matrix,
                                         uint64_t *nloci_out) {
    uint64_t n = 0;
    int rc = check_ctxs(c, ctxs, &n);
    if (rc) return rc;
    if (!offsets || !d_matrix) return fail(NPS_E_INVAL, "offsets / d_matrix are NULL");
    for (int r = 0; r < c->n; ++r)
        if (!d_matrix[r] && n) return fail(NPS_E_INVAL, "d_matrix[%d] is NULL", r);
    // every context finishes its score straight into its own row of its device's matrix (the in-place form of the
    // all-gather: sendbuff = recvbuff + rank * count), then one grouped ncclAllGather
    for (int r = 0; r < c->n; ++r) {
        uint64_t nl = 0;
        NPS_TRY(nps_finish_device(ctxs[r], offsets[r], d_matrix[r] + (uint64_t)r * n, &nl), "nps_finish_device");
        if (nloci_out) nloci_out[r] = nl;
    }
    if (n == 0) return NPS_OK;
    // (nps_finish_device returns with the result complete: it copies nloci back, which synchronises its stream)
    NCCL_TRY(ncclGroupStart());
    for (int r = 0; r < c->n; ++r) {
        ncclResult_t nr = ncclAllGather(d_matrix[r] + (uint64_t)r * n, d_matrix[r], n, ncclDouble, c->comm[r], c->stream[r]);
        if (nr != ncclSuccess) {
            (void)ncclGroupEnd();
            return fail(NPS_E_HIP, "ncclAllGather on rank %d failed: %s", r, ncclGetErrorString(nr));
        }
    }
    NCCL_TRY(ncclGroupEnd());
    for (int r = 0; r < c->n; ++r) {
        HIP_TRY(hipSetDevice(c->dev[r]));
        HIP_TRY(hipStreamSynchronize(c->stream[r]));
    }
    return NPS_OK;
}

extern "C" int nps_comm_allreduce_partial(nps_comm *c, nps_ctx *const *ctxs, double offset, double *const *d_scores,
                                          uint64_t *nloci_out) {
    uint64_t n = 0;
    int rc = check_ctxs(c, ctxs, &n);
    if (rc) return rc;
    if (!d_scores) return fail(NPS_E_INVAL, "d_scores is NULL");
    for (int r = 0; r < c->n; ++r)
        if (!d_scores[r] && n) return fail(NPS_E_INVAL, "d_scores[%d] is NULL", r);
    uint64_t total = 0;  // one process holds every rank's count: the nloci "all-reduce" is a host sum
    for (int r = 0; r < c->n; ++r) {
        uint64_t nl = 0;
        NPS_TRY(nps_partial_device(ctxs[r], d_scores[r], &nl), "nps_partial_device");
        total += nl;
    }
    if (n && c->n > 1) {
        NCCL_TRY(ncclGroupStart());
        for (int r = 0; r < c->n; ++r) {
            ncclResult_t nr = ncclAllReduce(d_scores[r], d_scores[r], n, ncclDouble, ncclSum, c->comm[r], c->stream[r]);
            if (nr != ncclSuccess) {
                (void)ncclGroupEnd();
                return fail(NPS_E_HIP, "ncclAllReduce on rank %d failed: %s", r, ncclGetErrorString(nr));
            }
        }
        NCCL_TRY(ncclGroupEnd());
        for (int r = 0; r < c->n; ++r) {
            HIP_TRY(hipSetDevice(c->dev[r]));
            HIP_TRY(hipStreamSynchronize(c->stream[r]));
        }
    }
    for (int r = 0; r < c->n; ++r) NPS_TRY(nps_normalize_device(ctxs[r], d_scores[r], total, offset), "nps_normalize_device");
    if (nloci_out) *nloci_out = total;
    return NPS_OK;
}

extern "C" int nps_comm_allreduce_partial_multi(nps_comm *c, nps_multi *const *ms, int n_scores, uint64_t n_samples,
                                                const double *offsets, double *const *d_matrix, uint64_t *nloci_out) {
    if (!c || !ms || !offsets || !d_matrix) return fail(NPS_E_INVAL, "an argument is NULL");
    if (n_scores < 1 || n_scores > NPS_MULTI_MAX_SCORES)
        return fail(NPS_E_INVAL, "n_scores %d outside 1..%d", n_scores, NPS_MULTI_MAX_SCORES);
    for (int r = 0; r < c->n; ++r)
        if (!ms[r] || (!d_matrix[r] && n_samples)) return fail(NPS_E_INVAL, "rank %d: scorer / d_matrix is NULL", r);
    // nps_multi_partial_device writes the scorer's OWN S x n doubles: what the caller says must be what the scorers are, and
    // every scorer must live on its rank's device (as check_ctxs asks of single-score contexts)
    for (int r = 0; r < c->n; ++r) {
        if (nps_multi_device(ms[r]) != c->dev[r])
            return fail(NPS_E_INVAL, "rank %d: the scorer is on device %d, the communicator's rank on device %d", r,
                        nps_multi_device(ms[r]), c->dev[r]);
        if (nps_multi_n_scores(ms[r]) != n_scores || nps_multi_n_samples(ms[r]) != n_samples)
            return fail(NPS_E_INVAL, "rank %d: the scorer holds %d scores x %llu samples, the call says %d x %llu", r,
                        nps_multi_n_scores(ms[r]), (unsigned long long)nps_multi_n_samples(ms[r]), n_scores,
                        (unsigned long long)n_samples);
    }
    uint64_t total[NPS_MULTI_MAX_SCORES] = {};
    for (int r = 0; r < c->n; ++r) {
        uint64_t nl[NPS_MULTI_MAX_SCORES] = {};
        NPS_TRY(nps_multi_partial_device(ms[r], d_matrix[r], nl), "nps_multi_partial_device");
        for (int s = 0; s < n_scores; ++s) total[s] += nl[s];
    }
    const uint64_t count = (uint64_t)n_scores * n_samples;
    if (count && c->n > 1) {
        NCCL_TRY(ncclGroupStart());
        for (int r = 0; r < c->n; ++r) {
            ncclResult_t nr = ncclAllReduce(d_matrix[r], d_matrix[r], count, ncclDouble, ncclSum, c->comm[r], c->stream[r]);
            if (nr != ncclSuccess) {
                (void)ncclGroupEnd();
                return fail(NPS_E_HIP, "ncclAllReduce on rank %d failed: %s", r, ncclGetErrorString(nr));
            }
        }
        NCCL_TRY(ncclGroupEnd());
    }
    double scal[2 * NPS_MULTI_MAX_SCORES] = {};
    for (int s = 0; s < n_scores; ++s) {
        scal[s] = (double)total[s] * 2.0;
        scal[NPS_MULTI_MAX_SCORES + s] = offsets[s];
    }
    for (int r = 0; r < c->n; ++r) {
        HIP_TRY(hipSetDevice(c->dev[r]));
        // (a blocking copy of 128 bytes: `scal` lives on this frame, and an early error return below must not leave a copy
        //  from it in flight; the stream's all-reduce is waited for first, the kernel follows on the same stream)
        HIP_TRY(hipStreamSynchronize(c->stream[r]));
        HIP_TRY(hipMemcpy(c->d_scal[r], scal, sizeof(scal), hipMemcpyHostToDevice));
        if (count)
            hipLaunchKernelGGL(normalize_matrix_kernel, dim3((uint32_t)((n_samples + 255) / 256), (uint32_t)n_scores), dim3(256),
                               0, c->stream[r], d_matrix[r], n_samples, c->d_scal[r], c->d_scal[r] + NPS_MULTI_MAX_SCORES);
        HIP_TRY(hipGetLastError());
    }
    for (int r = 0; r < c->n; ++r) {
        HIP_TRY(hipSetDevice(c->dev[r]));
        HIP_TRY(hipStreamSynchronize(c->stream[r]));
    }
    if (nloci_out)
        for (int s = 0; s < n_scores; ++s) nloci_out[s] = total[s];
    return NPS_OK;
}
