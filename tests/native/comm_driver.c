/* A host that is NOT Python: plain C over include/nps.h + include/nps_comm.h, the way the Nim host of INTEGRATION.md
 * section 4 would bind them.  One process, all visible GPUs (1 on the test box; `comm_driver N` uses N):
 *   layout 1 (score files sharded): context r scores definition r over a synthetic resident cohort on device r, then
 *       nps_comm_allgather_scores; row r of every device's matrix must equal nps_finish of context r, bit for bit;
 *   layout 2 (rows of one score sharded): context r scores rows [r m / N, (r + 1) m / N) (128-aligned) of ONE definition,
 *       then nps_comm_allreduce_partial; the result must equal the unsharded run within 1e-12 relative (blocked sums).
 * Self-checking: per rank one line with its device, nloci and an FNV-1a checksum of the result it ended up with (all ranks
 * must print the SAME checksum per layout, and it must equal the per-context reference's), then "comm_driver ok ..." and
 * status 0 -- or a message and a non-zero status.  The first run on an 8-GPU node is `comm_driver 8`.  Compiled by tests/test_gpu_comm.py with
 * gcc (C11) against libnps.so and libnps_rccl.so; no HIP headers -- the four HIP runtime calls it needs for its own result
 * buffers are declared by hand, as a Nim host would importc them. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nps.h"
#include "nps_comm.h"

extern int hipMalloc(void **p, size_t n);            /* (libamdhip64, which libnps.so brings in) */
extern int hipFree(void *p);
extern int hipSetDevice(int d);
extern int hipMemcpy(void *dst, const void *src, size_t n, int kind);

#define CHECK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "%s: %d: %s | %s\n", #x, rc_, nps_last_error(), nps_comm_last_error()); return 1; } } while (0)

static uint64_t fnv1a(const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *)p;
    uint64_t h = 0xcbf29ce484222325ull;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 0x100000001b3ull; }
    return h;
}
static uint32_t thr(double p) { double v = floor(p * 4294967296.0); return v > 4294967295.0 ? 4294967295u : (uint32_t)v; }

int main(int argc, char **argv) {
    int ndev = nps_device_count();
    if (ndev < 1) { fprintf(stderr, "no device\n"); return 2; }
    int N = argc > 1 ? atoi(argv[1]) : ndev;
    if (N < 1 || N > ndev || N > 8) { fprintf(stderr, "bad device count %d (visible %d)\n", N, ndev); return 2; }
    const uint64_t n = 5000, m = 1024;
    nps_params prm;
    memset(&prm, 0, sizeof(prm));
    prm.imp_locus = NPS_LOCUS_PS; prm.imp_missing = NPS_MISSING_HOMREF; prm.imp_sample = NPS_SAMPLE_INT_PS;
    prm.max_missing_rate = 0.05; prm.min_cs = 100;
    uint32_t *th = malloc(m * 4), *tm = malloc(m * 4), *tmi = malloc(m * 4);
    nps_row_desc *desc = malloc(sizeof(nps_row_desc) * m * (size_t)N);
    for (uint64_t j = 0; j < m; ++j) {
        double eaf = 0.05 + 0.4 * (double)((j * 2654435761u) % 1000) / 1000.0, miss = (j % 50 == 0) ? 0.2 : 0.01;
        th[j] = thr(eaf * eaf + 2 * eaf * (1 - eaf)); tm[j] = thr(eaf * eaf); tmi[j] = thr(miss);
        for (int r = 0; r < N; ++r) {
            nps_row_desc *d = &desc[(size_t)r * m + j];
            memset(d, 0, sizeof(*d));
            d->beta = 0.001 * (double)((int)((j * 97 + r * 131) % 201) - 100); d->eaf = eaf; d->kind = NPS_ROW_PRESENT;
        }
    }
    nps_comm *comm = NULL;
    CHECK(nps_comm_init_all(&comm, N, NULL));
    if (nps_comm_size(comm) != N) return 3;
    nps_cohort *co[8]; nps_ctx *ctx[8]; double *d_mat[8]; double offs[8]; uint64_t nloci[8];
    double *want = malloc(sizeof(double) * n * (size_t)N), *got = malloc(sizeof(double) * n * (size_t)N);
    for (int r = 0; r < N; ++r) {
        int dev = nps_comm_device(comm, r);
        CHECK(nps_cohort_create(&co[r], dev, n, m, NPS_FMT_GT_AUTO));
        CHECK(nps_cohort_synth(co[r], 0, m, 4242, th, tm, tmi));
        CHECK(nps_create(&ctx[r], dev, n, &prm));
        offs[r] = 0.1 * r;
        hipSetDevice(dev);
        if (hipMalloc((void **)&d_mat[r], sizeof(double) * n * (size_t)N)) return 4;
        /* the expected row: this definition through nps_finish on its own */
        CHECK(nps_score_cohort(ctx[r], co[r], 0, &desc[(size_t)r * m], m, NPS_MODE_AUTO));
        CHECK(nps_finish(ctx[r], offs[r], want + (size_t)r * n, &nloci[r]));
        CHECK(nps_reset(ctx[r], NULL));
        CHECK(nps_score_cohort(ctx[r], co[r], 0, &desc[(size_t)r * m], m, NPS_MODE_AUTO));
    }
    uint64_t nl[8];
    CHECK(nps_comm_allgather_scores(comm, ctx, offs, d_mat, nl));
    const uint64_t want_sum = fnv1a(want, sizeof(double) * n * (size_t)N);
    for (int r = 0; r < N; ++r) {
        hipSetDevice(nps_comm_device(comm, r));
        if (hipMemcpy(got, d_mat[r], sizeof(double) * n * (size_t)N, 2 /* hipMemcpyDeviceToHost */)) return 5;
        const uint64_t sum = fnv1a(got, sizeof(double) * n * (size_t)N);
        printf("comm_driver rank %d device %d: all-gather of %d score row(s), nloci of its own score %llu, matrix checksum %016llx%s\n",
               r, nps_comm_device(comm, r), N, (unsigned long long)nl[r], (unsigned long long)sum, sum == want_sum ? "" : " MISMATCH");
        if (memcmp(got, want, sizeof(double) * n * (size_t)N) != 0 || nl[r] != nloci[r]) {
            fprintf(stderr, "gathered matrix on device %d differs from the per-context results\n", r);
            return 6;
        }
    }
    /* layout 2: ONE definition (definition 0), its rows in N 128-aligned blocks */
    double worst = 0.0;
    uint64_t total = 0, blk_a[8], blk_b[8];
    for (int r = 0; r < N; ++r) {
        uint64_t a = (m * (uint64_t)r / (uint64_t)N) / 128 * 128, b = r == N - 1 ? m : (m * (uint64_t)(r + 1) / (uint64_t)N) / 128 * 128;
        blk_a[r] = a; blk_b[r] = b;
        CHECK(nps_reset(ctx[r], NULL));
        if (b > a) CHECK(nps_score_cohort(ctx[r], co[r], a, &desc[a], b - a, NPS_MODE_AUTO));
    }
    CHECK(nps_comm_allreduce_partial(comm, ctx, offs[0], d_mat, &total));
    if (total != nloci[0]) { fprintf(stderr, "nloci %llu != %llu\n", (unsigned long long)total, (unsigned long long)nloci[0]); return 7; }
    uint64_t sum0 = 0;
    for (int r = 0; r < N; ++r) {
        hipSetDevice(nps_comm_device(comm, r));
        if (hipMemcpy(got, d_mat[r], sizeof(double) * n, 2)) return 5;
        const uint64_t sum = fnv1a(got, sizeof(double) * n);
        if (r == 0) sum0 = sum;
        printf("comm_driver rank %d device %d: all-reduce, rows [%llu, %llu) of %llu, nloci of all blocks %llu, score checksum %016llx%s\n",
               r, nps_comm_device(comm, r), (unsigned long long)blk_a[r], (unsigned long long)blk_b[r], (unsigned long long)m,
               (unsigned long long)total, (unsigned long long)sum, sum == sum0 ? "" : " MISMATCH");
        if (sum != sum0) { fprintf(stderr, "rank %d ended the all-reduce with other bits than rank 0\n", r); return 9; }
        for (uint64_t i = 0; i < n; ++i) {
            double d = fabs(got[i] - want[i]), s = fabs(want[i]) > 1e-9 ? fabs(want[i]) : 1e-9;
            if (d / s > worst) worst = d / s;
        }
    }
    if (!(worst <= 1e-12)) { fprintf(stderr, "row-sharded result differs: max relative %g\n", worst); return 8; }
    for (int r = 0; r < N; ++r) {
        hipSetDevice(nps_comm_device(comm, r));
        hipFree(d_mat[r]);
        nps_destroy(ctx[r]);
        nps_cohort_destroy(co[r]);
    }
    nps_comm_destroy(comm);
    printf("comm_driver ok: %d device(s), %llu samples x %llu rows, all-gather bit-identical, all-reduce max relative %.3g, nloci %llu\n",
           N, (unsigned long long)n, (unsigned long long)m, worst, (unsigned long long)total);
    return 0;
}
