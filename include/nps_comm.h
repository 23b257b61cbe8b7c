/* nps_comm.h -- the multi-GPU exchange step of the nimpress hot path for hosts that are NOT Python:
 * a single process that drives all GPUs of one node (the Nim host of BASELINE.json's north_star, a C or C++ caller).
 *
 * The reference is single-threaded and has no exchange at all: main() calls computePolygenicScores once and prints
 * (nimpress.nim:747-753).  The two layouts SURVEY.md section 8(e) derives from north_star put one nps_ctx (include/nps.h)
 * on every GPU and end with ONE collective each:
 *
 *   score files sharded across the GPUs (configs[3]):  context r holds the finished score of file r on device r;
 *       nps_comm_allgather_scores builds the scores x samples matrix on every device -- what main()'s output loop
 *       (nimpress.nim:752-753) would print, column by column.
 *   rows of ONE score sharded across the GPUs (configs[2], [4] at 2/4/8 GPUs):  context r holds the un-normalised sums
 *       and nloci of its block of score rows; nps_comm_allreduce_partial adds them up and applies
 *       nimpress.nim:643-649 ( / (2 nloci) + offset ) on every device.
 *
 * This library (libnps_rccl.so) is separate from libnps.so so that libnps keeps no RCCL dependency; it binds libnps's
 * public C-ABI only.  One process, ncclCommInitAll over the devices, one stream per device, ncclGroupStart/End around
 * the per-device calls: on an MI355X node every GPU pair has its own xGMI link, the payloads are at most
 * samples x scores x 8 B (32 MB at 500 000 x 8), i.e. latency-bound.
 *
 * torch.distributed callers (one process per GPU: bench.py, tools/score_many.py) use nimpress_amd/multi.py instead; both
 * end in the same RCCL collectives.
 *
 * Every call returns 0 or a negative nps_status (include/nps.h); nps_comm_last_error() has the text.  A communicator is
 * used by one thread at a time. */
#ifndef NPS_COMM_H
#define NPS_COMM_H

#include <stdint.h>

#include "nps.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nps_comm nps_comm;

const char *nps_comm_last_error(void);

/* One communicator over n_devices GPUs of this node; devices = their HIP ordinals, NULL = 0 .. n_devices-1.
 * n_devices = 1 is legal (the collectives become copies). */
int nps_comm_init_all(nps_comm **out, int n_devices, const int *devices);
int nps_comm_size(const nps_comm *c);
int nps_comm_device(const nps_comm *c, int rank); /* HIP ordinal of rank `rank` */
void nps_comm_destroy(nps_comm *c);

/* Score files sharded across the GPUs.  ctxs[r] (created on device r of the communicator, all with the same n_samples)
 * holds the pushed / scored rows of score r; offsets[r] is that file's offset (nimpress.nim:648-649).
 * d_matrix[r]: device memory on device r, n_devices x n_samples doubles; on return EVERY d_matrix[r] holds the whole
 * matrix, row s = the scores of file s (nps_finish_device of ctxs[s]) for all samples.  nloci_out[r] (may be NULL) = the
 * loci context r used.  Replaces, for N files on N GPUs, N runs of nimpress.nim:747-753. */
int nps_comm_allgather_scores(nps_comm *c, nps_ctx *const *ctxs, const double *offsets, double *const *d_matrix,
                              uint64_t *nloci_out);

/* Rows of ONE score sharded across the GPUs.  ctxs[r] has scored its block of the score's rows (all samples);
 * d_scores[r]: n_samples doubles on device r.  On return every d_scores[r] holds the final scores
 * (sum over the blocks) / (2 x total nloci) + offset  (nimpress.nim:643-649); *nloci_out (may be NULL) = total nloci. */
int nps_comm_allreduce_partial(nps_comm *c, nps_ctx *const *ctxs, double offset, double *const *d_scores,
                               uint64_t *nloci_out);

/* Rows sharded x ALL S scores per GPU (the layout 8 GPUs and many score files should use: 1/N of the ingest per GPU).
 * ms[r] (nps_multi, S scores each) has scored its block of the union of the files' rows; d_matrix[r]: S x n_samples
 * doubles on device r.  On return every d_matrix[r] holds the final S x n_samples matrix; nloci_out (may be NULL): S
 * totals.  offsets: S values. */
int nps_comm_allreduce_partial_multi(nps_comm *c, nps_multi *const *ms, int n_scores, uint64_t n_samples,
                                     const double *offsets, double *const *d_matrix, uint64_t *nloci_out);

#ifdef __cplusplus
}
#endif
#endif /* NPS_COMM_H */
