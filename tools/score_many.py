#!/usr/bin/env python3
"""score_many.py -- BASELINE.json configs[3]: several score files evaluated on ONE cohort file, one score
definition per GPU at a time, the samples x scores matrix gathered over RCCL.

    python tools/score_many.py [--gpus N] [nimpress options] --out matrix.tsv  a.scores b.scores ...  cohort.bcf

Each rank (one process per GPU; torch.distributed, backend nccl = RCCL on ROCm) runs the reference's
computePolygenicScores (nimpress.nim:592-649, C++ host + libnps) for the score files
i = rank, rank + N, ... against the same cohort file; the only exchange is the all-gather of the
[scores, samples] matrix (nimpress_amd/multi.py).  Rank 0 writes one line per sample:
    <sample> TAB <score of file 1> TAB <score of file 2> ...
in the reference's float format (nimpress.nim:752-753).  With --gpus N > 1 and no launcher
(torch.distributed.run) this process starts the N ranks itself, before anything touches a GPU.
"""
import argparse
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--one-pass", action="store_true",
                    help="this rank's score files in ONE pass over the genotypes: the union of their loci is decoded once "
                         "into a resident cohort and the definitions are applied together on the matrix cores (falls back "
                         "to the per-file loop for inputs that path does not cover, e.g. FORMAT/DS records)")
    ap.add_argument("--shard", choices=["files", "rows"], default="files",
                    help="what the GPUs split.  files: each GPU takes every N-th score file (the whole cohort file is read "
                         "by every GPU).  rows: each GPU takes 1/N of the ROWS of the union of the files' loci and scores "
                         "ALL files on it in one matrix-core pass -- 1/N of the ingest and of the cohort per GPU -- and one "
                         "sum all-reduce of the [files, samples] sums and the per-file locus counts follows (implies "
                         "--one-pass; inputs that path does not cover are an error)")
    ap.add_argument("--out", default="-", help="output TSV (default: stdout)")
    ap.add_argument("--cov", default=None)
    ap.add_argument("--imp-locus", default="ps", choices=["ps", "homref", "fail", "ignore"])
    ap.add_argument("--imp-missing", default="homref", choices=["homref", "ignore"])
    ap.add_argument("--imp-sample", default="int_ps", choices=["ps", "homref", "fail", "int_ps", "int_fail"])
    ap.add_argument("--maxmis", type=float, default=0.05)
    ap.add_argument("--mincs", type=int, default=100)
    ap.add_argument("--afmisp", type=float, default=0.001)
    ap.add_argument("--ignorefilt", action="store_true")
    ap.add_argument("files", nargs="+", help="score files ..., then the genotype file (last)")
    a = ap.parse_args(argv)
    if len(a.files) < 2:
        ap.error("need at least one score file and the genotype file")
    return a


def spawn_ranks(n):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = list(procs)
    try:
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in pending:
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    sys.exit(rc)


def main(argv=None):
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import numpy as np
    from nimpress_amd import capi, host
    capi.load()
    if capi.device_count() < 1:
        sys.exit("score_many.py needs an MI355X: libnps has no CPU path")
    torch = dist = multi = device = None
    # (ranks that share a GPU exist only in tests: NIMPRESS_DIST_BACKEND=gloo, the exchange on CPU tensors)
    backend = os.environ.get("NIMPRESS_DIST_BACKEND", "nccl")
    local_rank %= capi.device_count()
    if world > 1:   # (one process, one GPU: no exchange, and torch's import is a fifth of such a run)
        import torch
        import torch.distributed as dist
        from nimpress_amd import multi
        if backend == "nccl":
            device = torch.device("cuda", local_rank)
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=device)
        else:
            device = torch.device("cpu")
            dist.init_process_group(backend)
    score_files, cohort = args.files[:-1], args.files[-1]
    names = host.sample_names(cohort)
    n = len(names)
    logs = {}

    def score_file(i):
        s, nloci, log = host.compute_polygenic_scores(
            score_files[i], cohort, cov=args.cov, imp_locus=args.imp_locus, imp_missing=args.imp_missing,
            imp_sample=args.imp_sample, maxmis=args.maxmis, mincs=args.mincs, afmisp=args.afmisp,
            ignorefilt=args.ignorefilt, device=local_rank, max_samples=max(n, 1))
        logs[i] = log
        return s

    one_pass_used = False
    t0 = time.perf_counter()
    cache = {}
    mat = None
    if args.shard == "rows":
        # rows sharded over the GPUs x all files on every GPU: one partial pass, one all-reduce, the normalisation
        sums, nl, offs, lg = host.compute_polygenic_scores_multi_partial(
            score_files, cohort, rank, world, cov=args.cov, imp_locus=args.imp_locus, imp_missing=args.imp_missing,
            imp_sample=args.imp_sample, maxmis=args.maxmis, mincs=args.mincs, afmisp=args.afmisp,
            ignorefilt=args.ignorefilt, device=local_rank, max_samples=max(n, 1))
        for i, lines in enumerate(lg):
            logs[i] = lines
        one_pass_used = True
        if world > 1:
            t, cnt = multi.all_reduce_partial_matrix(torch.from_numpy(sums).to(device), nl)
            mat = multi.normalize_matrix(t, cnt, offs).cpu().numpy()
        else:
            with np.errstate(divide="ignore", invalid="ignore"):
                mat = sums / (nl.astype(np.float64) * 2.0)[:, None] + offs[:, None]
    elif args.one_pass:
        mine = list(range(rank, len(score_files), world))
        try:
            sc, _nl, lg = host.compute_polygenic_scores_multi(
                [score_files[i] for i in mine], cohort, cov=args.cov, imp_locus=args.imp_locus,
                imp_missing=args.imp_missing, imp_sample=args.imp_sample, maxmis=args.maxmis, mincs=args.mincs,
                afmisp=args.afmisp, ignorefilt=args.ignorefilt, device=local_rank, max_samples=max(n, 1)) if mine else (None, None, [])
            cache = {i: sc[k] for k, i in enumerate(mine)}
            for k, i in enumerate(mine):
                logs[i] = lg[k]
            one_pass_used = True
        except capi.NpsError as e:
            sys.stderr.write("score_many: one-pass path not applicable (%s); scoring file by file\n" % e)

    def row(i):  # scores of file i (from the one pass, if there was one)
        return cache[i] if i in cache else score_file(i)

    if mat is not None:
        pass
    elif world == 1:
        mat = np.empty((len(score_files), n), dtype=np.float64)
        for i in range(len(score_files)):
            mat[i] = row(i)
    else:
        full = multi.evaluate_sharded(len(score_files), n,
                                      lambda i, out_row: out_row.copy_(torch.from_numpy(row(i)).to(out_row.device)), device)
        mat = full.cpu().numpy()
    elapsed = time.perf_counter() - t0
    for i in sorted(logs):
        for line in logs[i]:
            sys.stderr.write("[%s] %s\n" % (os.path.basename(score_files[i]), line))
    if rank == 0:
        host.write_matrix_tsv(args.out, names, mat)   # the reference's float format, 16 threads in C++
        sys.stderr.write("score_many: %d score files x %d samples on %d GPU(s) in %.2f s%s\n"
                         % (len(score_files), n, world, elapsed,
                            " (rows sharded over the GPUs, all files per GPU in one pass)" if args.shard == "rows" else
                            " (one pass over the genotypes)" if one_pass_used else ""))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
