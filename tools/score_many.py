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
import json
import os
import sys
import time

T_START = time.perf_counter()   # (the interpreter's own start-up is before this: bench.py measures from outside)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--one-pass", action="store_true",
                    help="this rank's score files in ONE pass over the genotypes: the union of their loci is decoded once "
                         "into a resident cohort and the definitions are applied together on the matrix cores (falls back "
                         "to the per-file loop for inputs that path does not cover, e.g. FORMAT/DS records)")
    ap.add_argument("--shard", choices=["auto", "files", "rows"], default="auto",
                    help="what the GPUs split.  files: each GPU takes every N-th score file (BASELINE.json's wording; the whole "
                         "cohort file is read by every GPU and 697-locus files sit beside 4-locus ones).  rows: each GPU takes "
                         "1/N of the ROWS of the union of the files' loci and scores ALL files on it in one matrix-core pass -- "
                         "1/N of the ingest and of the cohort per GPU, equal load -- and one sum all-reduce of the [files, "
                         "samples] sums and the per-file locus counts follows (implies --one-pass; inputs that path does not "
                         "cover, e.g. FORMAT/DS records, are an error: pass --shard files).  auto (default): rows when there "
                         "is more than one GPU and the genotype file can be fetched by locus (.tbi / .csi index, PLINK "
                         "fileset), files otherwise")
    ap.add_argument("--out", default="-", help="output TSV (default: stdout)")
    ap.add_argument("--timings", action="store_true",
                    help="rank 0 prints one JSON line on stderr saying where the run's time went (imports, HIP context, "
                         "file open, inflate + parse, push, kernels, warnings, gather, output)")
    ap.add_argument("--rank-timeout", type=float, default=1800.0,
                    help="seconds after which a self-started multi-rank run is killed as a whole (exit status 124)")
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


def has_locus_index(path):
    """the genotype file can be fetched locus by locus (every rank reads its block of rows only)"""
    if any(os.path.exists(path + ext) for ext in (".tbi", ".csi")):
        return True
    stem, ext = os.path.splitext(path)
    if ext == ".bed":
        return os.path.exists(stem + ".bim")
    if ext == ".pgen":
        return os.path.exists(stem + ".pvar") or os.path.exists(stem + ".bim")
    return False


def spawn_ranks(n, timeout_s):
    import importlib.util
    spec = importlib.util.spec_from_file_location("nps_launch", os.path.join(ROOT, "nimpress_amd", "launch.py"))
    launch = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(launch)
    # (every rank's warnings -- with --shard rows those of its own block of rows -- go to this command's stderr)
    launch.spawn_ranks(__file__, sys.argv[1:], n, timeout_s, name="score_many.py", inherit_stderr=True)   # does not return


def main(argv=None):
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus, args.rank_timeout)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import numpy as np
    from nimpress_amd import capi, host
    capi.load(with_torch=world > 1 or os.environ.get("NIMPRESS_DEVICE_RESULTS") == "1")   # (one process, one GPU: torch is never imported -- its import is a third of such a run)
    stamp = {"imports_s": time.perf_counter() - T_START}   # numpy, ctypes, dlopen of libnps + the HIP runtime
    torch = dist = multi = device = None
    # (ranks that share a GPU exist only in tests: NIMPRESS_DIST_BACKEND=gloo, the exchange on CPU tensors)
    backend = os.environ.get("NIMPRESS_DIST_BACKEND", "nccl")
    if world > 1:   # (one process, one GPU: no exchange, and torch's import is a fifth of such a run)
        ndev = capi.device_count()
        if ndev < 1:
            sys.exit("score_many.py needs an MI355X: libnps has no CPU path")
        if backend == "nccl" and local_rank >= ndev:
            sys.exit("score_many.py: LOCAL_RANK %d but only %d GPU(s) visible: one RCCL rank per GPU (start at most %d "
                     "ranks per node)" % (local_rank, ndev, ndev))
        local_rank %= ndev   # (only ever wraps for the GPU-sharing test backend)
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
    # results stay in device memory between the scoring and the exchange when the exchange runs on the GPU (RCCL);
    # NIMPRESS_DEVICE_RESULTS=1 takes that path with a single rank too (tests: the one-GPU rehearsal of the 8-GPU code)
    force_dev = os.environ.get("NIMPRESS_DEVICE_RESULTS") == "1"
    on_gpu = (world > 1 and backend == "nccl") or force_dev
    if force_dev and world == 1:
        import torch
        from nimpress_amd import multi
        device = torch.device("cuda", local_rank)
        torch.cuda.set_device(local_rank)
    score_files, cohort = args.files[:-1], args.files[-1]
    shard_auto = args.shard == "auto"
    if shard_auto:   # (every rank sees the same files: the same choice)
        args.shard = "rows" if world > 1 and not args.one_pass and has_locus_index(cohort) else "files"
    names = host.sample_names(cohort)
    n = len(names)
    S = len(score_files)
    logs = {}
    tim = {k: 0.0 for k in host.TIMING_KEYS}

    def took():
        for k, v in host.last_timings().items():
            tim[k] += v

    kw = dict(cov=args.cov, imp_locus=args.imp_locus, imp_missing=args.imp_missing, imp_sample=args.imp_sample,
              maxmis=args.maxmis, mincs=args.mincs, afmisp=args.afmisp, ignorefilt=args.ignorefilt, device=local_rank,
              max_samples=max(n, 1))

    def score_file(i, d_out=None):
        sc, nloci, log = host.compute_polygenic_scores(score_files[i], cohort, d_out=d_out, **kw)
        took()
        logs[i] = log
        return sc

    one_pass_used = False
    t0 = time.perf_counter()
    t_gather = 0.0
    mat = None
    if args.shard == "rows":
        # rows sharded over the GPUs x all files on every GPU: one partial pass, one all-reduce, the normalisation
        d_sums = torch.zeros((S, n), dtype=torch.float64, device=device) if on_gpu else None
        # The one-pass multi-score path does not take everything the file-by-file path takes (FORMAT/DS records; a
        # definition whose |beta| span exceeds 2^25): under `--shard auto` such an input falls back to `--shard files`, as
        # the one-pass branch below does -- on EVERY rank or on none (the ranks agree through an all-reduce of a flag);
        # an explicit `--shard rows` stays a hard error.
        failed, why = 0, ""
        try:
            sums, nl, offs, lg = host.compute_polygenic_scores_multi_partial(
                score_files, cohort, rank, world, d_out=d_sums.data_ptr() if on_gpu else None, **kw)
            took()
        except (capi.NpsError, RuntimeError) as e:
            if not shard_auto:
                raise
            failed, why = 1, str(e)
        if shard_auto and world > 1:
            flag = torch.tensor([failed], dtype=torch.int32, device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            failed = int(flag.item())
        if failed:
            if rank == 0:
                sys.stderr.write("score_many: rows x all-files path not applicable (%s); sharding the score files instead\n"
                                 % (why or "refused on another rank"))
            args.shard = "files"
            d_sums = None
    if args.shard == "rows":
        for i, lines in enumerate(lg):
            logs[i] = lines
        one_pass_used = True
        if world > 1 or on_gpu:
            tg = time.perf_counter()
            t, cnt = multi.all_reduce_partial_matrix(d_sums if on_gpu else torch.from_numpy(sums).to(device), nl)
            mat = multi.normalize_matrix(t, cnt, offs).cpu().numpy()
            t_gather = time.perf_counter() - tg
        else:
            with np.errstate(divide="ignore", invalid="ignore"):
                mat = sums / (nl.astype(np.float64) * 2.0)[:, None] + offs[:, None]
    if args.shard != "rows":
        mine = list(range(rank, S, world))
        # this rank's rows of the matrix, in shard order: on the GPU when the gather runs there
        local = None if (world == 1 and not on_gpu) else torch.empty((len(mine), n), dtype=torch.float64, device=device)
        done = set()
        if args.one_pass and mine:
            try:
                sc, _nl, lg = host.compute_polygenic_scores_multi(
                    [score_files[i] for i in mine], cohort, d_out=local.data_ptr() if on_gpu else None, **kw)
                took()
                for k, i in enumerate(mine):
                    logs[i] = lg[k]
                    if local is not None and not on_gpu:
                        local[k].copy_(torch.from_numpy(sc[k]))
                done = set(mine)
                one_pass_used = True
                if world == 1 and not on_gpu:
                    mat = sc
            except capi.NpsError as e:
                sys.stderr.write("score_many: one-pass path not applicable (%s); scoring file by file\n" % e)
        if mat is None and world == 1 and not on_gpu:
            mat = np.empty((S, n), dtype=np.float64)
            for i in range(S):
                mat[i] = score_file(i)
        elif world > 1 or on_gpu:
            for k, i in enumerate(mine):
                if i in done:
                    continue
                if on_gpu:
                    score_file(i, d_out=local[k].data_ptr())   # nps_finish_device writes the gather's send buffer
                else:
                    local[k].copy_(torch.from_numpy(score_file(i)))
            tg = time.perf_counter()
            mat = multi.gather_scores(local, S).cpu().numpy()
            t_gather = time.perf_counter() - tg
    elapsed = time.perf_counter() - t0
    for i in sorted(logs):
        for line in logs[i]:
            sys.stderr.write("[%s] %s\n" % (os.path.basename(score_files[i]), line))
    if rank == 0:
        tw = time.perf_counter()
        host.write_matrix_tsv(args.out, names, mat)   # the reference's float format, 16 threads in C++
        t_write = time.perf_counter() - tw
        sys.stderr.write("score_many: %d score files x %d samples on %d GPU(s) in %.2f s%s\n"
                         % (S, n, world, elapsed,
                            " (rows sharded over the GPUs, all files per GPU in one pass)" if args.shard == "rows" else
                            " (one pass over the genotypes)" if one_pass_used else ""))
        if args.timings:
            out = {"imports_s": stamp["imports_s"], "sample_names_s": t0 - T_START - stamp["imports_s"]}
            out.update(tim)
            out.update({"gather_s": t_gather, "write_s": t_write, "total_in_process_s": time.perf_counter() - T_START,
                        "ranks": world, "exchange_on": "gpu (RCCL)" if on_gpu else ("cpu (%s)" % backend if world > 1 else "none")})
            sys.stderr.write(json.dumps({"score_many_timings": out}) + "\n")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
