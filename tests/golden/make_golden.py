#!/usr/bin/env python3
"""tests/golden/make_golden.py -- regenerates the committed golden fixtures.

Run in the authoring container only (it reads /root/reference, which does not exist on the GPU
box).  The reference is Nim + htslib and cannot be executed here, so the golden vectors are the
EXPECTED VALUES the reference's own test-suite holds, extracted as data:

  set1_cases.json   <- tests/test_set1.nim:36-190   13 active cases x 6 samples (tol 1e-4 abs,
                       NaN positions exact), with each case's explicit parameters
  stats_kats.json   <- tests/test_stats.nim:21-139  87 known answers for betai/dbinom/pbinom/
                       binom_test (rel 1e-5 / abs 1e-9)
  data files the reference's tests and repo hold (copied byte for byte, they are data):
    set1.vcf.gz(.tbi) set1.score set1.bed set1.plink190.result set1.plink200.result
    set1.plink.freq set1.plink.score
    scores/*.scores  makescore/example/*/..score   (the 8 score definitions of config 4)
    result_format/*_nimpress_res.txt              (pins the "%.16g" output format)

No reference SOURCE text is stored: only numbers, parameter names and data files.
"""
import json
import os
import re
import shutil
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def parse_float_expr(tok: str) -> float:
    tok = tok.strip()
    if tok == "NaN":
        return float("nan")
    # expressions like 0.123-0.03 / 0.123+0.0208 appear in the PLINK case
    if not re.fullmatch(r"[0-9eE+\-. ]+", tok):
        raise ValueError(tok)
    return float(eval(tok, {"__builtins__": {}}))


def extract_set1():
    src = open(os.path.join(REF, "tests/test_set1.nim")).read().split("\n")
    cases = []
    i = 0
    while i < len(src):
        line = src[i]
        m = re.match(r'\s*test "(.*)":\s*$', line)
        if m and not line.lstrip().startswith("#"):
            name = m.group(1)
            start = i + 1
            j = i + 1
            block = []
            while j < len(src) and "check(checkFloats" not in src[j]:
                block.append(src[j])
                j += 1
            block.append(src[j])
            text = " ".join(s.strip() for s in block)
            call = re.search(r"computePolygenicScores\((.*?)\)\s*(#.*?)?check\(", text).group(1)
            args = [a.strip() for a in call.split(",")]
            # positional: scores, scoreFile, genotypeVcf, cov(bool), coveredBed, locus, missing,
            # sample, maxMissingRate=, afMismatchPthresh=, minGtForInternalImput=, ignoreFilterField=
            def kw(a):
                return a.split("=")[1].strip() if "=" in a else a
            case = {
                "name": name,
                "ref_lines": [start, j + 1],
                "restrict_to_covered": args[3] == "true",
                "imp_locus": args[5].split(".")[1],
                "imp_missing": args[6].split(".")[1],
                "imp_sample": args[7].split(".")[1],
                "maxmis": float(kw(args[8])),
                "afmisp": float(kw(args[9])),
                "mincs": int(kw(args[10])),
                "ignore_filter": kw(args[11]) == "true",
            }
            exp = re.search(r"checkFloats\(scores,\s*@\[(.*?)\]\)\)", text).group(1)
            vals = [parse_float_expr(t) for t in exp.split(",")]
            case["expected"] = [None if v != v else v for v in vals]
            cases.append(case)
            i = j
        i += 1
    assert len(cases) == 13, len(cases)
    out = {"source": "tests/test_set1.nim", "tolerance_abs": 1e-4,
           "inputs": {"vcf": "set1.vcf.gz", "score": "set1.score", "bed": "set1.bed"},
           "samples": ["S1", "S2", "S3", "S4", "S5", "S6"], "cases": cases}
    json.dump(out, open(os.path.join(HERE, "set1_cases.json"), "w"), indent=1)
    return len(cases)


def extract_stats():
    src = open(os.path.join(REF, "tests/test_stats.nim")).read()
    kats = []
    for m in re.finditer(r"check_floatvalue\((\w+)\(\s*([^)]*?)\),\s*([0-9eE+\-.]+)\)", src):
        fn, args, target = m.group(1), m.group(2), m.group(3)
        kats.append({"fn": fn, "args": [float(a) for a in args.split(",")],
                     "expected": float(target), "mode": "approx"})
    for m in re.finditer(r"^\s*(dbinom|binom_test)\(\s*([^)]*?)\)\s*==\s*([0-9.]+)\s*$", src, re.M):
        kats.append({"fn": m.group(1), "args": [float(a) for a in m.group(2).split(",")],
                     "expected": float(m.group(3)), "mode": "exact"})
    assert len(kats) == 87, len(kats)
    out = {"source": "tests/test_stats.nim", "rel_tol": 1e-5, "abs_tol": 1e-9, "kats": kats}
    json.dump(out, open(os.path.join(HERE, "stats_kats.json"), "w"), indent=1)
    return len(kats)


def copy_data():
    files = ["set1.vcf.gz", "set1.vcf.gz.tbi", "set1.plink.vcf.gz", "set1.plink.vcf.gz.tbi", "set1.score", "set1.bed", "set1.plink190.result",
             "set1.plink200.result", "set1.plink.freq", "set1.plink.score"]
    for f in files:
        shutil.copyfile(os.path.join(REF, "tests", f), os.path.join(HERE, f))
    os.makedirs(os.path.join(HERE, "scores"), exist_ok=True)
    for f in sorted(os.listdir(os.path.join(REF, "scores"))):
        if f.endswith(".scores"):
            shutil.copyfile(os.path.join(REF, "scores", f), os.path.join(HERE, "scores", f))
    for k in (1, 2, 3):
        shutil.copyfile(
            os.path.join(REF, "makescore/example/wood_height_small_example%d" % k,
                         "wood_height_input_small.score"),
            os.path.join(HERE, "scores", "wood_height_small_example%d.score" % k))
    os.makedirs(os.path.join(HERE, "result_format"), exist_ok=True)
    for f in sorted(os.listdir(os.path.join(REF, "scores"))):
        if f.endswith("_nimpress_res.txt"):
            shutil.copyfile(os.path.join(REF, "scores", f),
                            os.path.join(HERE, "result_format", f))
    for root, _, fs in os.walk(HERE):
        for f in fs:
            os.chmod(os.path.join(root, f), 0o644)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; the committed fixtures are already in tests/golden/")
    n1 = extract_set1()
    n2 = extract_stats()
    copy_data()
    print("set1 cases: %d, stats KATs: %d" % (n1, n2))
