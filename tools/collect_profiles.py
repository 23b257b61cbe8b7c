#!/usr/bin/env python3
"""Copy what the round's GPU runs left under gpurun_out/ (tools/profile_round.sh r05, tools/pmc1.sh r05, the timers builds,
the counted test run, tools/soak.py) into profiles/ with the headers that say how each file was made, then regenerate the
resource report and profiles/README.md's rows:     python tools/collect_profiles.py r05"""
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")


def lines(name, keep):
    path = os.path.join(G, name)
    if not os.path.exists(path):
        return []
    return [l for l in open(path, errors="replace").read().splitlines() if "amdgpu" not in l and any(k in l for k in keep)]


for sub in ("prof_%s" % tag, "prof_%s_ds" % tag):
    d = os.path.join(G, sub, "summary")
    if os.path.isdir(d):
        for f in os.listdir(d):
            shutil.copy(os.path.join(d, f), os.path.join(P, f))
sq = os.path.join(G, "pmc_%s" % tag, "summary.txt")
if os.path.exists(sq):
    open(os.path.join(P, "%s_pmc_sq.txt" % tag), "w").write(
        "# round %s, final tree: SQ counters of the strip kernel at the bench shape (500 000 x 1 000 000: 253 strips of 62 units),\n"
        "# tools/pmc1.sh %s: three separate rocprofv3 --pmc passes (no trace domains combined), average per launch over 3 launches.\n" % (tag[1:].lstrip("0"), tag)
        + open(sq).read())
mx, mxg = lines("%s_mx_timers_raw.txt" % tag, ("timers", "best")), lines("%s_mxg_timers_raw.txt" % tag, ("timers", "best"))
if mx:
    open(os.path.join(P, "%s_mx_timers.txt" % tag), "w").write("\n".join([
        "# round %s, final tree: cycles per phase of a step of the shipped strip kernel (nps_mx.hip: strips of 62 units, data waves 0..3 ten" % tag[1:].lstrip("0"),
        "# units, data waves 4, 5 nine, control waves 6, 7 two units + 64 rows each; wave w runs on SIMD w % 4) for ALL eight waves, and of the",
        "# given-tallies kernel (nps_mxg.hip); build: tools/mkexp.sh t -DNPS_MX_TIMERS -DNPS_DIAGNOSTICS; run: cd exp/t && python tools/qb_mx.py",
        "# --mode 2 | --mode 1   (500 000 x 1 000 000).",
        "# Reading: busy = front + accumulate (+ the control waves' look, operands, publication); what is left of the step is the wait at the",
        "# barrier.  The waves that wait least set the step: the second data wave of SIMD 0 / 1 and the control waves are within a few",
        "# hundred cycles of each other.  (Nine + five units, until round 5's last changes: control waves last, path 6 900, data waves",
        "# waiting 3 300.)",
        "## first form, mode 2 (tallies in the pass)"] + mx[-9:] + ["## tallies given, mode 1 (tally pass + nps_mxg.hip)"] + mxg[-4:]) + "\n")
m2, m2ps = lines("%s_mx2_timers_raw.txt" % tag, ("timers", "best")), lines("%s_mx2ps_timers_raw.txt" % tag, ("timers", "best"))
old = os.path.join(P, "%s_mx2_timers.txt" % tag)
if m2 and os.path.exists(old):
    head = [l for l in open(old).read().splitlines() if l.startswith("# ")]
    open(old, "w").write("\n".join(head + ["## --imp-sample int_ps (the CLI default): operands computed by the control wave"] + m2[-5:] +
                                   ["## --imp-sample ps: operands ready made (Bm table by LDS-DMA)"] + m2ps[-5:]) + "\n")
cnt = lines("%s_test_counts_raw.txt" % tag, ("check_scores", "passed", "failed"))
soak = lines("%s_soak_raw.txt" % tag, ("soak ok", "Error", "error"))
# (long soak runs appended by hand below the generated part survive a regeneration)
prev = os.path.join(P, "%s_test_counts.txt" % tag)
extra = []
if os.path.exists(prev):
    body = open(prev).read().splitlines()
    cut = [i for i, l in enumerate(body) if l.startswith("# tools/soak.py --format")]
    extra = body[cut[0]:] if cut else []
if cnt:
    open(os.path.join(P, "%s_test_counts.txt" % tag), "w").write("\n".join([
        "# round %s, final tree: python -m pytest tests/test_gpu_mx.py -q -s (one MI355X): the samples that passed check_scores only through" % tag[1:].lstrip("0"),
        "# its 2^-50 escape, over every call of the module (asserted < 1 in 1000 per call and in total)"] +
        [re.sub(r"^\.+", "", l) for l in cnt] +
        ["# tools/soak.py on the same box (bit-identical passes, two definitions alternating):"] + soak + extra) + "\n")
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), tag])
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "profiles_readme.py"), tag])
