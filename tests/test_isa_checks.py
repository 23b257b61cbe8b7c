"""Build-time guards for the two hand-timed memory operations of the strip kernels: a load or
returning atomic issued through inline asm writes its destination registers LATER, when the hand-written `s_waitcnt vmcnt`
has passed -- the compiler believes they are ready at once.  If it ever copies, spills or otherwise touches those
registers between the two statements, the kernel computes from stale data and no build error says so (in round 5 a
compiler-placed `v_mov` in front of one of two per-branch wait statements did exactly that to nps_mx2.hip's first draft;
the parity tests caught it as a timeout, this test would have named the line).

The check compiles the two kernels' device code to assembly (hipcc cross-compiles without a GPU) and, for every such
operation, walks the text up to the first hand-written vmcnt wait that follows: no instruction in between may name the
destination registers.  (Text order, not control flow: sufficient -- a window with no mention at all cannot contain a
misplaced copy on any path -- and cheap.)"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nimpress_amd", "csrc")


def device_asm(src, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = str(tmp_path / (os.path.basename(src) + ".s"))
    r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-parameter",
                        "--cuda-device-only", "-S", src, "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read().splitlines()


def vgprs(text):
    """every VGPR a line names: v7, v[4:5] -> {7}, {4, 5}"""
    regs = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        regs.update(range(int(a), int(b) + 1))
    for a in re.findall(r"\bv(\d+)\b", text):
        regs.add(int(a))
    return regs


def check_windows(lines, issue_re, what):
    """for every asm-issued operation matching issue_re: its destination registers are not named again until the
    hand-written (inside an ASM block) vmcnt wait that follows has gone by -- and, where the source has one wait statement
    per branch, until the second one has: the compiler lays the two blocks out one behind the other with nothing but
    exec-mask bookkeeping between them, so a wait block that starts within a dozen lines of the previous one belongs to the
    same site and the walk goes on through it.  Returns the number of operations checked."""
    def is_wait_block(k):     # lines[k] is an ASMSTART line: does the block hold a vmcnt wait?  -> (yes, index of its ASMEND)
        e = k + 1
        hit = False
        while e < len(lines) and "#ASMEND" not in lines[e]:
            hit = hit or re.search(r"s_waitcnt\s+vmcnt\(", lines[e]) is not None
            e += 1
        return hit, e

    n = 0
    in_asm = False
    for i, ln in enumerate(lines):
        if "#ASMSTART" in ln:
            in_asm = True
        elif "#ASMEND" in ln:
            in_asm = False
        m = issue_re.search(ln) if in_asm else None
        if not m:
            continue
        dest = vgprs(m.group(1))
        assert dest, ln
        j, waits = i + 1, 0
        while j < len(lines) and "#ASMEND" not in lines[j]:   # (the rest of the issuing block)
            j += 1
        j += 1
        while j < len(lines):
            t = lines[j]
            if t.strip().startswith(".Lfunc_end") or t.strip().startswith(".section"):
                break
            if "#ASMSTART" in t:
                hit, e = is_wait_block(j)
                if hit:
                    waits += 1
                    # another wait block of the same site right behind this one?
                    nxt = [k for k in range(e + 1, min(e + 14, len(lines))) if "#ASMSTART" in lines[k] and is_wait_block(k)[0]]
                    if not nxt:
                        break
                j = e + 1
                continue
            code = t.split(";")[0]
            hit = dest & vgprs(code)
            assert not hit, ("%s: line %d touches v%s between the asm-issued operation at line %d (%s) and its hand-written "
                             "wait(s):\n%s" % (what, j + 1, sorted(hit), i + 1, ln.strip(), t))
            j += 1
        assert waits, "%s: no hand-written vmcnt wait follows line %d (%s)" % (what, i + 1, ln.strip())
        n += 1
    return n


def test_mx_look_registers_untouched_until_the_hand_written_wait(tmp_path):
    """nps_mx.hip: the control waves' look at the row tally words (`global_load_dwordx2 ... sc1` in inline asm)"""
    lines = device_asm(os.path.join(CSRC, "nps_mx.hip"), tmp_path)
    n = check_windows(lines, re.compile(r"global_load_dwordx2\s+(v\[\d+:\d+\]),.*\bsc1\b"), "nps_mx.hip look")
    assert n >= 2   # (two control-wave bodies -- full and ragged strips -- per instantiated kernel)


def test_register_bound_kernels_do_not_spill(tmp_path):
    """Two kernels sit at their register limit by design and lose a third of their speed with the first spilled register:
    the float32 single-read DS kernel (128 VGPRs: sixteen waves per compute unit; round 5: a harmless-looking change of its
    ring's element type spilled 80 bytes per lane, 38 -> 61 ms) and the strip kernel.  The compiler's own resource remarks
    must say ScratchSize 0 for them."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    want = {"nps_ds_fused.hip": ["ds_fused_kernelILi1024ELb0", "ds_fused_kernelILi960ELb0", "ds_fused_kernelILi896ELb0",
                                 "ds_fused_kernelILi1024ELb1"],
            "nps_mx.hip": ["fused_mx_kernelILi0ELb0"]}
    for src, kernels in want.items():
        r = subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-parameter",
                            "--cuda-device-only", "-c", os.path.join(CSRC, src), "-o", str(tmp_path / (src + ".o")),
                            "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        name, seen = None, {}
        for ln in r.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", ln)
            if m:
                name = m.group(1)
            m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", ln)
            if m and name:
                seen[name] = int(m.group(1))
        for k in kernels:
            hits = [v for nm, v in seen.items() if k in nm]
            assert hits, (src, k, sorted(seen))
            assert all(v == 0 for v in hits), (src, k, hits)
