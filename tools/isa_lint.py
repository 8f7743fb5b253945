#!/usr/bin/env python
"""ISA lint for the LDS-DMA rings (CPU only: hipcc -S).

hipcc waits for a VGPR-returning VMEM load (global / buffer / scratch reload) with a COUNTED `s_waitcnt vmcnt(N)`, N = the
VMEM instructions it issued after the load, which is right only if everything retires in issue order.  LDS-DMA
(`buffer_load ... lds`) and VGPR-returning loads do not retire in one order, so a load whose first use
is guarded only by counted waits, with LDS-DMA instructions issued in between, can be read before it has landed.  The
kernels avoid the pattern by construction (explicit vmcnt(0) before a ring starts and before every burst); this script
proves it on the generated code: for every VGPR-returning load it follows the straight-line code to the first read of
its destination and reports the load if LDS-DMA was issued on the way and no wait on the way was vmcnt(0).

Second rule (tail.hip issues its MFMAs from inline assembly, which hipcc's hazard recogniser does not see): between such
an MFMA and the first instruction that reads its result without being an MFMA accumulating into the same registers there must
be >= 12 wait states (what hipcc itself leaves behind the 8-pass 32x32x16 MFMAs these kernels use), counted
conservatively: s_nop N = N + 1, any other MFMA = 16 (4x4x4: 8), anything else = 1.

Third rule (the race of DESIGN.md section 3): in a kernel that issues LDS-DMA, no LDS read sits between an s_barrier and
the nearest wait in front of it that drains the wave's LDS counter (lgkmcnt(0)) -- otherwise a
fast wave's refill of a ring stage can overtake a slow wave's outstanding fragment reads of that stage.

Fourth rule (round 3: a BT_PREC_F32X3 attention variant with 12 bytes of scratch gave different results on every run --
two registers spilled in front of the key loop and reloaded behind it, next to LDS-DMA still in flight, the covering
vmcnt(0) in a conditionally skipped block, which the straight-line scan of the first rule took for a guard): a kernel
that issues LDS-DMA must not use scratch memory at all, unless it is on the allow list below (tail.hip's 512-register
kernel, whose reloads are drained by an explicit vmcnt(0) in front of every LDS-DMA burst).

Fifth rule (round 5): no kernel touches scratch beyond its budget in SCRATCH_OK (0 unless listed) -- spills are never wrong
without LDS-DMA, but they are never intended either, and nothing else reports them.

    python tools/isa_lint.py [source.hip ...]        exit status 1 if anything is reported
"""
import concurrent.futures
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from beat_this_amd import _lib  # noqa: E402

LOAD = re.compile(r"^\s*(scratch_load|global_load|buffer_load|flat_load)\w*\s+(v\[(\d+):(\d+)\]|v(\d+))")
WAIT = re.compile(r"vmcnt\((\d+)\)")


def asm_of(src, defines=()):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    subprocess.run(_lib.device_asm_command(src, out, defines), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(out).read()
    os.unlink(out)
    return text


MFMA = re.compile(r"^\s*v_mfma_\w+\s+([av])\[(\d+):(\d+)\]")
REG = re.compile(r"\b([av])(\d+)\b|\b([av])\[(\d+):(\d+)\]")
MFMA_WAIT_STATES = 12


def lint_mfma(lines, src):
    findings = []
    kernel = "?"
    in_asm = False
    for i, line in enumerate(lines):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel = m.group(1)
        if "#ASMSTART" in line:
            in_asm = True
        elif "#ASMEND" in line:
            in_asm = False
        m = MFMA.match(line)
        if not m or not in_asm:   # (MFMAs the compiler emitted itself are covered by its hazard recogniser)
            continue
        bank, lo, hi = m.group(1), int(m.group(2)), int(m.group(3))
        states = 0
        for j in range(i + 1, min(i + 200, len(lines))):
            x = lines[j].split(";")[0]
            if not x.strip() or x.lstrip().startswith(".") or x.rstrip().endswith(":"):
                continue
            if "s_endpgm" in x or states >= MFMA_WAIT_STATES:
                break
            ops = x.split(None, 1)
            touches = any((g[0] == bank and lo <= int(g[1]) <= hi) if g[0] else (g[2] == bank and int(g[3]) <= hi and int(g[4]) >= lo)
                          for g in REG.findall(ops[1] if len(ops) > 1 else ""))
            m2 = MFMA.match(x)
            if touches and not (m2 and m2.group(1) == bank and int(m2.group(2)) == lo):
                findings.append(f"{os.path.basename(src)}: {kernel}: line {j + 1}: `{x.strip()}` uses the result of the MFMA at line "
                                f"{i + 1} after {states} wait state(s)")
                break
            n = re.match(r"\s*s_nop\s+(\d+)", x)
            states += int(n.group(1)) + 1 if n else ((8 if "_4x4x4" in x else 16) if m2 else 1)
    return findings


def lint_barriers(lines, src):
    findings = []
    bounds = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)] + [len(lines)]
    for a, b in zip(bounds, bounds[1:]):
        body = lines[a:b]
        if not any("buffer_load" in x and " lds" in x for x in body):
            continue
        kernel = body[0].split(":")[0]
        for i, x in enumerate(body):
            if not re.match(r"\s*s_barrier\b", x):
                continue
            ok = True
            for j in range(i - 1, 0, -1):   # back to the nearest drain of the LDS counter; an LDS read on the way is outstanding
                y = body[j].split(";")[0]
                if "s_waitcnt" in y and "lgkmcnt(0)" in y:
                    break
                if re.match(r"\s*ds_(read|bpermute|permute|swizzle)", y):
                    ok = False
                    break
            if not ok:
                findings.append(f"{os.path.basename(src)}: {kernel}: line {a + i + 1}: s_barrier without a preceding lgkmcnt(0) "
                                f"(LDS reads may be outstanding when another wave refills the stage)")
    return findings


SCRATCH_OK = {"layer_tail_kernel": 40}   # kernels that may touch scratch, with their budget of accesses (the 512-register tail: 36;
                                          # with LDS-DMA only because its bursts sit behind explicit vmcnt(0) drains)


def lint_scratch(lines, src):
    findings = []
    bounds = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)] + [len(lines)]
    for a, b in zip(bounds, bounds[1:]):
        body = lines[a:b]
        kernel = body[0].split(":")[0]
        n = sum(1 for x in body if re.match(r"\s*scratch_(load|store)", x))
        allowed = next((v for k, v in SCRATCH_OK.items() if k in kernel), 0)
        if n and not allowed and any("buffer_load" in x and " lds" in x for x in body):
            findings.append(f"{os.path.basename(src)}: {kernel}: {n} scratch access(es) in a kernel that issues LDS-DMA "
                            f"(spill reloads are waited for with counted vmcnt values that LDS-DMA invalidates)")
        # fifth rule: no kernel spills silently (a burst of register-staged loads in the attention's fix-up launch once came back
        # as 97 scratch accesses and a launch of a millisecond; correct, and invisible without this count)
        elif n > allowed:
            findings.append(f"{os.path.basename(src)}: {kernel}: {n} scratch access(es), budget {allowed} (SCRATCH_OK): "
                            f"a spill nobody asked for")
    return findings


def lint(src):
    findings = []
    kernel = "?"
    lines = asm_of(src).split("\n")
    findings += lint_mfma(lines, src)
    findings += lint_barriers(lines, src)
    findings += lint_scratch(lines, src)
    for i, line in enumerate(lines):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kernel = m.group(1)
        m = LOAD.match(line)
        if not m or " lds" in line:
            continue
        lo, hi = (int(m.group(3)), int(m.group(4))) if m.group(3) else (int(m.group(5)), int(m.group(5)))
        regs = re.compile(r"\bv(?:%s)\b|\bv\[(\d+):(\d+)\]" % "|".join(str(r) for r in range(lo, hi + 1)))
        dma, drained = 0, False
        for j in range(i + 1, min(i + 4000, len(lines))):
            x = lines[j]
            if "s_endpgm" in x:
                break
            w = WAIT.search(x)
            if w and int(w.group(1)) == 0:
                drained = True
                break
            if "buffer_load" in x and " lds" in x:
                dma += 1
                continue
            used = False
            for u in regs.finditer(x.split(";")[0]):
                if u.group(1) is None or (int(u.group(1)) <= hi and int(u.group(2)) >= lo):
                    used = True
            if used and not LOAD.match(x):
                break
        if dma and not drained:
            findings.append(f"{os.path.basename(src)}: {kernel}: line {i + 1}: `{line.strip().split(';')[0].strip()}` is first used "
                            f"behind {dma} LDS-DMA instruction(s) without a vmcnt(0) in between")
    return findings


def main(argv):
    srcs = argv or [os.path.join(_lib.PKG_DIR, "csrc", s) for s in _lib.SOURCES]
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lint, srcs))
    bad = [f for r in res for f in r]
    for f in bad:
        print(f)
    print(f"isa_lint: {len(srcs)} sources, {len(bad)} finding(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
