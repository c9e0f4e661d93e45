#!/usr/bin/env python
"""Static check of the built score kernels: registers written by the hand-issued (inline asm,
not scoreboarded by hipcc) global loads must not be read, copied, spilled or overwritten before
a `s_waitcnt vmcnt(N)` that covers the load (N <= number of younger loads).  The scan follows the
program text (the head loop is unrolled by its register sets, so the text order is the execution order
of consecutive heads); it is a tripwire for bad register allocation, not a proof.  hipcc knows nothing about those loads being in flight, so a
register move or spill it inserts in that window would capture stale data.

usage: python tools/check_isa.py [file.s]   (default: compiles kvquant_amd/csrc/kvq_score_k.hip
with -save-temps into a temporary directory)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wno-unused-value", "-fvisibility=hidden", "-DKVQ_BUILD"]

LOAD = re.compile(r"^\s*global_load_dword\s+([va])(\d+),\s*v\d+,\s*s\[\d+:\d+\]")
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def regs_of(line):
    """registers named by an instruction: {('v', 12), ('a', 0), ...}"""
    out = set()
    for m in REG.finditer(line.split(";")[0]):
        if m.group(1) is not None:
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), r) for r in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def check(path, kernel_substr="score_k_kernel"):
    lines = open(path).read().splitlines()
    problems = []
    kernels = 0
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w*%s\w*):" % kernel_substr, lines[i])
        if not m:
            i += 1
            continue
        name = m.group(1)
        kernels += 1
        j = i + 1
        body = []
        while j < len(lines) and "s_endpgm" not in lines[j]:
            body.append(lines[j])
            j += 1
        in_asm = False
        for k, ln in enumerate(body):
            if "#ASMSTART" in ln:
                in_asm = True
                continue
            if "#ASMEND" in ln:
                in_asm = False
                continue
            lm = LOAD.match(ln)
            if not (in_asm and lm):
                continue
            dest = (lm.group(1), int(lm.group(2)))
            asm2 = False
            younger = 0
            for ln2 in body[k + 1:]:
                if "#ASMSTART" in ln2:
                    asm2 = True
                    continue
                if "#ASMEND" in ln2:
                    asm2 = False
                    continue
                code = ln2.split(";")[0].strip()
                if not code or code.endswith(":") or code.startswith("."):
                    continue
                wm = re.search(r"vmcnt\((\d+)\)", code) if code.startswith("s_waitcnt") else None
                if wm:
                    # vmcnt(N) leaves at most the N youngest operations outstanding and loads return in order
                    # among themselves (stores may overtake them, so only younger LOADS count)
                    if younger >= int(wm.group(1)):
                        break
                    continue
                if re.match(r"(global|buffer|scratch|flat)_load", code):
                    younger += 1
                if code.startswith("s_branch"):       # straight-line path ends (what follows is another path)
                    break
                if dest in regs_of(code):
                    problems.append("%s: %s%d (asm load in flight) touched by `%s`" % (name, dest[0], dest[1], code))
                    break
        i = j
    return kernels, problems


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        tmp = tempfile.mkdtemp(prefix="kvq_isa_")
        src = os.path.join(ROOT, "kvquant_amd", "csrc", "kvq_score_k.hip")
        subprocess.check_call([HIPCC] + FLAGS + ["-save-temps", "-c", src, "-o", os.path.join(tmp, "k.o")], cwd=tmp,
                              stderr=subprocess.DEVNULL)
        path = [os.path.join(tmp, f) for f in os.listdir(tmp) if f.endswith(".s") and "amdgcn" in f][0]
    kernels, problems = check(path)
    print("%d kernels checked, %d problems" % (kernels, len(problems)))
    for p in problems:
        print("  " + p)
    return 1 if problems or kernels == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
