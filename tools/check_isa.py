#!/usr/bin/env python
"""Static check of the built score kernels: registers written by the hand-issued (inline asm,
not scoreboarded by hipcc) global loads must not be read, copied, spilled or overwritten before
a `s_waitcnt vmcnt(N)` that covers the load (N <= number of younger loads).  The scan follows the
control flow of the generated code -- both sides of every branch, loop back-edges included: the mirror
kernel re-loads a register set for head h+2 while head h is being decoded, so a load is in flight across
a whole loop iteration -- and assumes the fewest younger loads on every path.  hipcc knows nothing about
those loads being in flight, so a register move or spill it inserts in that window would capture stale data.

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


def _parse(body):
    """instructions of a kernel body: [(text, in_asm)], label -> instruction index"""
    insts, labels = [], {}
    in_asm = False
    for ln in body:
        if "#ASMSTART" in ln:
            in_asm = True
            continue
        if "#ASMEND" in ln:
            in_asm = False
            continue
        code = ln.split(";")[0].strip()
        if not code or code.startswith("."):
            lm = re.match(r"^(\.?[\w$.]+):", code)
            if lm:
                labels[lm.group(1)] = len(insts)
            continue
        lm = re.match(r"^([\w$.]+):$", code)
        if lm:
            labels[lm.group(1)] = len(insts)
            continue
        insts.append((code, in_asm))
    return insts, labels


def _successors(insts, labels, i):
    code = insts[i][0]
    op = code.split()[0]
    if op == "s_endpgm":
        return []
    if op == "s_branch":
        return [labels[code.split()[1]]]
    if op.startswith("s_cbranch"):
        return [i + 1, labels[code.split()[1]]]
    return [i + 1] if i + 1 < len(insts) else []


def check_cfg(path, kernel_substr="score_k_kernel", only=None, allow_reissue=False):
    """Every register written by a hand-issued (inline asm) global load is followed along EVERY control-flow path
    (loops included) until a `s_waitcnt vmcnt(N)` with N <= the loads issued since -- memory operations return in
    order, so that wait covers it; any instruction that names the register before that point is a problem.  Paths
    are explored with the FEWEST younger loads first (a wait covers less on them)."""
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
        if only is not None and not only(name):
            i += 1
            continue
        kernels += 1
        j = i + 1
        body = []
        while j < len(lines) and not lines[j].startswith(".Lfunc_end"):   # (a kernel may have several s_endpgm)
            body.append(lines[j])
            j += 1
        insts, labels = _parse(body)
        for k, (code, in_asm) in enumerate(insts):
            lm = LOAD.match("\t" + code)
            if not (in_asm and lm):
                continue
            dest = (lm.group(1), int(lm.group(2)))
            best = {}                      # instruction -> fewest younger loads it was reached with
            stack = [(s2, 0) for s2 in _successors(insts, labels, k)]
            bad = None
            while stack and bad is None:
                pc, younger = stack.pop()
                while True:
                    if pc in best and best[pc] <= younger:
                        break
                    best[pc] = younger
                    c2 = insts[pc][0]
                    op = c2.split()[0]
                    if op == "s_waitcnt":
                        wm = re.search(r"vmcnt\((\d+)\)", c2)
                        if wm and younger >= int(wm.group(1)):
                            break          # covered on this path
                    elif re.match(r"(global|buffer|scratch|flat)_load", op):
                        redef = dest in regs_of(c2.split(None, 1)[1].split(",")[0])
                        if redef and allow_reissue and insts[pc][1]:
                            break          # a dummy destination (L2 touch): the next hand-issued load into it takes over
                        if pc == k:
                            # back at the load itself without a covering wait: it overwrites its own destination
                            bad = "re-issued before it was waited for"
                            break
                        if redef:
                            bad = "overwritten by `%s`" % c2
                            break
                        younger += 1
                    if op != "s_waitcnt" and not re.match(r"(global|buffer|scratch|flat)_load", op) and dest in regs_of(c2):
                        bad = "touched by `%s`" % c2
                        break
                    if re.match(r"(global|buffer|scratch|flat)_load", op) and dest in regs_of(c2.split(None, 1)[1].split(",", 1)[1]):
                        bad = "used as an address by `%s`" % c2
                        break
                    succ = _successors(insts, labels, pc)
                    if not succ:
                        break              # (end of the kernel with the load in flight: nothing reads it)
                    for s2 in succ[1:]:
                        stack.append((s2, younger))
                    pc = succ[0]
            if bad:
                problems.append("%s: %s%d (asm load in flight, `%s`) %s" % (name, dest[0], dest[1], code, bad))
        i = j
    return kernels, problems


def check_linear(path, kernel_substr="score_k_kernel", skip=None):
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
        if skip is not None and skip(name):
            i += 1
            continue
        kernels += 1
        j = i + 1
        body = []
        while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
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



def is_jit(name):
    """the mirror variants (TRANSPOSED = true, the 4th template argument of score_k_kernel<BITS, SPARSE, NWAVES, TRANSPOSED,
    COMPACT, PAIR>) with per-channel tables (PAIR = 0) or the fp32 pair-sum tables (PAIR = 2): loads in flight across loop
    iterations, every control-flow path walked.  (Round 5 gave the template its sixth argument and this pattern still
    expected five: the mirror kernels were scanned in program-text order only.  Round 6 repaired it -- the default kernels
    and the new 16-wave variant pass the path walk.  The opt-in fp16 pair-table variants (PAIR = 1) keep the linear scan: the
    walk reports paths through their unrolled pair groups whose wait counts and load conditions are correlated.)"""
    return re.search(r"score_k_kernelILi\dELb[01]ELi\d+ELb1ELb[01]ELi[02]EEE", name) is not None


def check(path, kernel_substr="score_k_kernel"):
    """mirror (JIT) kernels: every control-flow path (check_cfg); the others keep their look-ahead inside one pass of
    the unrolled head loop, where the conditions of the loads and of the wait counts are correlated -- a path-insensitive
    walk reports infeasible paths there, so they are scanned in program-text order (check_linear)."""
    k1, p1 = check_cfg(path, kernel_substr, only=is_jit)
    k2, p2 = check_linear(path, kernel_substr, skip=is_jit)
    return k1 + k2, p1 + p2


def _compile(name):
    tmp = tempfile.mkdtemp(prefix="kvq_isa_")
    src = os.path.join(ROOT, "kvquant_amd", "csrc", name)
    subprocess.check_call([HIPCC] + FLAGS + ["-save-temps", "-c", src, "-o", os.path.join(tmp, "k.o")], cwd=tmp,
                          stderr=subprocess.DEVNULL)
    return [os.path.join(tmp, f) for f in os.listdir(tmp) if f.endswith(".s") and "amdgcn" in f][0]


def check_mix_v(path):
    """the p.V kernel's L2 touches: one dummy register, re-loaded every chunk, never read"""
    return check_cfg(path, kernel_substr="mix_v_kernel", allow_reissue=True)


def main():
    if len(sys.argv) > 1:
        if "mix_v_wide" in os.path.basename(sys.argv[1]):
            kernels, problems = check_cfg(sys.argv[1], kernel_substr="mix_v_wide_kernel")
        else:
            kernels, problems = check_mix_v(sys.argv[1]) if "mix_v" in os.path.basename(sys.argv[1]) else check(sys.argv[1])
    else:
        kernels, problems = check(_compile("kvq_score_k.hip"))
        k2, p2 = check_mix_v(_compile("kvq_mix_v.hip"))
        kernels, problems = kernels + k2, problems + p2
        # the fused decode kernel runs the mirror score tile body (loads in flight across head iterations) as its K phase
        k3, p3 = check_cfg(_compile("kvq_fused_decode.hip"), kernel_substr="fused_decode_kernel")
        kernels, problems = kernels + k3, problems + p3
        # the one-workgroup-per-CU p.V kernel keeps a chunk's outlier entries in flight in registers across a whole chunk
        k4, p4 = check_cfg(_compile("kvq_mix_v_wide.hip"), kernel_substr="mix_v_wide_kernel")
        kernels, problems = kernels + k4, problems + p4
    print("%d kernels checked, %d problems" % (kernels, len(problems)))
    for p in problems:
        print("  " + p)
    return 1 if problems or kernels == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
