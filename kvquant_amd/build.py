"""Builds libkvq.so (the C-ABI library of hand-written gfx950 kernels) in-tree.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the
GPU box with the gpurun snapshot.  Nothing here falls back to a CPU path.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libkvq.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wno-unused-value", "-fvisibility=hidden", "-DKVQ_BUILD"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def kernel_source_hash():
    """sha256 over the CODE of every kernel source of the library (// comments and white space stripped) -- every launch
    of the timed decode step comes from one of them: stamps measured per-kernel numbers that are kept in the tree
    (profiles/pmc_traffic.json), so that bench.py can tell when the kernels have changed since they were measured"""
    import hashlib
    import re
    h = hashlib.sha256()
    for f in sorted(x for x in os.listdir(CSRC) if x.endswith((".hip", ".h"))):
        with open(os.path.join(CSRC, f)) as fh:
            for line in fh:
                line = re.sub(r"\s+", "", re.sub(r"//.*", "", line))
                if line:
                    h.update(line.encode())
    return h.hexdigest()[:16]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "kvq.h"))
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "_obj"), exist_ok=True)
    for src in sources():
        obj = os.path.join(HERE, "_obj", os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _newer(obj, [src] + hdrs):
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(out.decode())
    if failed:
        raise RuntimeError("hipcc failed")
    if force or procs or _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
