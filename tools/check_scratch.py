#!/usr/bin/env python
"""Static check of the built library: no kernel of the DEFAULT decode / append / prefill path may use scratch memory
(`private_segment_fixed_size` = register spills: a reload inside a hand-scheduled loop waits for vmcnt(0) and drains the
loads in flight).  Reads the metadata of the gfx950 code objects embedded in kvquant_amd/_obj/*.o (no recompilation).
Kernels that may spill are listed in ALLOWED with the reason; anything else fails __graft_entry__.build().

usage: python tools/check_scratch.py [-v]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("KVQ_LLVM_BIN", os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin"))
ARCH = os.environ.get("KVQ_ARCH", "gfx950")

# (regex on the demangled-ish kernel name, why scratch is tolerated there)
ALLOWED = [
    (r"score_k_kernelILi\dELb1ELi8ELb0ELb0ELi0E", "row-layout sparse score kernel behind kvq_score_k with the reference's outlier rows "
                                                  "(q_len > 1, direct C callers; decode_kv and, from 16K tokens, quant_cuda's _opt2 read a "
                                                  "token-contiguous mirror instead): 4-7 VGPRs outside the head loop"),
    (r"fused_decode_kernelILi2E", "fused attend at 2 bit (opt-in route, never a default)"),
]


def kernels_of(obj):
    """[(name, vgprs, vgpr_spills, scratch_bytes)] of the gfx950 code object embedded in a host object file"""
    with tempfile.TemporaryDirectory() as t:
        fat, co = os.path.join(t, "f.bin"), os.path.join(t, "k.co")
        subprocess.check_call([LLVM + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat])
        if not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return []
        subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o",
                               "--targets=hipv4-amdgcn-amd-amdhsa--" + ARCH, "--input=" + fat, "--output=" + co],
                              stderr=subprocess.DEVNULL)
        notes = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", co]).decode()
    out, cur = [], {}
    for ln in notes.splitlines():
        m = re.match(r"\s*-?\s*\.(name|vgpr_count|vgpr_spill_count|private_segment_fixed_size):\s*(\S+)", ln)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key == "name" and not val.startswith("_Z") and not val.endswith("_kernel"):
            continue                      # (argument names)
        cur[key] = val
        if key == "vgpr_spill_count" or (key == "vgpr_count" and "vgpr_spill_count" in cur):
            pass
        if all(k in cur for k in ("name", "vgpr_count", "vgpr_spill_count", "private_segment_fixed_size")):
            out.append((cur["name"], int(cur["vgpr_count"]), int(cur["vgpr_spill_count"]), int(cur["private_segment_fixed_size"])))
            cur = {}
    return out


def check(verbose=False):
    objs = sorted(glob.glob(os.path.join(ROOT, "kvquant_amd", "_obj", "*.o")))
    srcs = {os.path.basename(p)[:-4] for p in glob.glob(os.path.join(ROOT, "kvquant_amd", "csrc", "*.hip"))}
    n, problems, tolerated = 0, [], []
    for o in objs:
        if os.path.basename(o)[:-2] not in srcs:
            continue                      # (a stale object of a source that left the library)
        for name, vg, sp, scratch in kernels_of(o):
            n += 1
            if verbose:
                print("%-90s vgpr %3d spill %3d scratch %4d" % (name[:90], vg, sp, scratch))
            if scratch == 0:
                continue
            why = next((w for pat, w in ALLOWED if re.search(pat, name)), None)
            (tolerated if why else problems).append("%s: %d B of scratch (%d VGPR spills)%s" % (name, scratch, sp, " -- " + why if why else ""))
    return n, problems, tolerated


def main():
    # another ROCm layout / no LLVM tools: nothing to read the code objects with -- say so instead of failing the build
    if not all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")):
        print("0 kernels checked: LLVM tools not found under %s (set KVQ_LLVM_BIN); scratch check SKIPPED" % LLVM)
        return 0
    n, problems, tolerated = check("-v" in sys.argv)
    print("%d kernels, %d with scratch in a default path, %d tolerated" % (n, len(problems), len(tolerated)))
    for t in tolerated:
        print("  tolerated: " + t)
    for p in problems:
        print("  PROBLEM: " + p)
    return 1 if problems or n == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
