"""Builds oracle/_ref/quant_cuda_ref*.so: the REFERENCE's own extension
(/root/reference/deployment/kvquant/quant_cuda.cpp + quant_cuda_kernel.cu),
compiled for gfx950 from the sources where they lie.

TEST INFRASTRUCTURE ONLY.  The .so is the checker that pins the C oracle
(oracle/kvq_oracle.c) and the HIP kernels (kvquant_amd/libkvq.so) to the running
reference; nothing under kvquant_amd/ or any timed region imports it
(tests/test_abi_cpu.py enforces that).  It is git-ignored (no reference source
or binary enters history) but not gpurun-ignored: it travels to the GPU box,
where /root/reference does not exist.

How: the two reference files are CUDA; torch-ROCm's cpp_extension translates
CUDA runtime spellings to HIP ones at build time ("hipify") and drives hipcc.
The translated copies live only in a temporary build directory outside the repo
and are deleted afterwards.  The reference's kernels use no warp-level
intrinsics (only atomicAdd and <<<>>> launches), so their semantics carry over
to 64-wide wavefronts unchanged.  hipcc cross-compiles without a GPU.

Usage:  python -m oracle.build_ref            (no-op when up to date or when /root/reference is absent)
"""
import glob
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
REF_DIR = os.environ.get("KVQ_REFERENCE", "/root/reference/deployment/kvquant")
SOURCES = ["quant_cuda.cpp", "quant_cuda_kernel.cu"]
NAME = "quant_cuda_ref"


def built():
    """path of the built reference extension, or None"""
    hits = sorted(glob.glob(os.path.join(OUT, NAME + "*.so")))
    return hits[0] if hits else None


def available():
    return all(os.path.exists(os.path.join(REF_DIR, s)) for s in SOURCES)


def build(force=False, verbose=False):
    so = built()
    if not available():
        return so            # GPU box / no reference checkout: use what travelled with the snapshot
    srcs = [os.path.join(REF_DIR, s) for s in SOURCES]
    if so and not force and all(os.path.getmtime(so) >= os.path.getmtime(s) for s in srcs):
        return so
    os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
    os.environ.setdefault("MAX_JOBS", "8")
    from torch.utils import cpp_extension
    tmp = tempfile.mkdtemp(prefix="kvq_ref_build_")
    try:
        # the reference tree is read-only and hipify writes next to its inputs: work on scratch copies
        work = [shutil.copy(s, tmp) for s in srcs]
        bdir = os.path.join(tmp, "build")
        os.makedirs(bdir)
        # the module name is baked into the binary (PYBIND11_MODULE(TORCH_EXTENSION_NAME, ...)): a name of
        # its own keeps it from ever shadowing kvquant_amd's drop-in `quant_cuda`
        cpp_extension.load(name=NAME, sources=work, build_directory=bdir, verbose=verbose,
                           extra_cflags=["-O2"], extra_cuda_cflags=["-O2"],   # hipcc defaults, like nvcc's
                           is_python_module=False)
        os.makedirs(OUT, exist_ok=True)
        for old in glob.glob(os.path.join(OUT, NAME + "*.so")):
            os.remove(old)
        made = glob.glob(os.path.join(bdir, NAME + "*.so"))
        if not made:
            raise RuntimeError("reference extension did not produce a .so")
        dst = os.path.join(OUT, os.path.basename(made[0]))
        shutil.copy(made[0], dst)
        return dst
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def load():
    """import the built reference extension as a python module (GPU tests only)"""
    so = built()
    if so is None:
        raise ImportError("oracle/_ref is not built (python -m oracle.build_ref, needs /root/reference)")
    import importlib.util
    import torch  # noqa: F401  (the extension links against libtorch)
    spec = importlib.util.spec_from_file_location(NAME, so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
