"""CPU: the C-ABI library builds, loads, and exports every symbol that
include/kvq.h declares (no compute calls without a GPU); host-side argument
checking of the Python shim; legacy quant_cuda surface."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "kvq.h")).read()
    return sorted(set(re.findall(r"KVQ_API[^;(]*?\b(kvq_\w+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from kvquant_amd import build
    build.build()
    from kvquant_amd import _lib
    return _lib.lib()


def test_header_symbols_exported(lib):
    names = _declared()
    assert len(names) >= 15
    raw = ctypes.CDLL(os.path.join(ROOT, "kvquant_amd", "libkvq.so"))
    for n in names:
        assert hasattr(raw, n), "libkvq.so does not export " + n
    from kvquant_amd import _lib
    assert sorted(_lib.SIGNATURES) == names, "python binding table and include/kvq.h disagree"


def test_version_and_errors(lib):
    from kvquant_amd import _lib
    assert lib.kvq_version() // 100 == _lib.ABI_MAJOR == 4
    assert b"invalid" in lib.kvq_strerror(-1)
    # argument validation happens before any launch, so these are safe without a GPU
    assert lib.kvq_append_k(5, None, None, None, 32, 128, 16, 0, None) == -1
    assert lib.kvq_score_k(4, None, None, None, None, 1, 32, 128, 10, 16, 10000.0, 0, None, None, 0, 0, None, 0,
                           None) == -1
    assert lib.kvq_mix_v_workspace_bytes(4, 1, 32, 128, 131072) > 0
    assert lib.kvq_mix_v_workspace_bytes(4, 1, 32, 64, 131072) == 0      # head_dim must be 128
    assert lib.kvq_score_k_workspace_bytes(4, 1, 32) == 32 * (16384 + 128 * 4)   # tables + fp32 copy of q


def test_score_launch_planner(lib):
    """host logic of the score launches (kvq_score_k_head_groups): the head groups per token tile for LLaMA-2-7B's 32
    heads on 512 workgroup slots -- the choices measured in profiles/r03_b_planner.txt"""
    g = lambda tiles, H=32, q_len=1, mx=32, slots=512: lib.kvq_score_k_head_groups(H, tiles, q_len, mx, slots)  # noqa: E731
    assert g(512) == 1            # 128K: one workgroup per tile, all 32 heads, the chip exactly full
    assert g(1024) == 1           # 256K: two full generations of them
    assert g(384) == 1            # 96K: a partial generation of whole tiles (round counting cut it into 1536 pieces)
    assert g(640) == 2            # 160K: 1280 half-tile workgroups instead of two rounds of 32 heads
    assert g(128) == 4            # 32K: four groups of 8 heads fill the chip
    assert g(64) == 8
    for tiles in (1, 3, 17, 100, 513, 4096, 40000):
        for H, mx in ((32, 32), (40, 32), (8, 8), (128, 32)):
            d = g(tiles, H=H, mx=mx)
            assert d >= 1 and H % d == 0 and H // d <= mx, (tiles, H, d)
    assert g(0) == 0 and g(512, H=0) == 0 and g(512, slots=0) == 0     # invalid arguments


def test_legacy_module_surface():
    from kvquant_amd import quant_cuda
    from oracle import quant_cuda_ref
    ours = {n for n in dir(quant_cuda) if n.startswith("vecquant")}
    ref = set(quant_cuda_ref.NAMES)
    assert ours == ref and len(ours) == 34


def test_shim_rejects_cpu_tensors():
    from kvquant_amd import quant_cuda
    m = torch.zeros(32, 16, 8, dtype=torch.int32)
    with pytest.raises(ValueError):
        quant_cuda.vecquant4appendvecK(m, torch.zeros(32, 128, 16), torch.zeros(4096), 0)


def test_no_oracle_import_in_product():
    """the product package must never import the oracle (test infrastructure)"""
    pkg = os.path.join(ROOT, "kvquant_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "quant_cuda_ref" not in txt and "build_ref" not in txt and "oracle/_ref" not in txt, f
    # bench.py may use the oracle only inside cpu_baseline(); the reference build never
    b = open(os.path.join(ROOT, "bench.py")).read()
    assert "build_ref" not in b and "quant_cuda_ref" not in b
