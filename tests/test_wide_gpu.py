"""The one-workgroup-per-CU p.V kernel (kvq_mix_v_wide.hip) takes a decode step's p.V from 6144 cached tokens on, so the
suites reach it only at their larger sizes.  This runs the p.V suites once more in a process where the geometry is forced
at EVERY length (KVQ_V_WIDE_FROM=1: the library reads it once per process) -- the oracle comparisons of tests/test_ops_gpu.py,
the head shards (head counts other than 32: partial unit groups), compact entries, sink tokens, ties -- plus
tests/wide_shapes_check.py (odd head counts, two unit groups, ragged lengths, both softmax modes)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*files, k=None):
    env = dict(os.environ, KVQ_V_WIDE_FROM="1")
    cmd = [sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"] + [os.path.join(ROOT, "tests", f) for f in files]
    if k:
        cmd += ["-k", k]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, env=env, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    return r.stdout


def test_wide_geometry_at_every_length_against_the_oracle():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = _run("test_ops_gpu.py", "test_compact_gpu.py", "test_ties_gpu.py", k="mix_v or compact or ties")
    assert " passed" in out


def test_wide_geometry_head_shards_and_sink_tokens():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = _run("test_head_shard_gpu.py", "test_decode_kv_gpu.py")
    assert " passed" in out


def test_wide_geometry_odd_shapes():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    out = _run("wide_shapes_check.py")
    assert " passed" in out
