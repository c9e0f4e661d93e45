"""GPU: kvquant_amd.QuantK/QuantV (HIP kernels through the C ABI) replay the
fixtures generated from the reference's own classes: packed caches, LUTs and
outlier rows bit-exact, scores / outputs within the north-star tolerance."""
import glob
import os

import pytest
import torch

from tests import scenario

pytestmark = pytest.mark.gpu


def _paths():
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return sorted(glob.glob(os.path.join(here, "ref_nuq*.npz")))


@pytest.mark.parametrize("host_topk", [True, False], ids=["topk_args", "topk_gpu"])
@pytest.mark.parametrize("path", _paths(), ids=lambda p: os.path.basename(p)[:-4])
def test_cache_classes_match_reference_golden(path, host_topk):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from kvquant_amd.cache import QuantK, QuantV
    g = scenario.load(path)
    out = scenario.replay(g, QuantK, QuantV, device="cuda", v_topk_on_host=host_topk)
    # with selection on the GPU only boundary TIES may be resolved differently from torch.topk
    scenario.compare(g, out, v_ties_ok=not host_topk)
