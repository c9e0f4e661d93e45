import pytest
import torch

from tests import decode_check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bits,prefill,steps,max_len", [(4, 40, 4, 64), (4, 0, 5, 16), (3, 30, 3, 64), (2, 20, 3, 64)])
def test_decode_kv_matches_reference_pipeline(bits, prefill, steps, max_len):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    err = decode_check.run(torch.device("cuda:0"), bits=bits, prefill=prefill, steps=steps, max_len=max_len)
    assert err < 2e-3
