"""CPU: the oracle (C kernels + restated glue) reproduces the fixtures made from
the reference's own QuantK/QuantV classes bit-for-bit."""
import glob
import os

import pytest

from oracle.glue import OracleQuantK, OracleQuantV
from tests import scenario


def _paths():
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    return sorted(glob.glob(os.path.join(here, "ref_nuq*.npz")))


@pytest.mark.parametrize("path", _paths(), ids=lambda p: os.path.basename(p)[:-4])
def test_oracle_matches_reference_golden(path):
    g = scenario.load(path)
    out = scenario.replay(g, OracleQuantK, OracleQuantV)
    scenario.compare(g, out, exact_scores=True)


def test_golden_present():
    assert len(_paths()) >= 4
