"""The score kernel issues its look-ahead loads from inline asm, outside hipcc's vmcnt scoreboard.
That is only correct if the generated code never copies, spills or overwrites a destination register
while its load is in flight; tools/check_isa.py verifies it on the assembly hipcc produces for every
template variant (cross-compiles for gfx950, no GPU needed)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_no_register_touched_while_its_asm_load_is_in_flight():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_isa.py")], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 problems" in r.stdout


def test_checker_flags_a_spilled_in_flight_register(tmp_path):
    """the checker itself: a spill right after an asm load must be reported"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_isa
    bad = tmp_path / "bad.s"
    bad.write_text("\n".join([
        "_ZN3kvq14score_k_kernelILi4ELb1ELi8ELb0EEEvNS_10ScoreKArgsENS_9RopeFreqsE:",
        "\t;;#ASMSTART", "\tglobal_load_dword v3, v2, s[48:49]", "\t;;#ASMEND",
        "\tscratch_store_dword off, v3, off", "\ts_waitcnt vmcnt(0)", "\ts_endpgm", ""]))
    n, problems = check_isa.check(str(bad))
    assert n == 1 and len(problems) == 1
    good = tmp_path / "good.s"
    good.write_text("\n".join([
        "_ZN3kvq14score_k_kernelILi4ELb1ELi8ELb0EEEvNS_10ScoreKArgsENS_9RopeFreqsE:",
        "\t;;#ASMSTART", "\tglobal_load_dword v3, v2, s[48:49]", "\t;;#ASMEND",
        "\tv_add_f32_e32 v4, v5, v6", "\t;;#ASMSTART", "\ts_waitcnt vmcnt(0)", "\t;;#ASMEND",
        "\tscratch_store_dword off, v3, off", "\ts_endpgm", ""]))
    n, problems = check_isa.check(str(good))
    assert n == 1 and not problems


def test_no_scratch_in_default_path_kernels():
    """every kernel of the built library: scratch memory (register spills / stack objects) only where tools/check_scratch.py
    tolerates it by name -- the legacy row-layout instantiations -- and nowhere in a default decode / append / prefill path"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_scratch
    from kvquant_amd import build as kb
    kb.build()
    n, problems, tolerated = check_scratch.check()
    assert n > 50, n
    assert not problems, problems
    assert len(tolerated) <= 6, tolerated
