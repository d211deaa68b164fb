"""A hazard hipcc cannot see, checked on the compiled code: the kernels issue LDS reads / global loads from inline asm and wait
for them in a LATER asm statement, so that the latency is covered by other work.  In between, the destination registers are
ordinary values to the compiler - under register pressure it copies or spills them (before the data has arrived) or, when the
value is never used, hands the registers to something else (which the late data then overwrites).  Two real instances were
found this way (chain.hip's LayerNorm table reads in the e4m3 form; conv2.hip's unused last fragments):
tools/pending_reg_check.py scans the gfx950 assembly of every source that uses the idiom; tools/mfma_hazard_check.py does the same
for the wait states between an MFMA issued from asm and the first other instruction that touches its result."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))
import mfma_hazard_check  # noqa: E402
import pending_reg_check  # noqa: E402

SOURCES = ["chain", "conv2", "attention", "fused", "fused_x3", "gemm", "genmax", "proj_x3", "ast"]


def test_no_instruction_touches_a_register_with_a_load_in_flight(tmp_path):
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    csrc = os.path.join(REPO, "cassnat_asr_public_amd", "csrc")

    def asm(name):
        out = str(tmp_path / (name + ".s"))
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, os.path.join(csrc, name + ".hip")],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return out

    with ThreadPoolExecutor(max_workers=4) as ex:
        listings = list(ex.map(asm, SOURCES))
    found = [f for path in listings for f in pending_reg_check.scan(path)]
    assert not found, "\n".join(found[:20])
    # the second invisible hazard: an MFMA written in asm whose result compiler-scheduled code touches before the matrix pipe
    # has delivered it (no hardware interlock; a block followed by such code must end with the wait states itself)
    found = [f for path in listings for f in mfma_hazard_check.scan(path)]
    assert not found, "\n".join(found[:20])
    # (the scan is not vacuous: without the waits the chain kernel's blocks end with, it reports)
    chain = open(listings[SOURCES.index("chain")]).read()
    bad = str(tmp_path / "chain_without_drain.s")
    open(bad, "w").write(chain.replace("s_nop 13", "s_nop 1"))
    assert mfma_hazard_check.scan(bad)
