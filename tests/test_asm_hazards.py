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


def test_checkers_on_hand_written_listings(tmp_path):
    """The two scans on listings small enough to read: what they must report and what they must not."""
    def listing(body):
        path = str(tmp_path / f"k{abs(hash(body))}.s")
        open(path, "w").write("_Z1kv:\n" + body + "\ts_endpgm\n")
        return path

    # a register copied while its LDS read is in flight; the same after the wait; a counted wait that covers the older read only
    assert pending_reg_check.scan(listing("\tds_read_b128 v[4:7], v0\n\tv_mov_b32_e32 v9, v5\n\ts_waitcnt lgkmcnt(0)\n"))
    assert not pending_reg_check.scan(listing("\tds_read_b128 v[4:7], v0\n\ts_waitcnt lgkmcnt(0)\n\tv_mov_b32_e32 v9, v5\n"))
    two = "\tds_read_b128 v[4:7], v0\n\tds_read_b128 v[8:11], v0 offset:16\n\ts_waitcnt lgkmcnt(1)\n"
    assert not pending_reg_check.scan(listing(two + "\tv_add_f32_e32 v1, v4, v4\n"))
    assert pending_reg_check.scan(listing(two + "\tv_add_f32_e32 v1, v8, v8\n"))
    # vector-memory loads count on vmcnt, in order with LDS-DMA requests (which have no destination register)
    assert pending_reg_check.scan(listing("\tglobal_load_dwordx4 a[0:3], v[2:3], off\n\tv_accvgpr_mov_b32 a8, a1\n\ts_waitcnt vmcnt(0)\n"))
    assert not pending_reg_check.scan(listing("\tglobal_load_dwordx4 v[4:7], v[2:3], off\n\tglobal_load_lds_dwordx4 v1, s[2:3]\n"
                                              "\ts_waitcnt vmcnt(1)\n\tv_mov_b32_e32 v9, v4\n"))
    # an 8-pass MFMA's result read after 2 instructions (needs 12 wait states on gfx950), after s_nop 11 (12 states), by a dependent MFMA
    mf = "\tv_mfma_f32_32x32x16_bf16 v[0:15], v[16:19], v[20:23], v[0:15]\n"
    assert mfma_hazard_check.scan(listing(mf + "\ts_mov_b32 s0, 0\n\tv_add_f32_e32 v30, v3, v3\n"))
    assert not mfma_hazard_check.scan(listing(mf + "\ts_nop 11\n\tv_add_f32_e32 v30, v3, v3\n"))
    assert not mfma_hazard_check.scan(listing(mf + mf))  # accumulating onto the same registers: back to back is fine
    assert mfma_hazard_check.scan(listing(mf + "\tv_mfma_f32_32x32x16_bf16 v[32:47], v[0:3], v[20:23], v[32:47]\n"))  # result as an A operand
    # the 16-pass e4m3 form needs 20
    m8 = "\tv_mfma_scale_f32_32x32x64_f8f6f4 a[0:15], v[16:23], v[24:31], a[0:15], v40, v41 op_sel_hi:[0,0,0]\n"
    assert mfma_hazard_check.scan(listing(m8 + "\ts_nop 15\n\tv_accvgpr_read_b32 v50, a3\n"))
    assert not mfma_hazard_check.scan(listing(m8 + "\ts_nop 15\n\ts_nop 3\n\tv_accvgpr_read_b32 v50, a3\n"))
