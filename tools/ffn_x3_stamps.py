"""Where the workgroups of the split-bf16 FFN kernel (fused_x3.hip) spend their cycles: builds a measurement variant of
the library (-DFX_STAMPS: four s_memtime stamps per wave) beside the product one, runs the kernel on one layer of the
benchmark shape and prints the medians of prologue (LayerNorm + first requests), main loop and epilogue (cross-wave sum,
residual, next LayerNorm), in shader cycles (s_memtime; the counters of different XCDs are not aligned, so only per-wave differences mean anything).

    python tools/ffn_x3_stamps.py [M] [dff] [-DFX_EXP_...]         (GPU box)
"""
import ctypes as C
import os

MIX = int(os.environ.get("FFN_X3_MIX", "1"))  # (1: the mixed arithmetic the engine runs; 0: three bf16 MFMAs per product)
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cassnat_asr_public_amd import build as B  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
    dff = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    out = os.environ.get("TMPDIR", "/tmp")
    extra = sys.argv[3:]  # e.g. -DFX_EXP_NO_WLOAD / -DFX_EXP_NO_XREAD: timing experiments (wrong results)
    tag = "".join(c for c in "".join(extra) if c.isalnum())
    lib = B.build(extra_flags=["-DFX_STAMPS"] + extra, lib=os.path.join(out, "libcassnat_hip_stamps%s.so" % tag), objdir=os.path.join(out, "cn_stamps_obj" + tag))
    L = C.CDLL(lib)
    L.cn_op_ffn_x3.argtypes = [C.c_void_p] * 10 + [C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p]
    L.cn_debug_ffn_x3_stamps.argtypes = [C.c_void_p, C.c_int32]
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(M, 256, generator=g) * 2).cuda()
    a1, b1n, a2, b2n = (torch.randn(256, generator=g).cuda() for _ in range(4))
    w1, bb1 = torch.randn(dff, 256, generator=g) / 16, (torch.randn(dff, generator=g) * 0.1).cuda()
    w2, bb2 = torch.randn(256, dff, generator=g) / dff ** 0.5, (torch.randn(256, generator=g) * 0.1).cuda()
    xn = torch.zeros(M, 256, dtype=torch.int32, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    for _ in range(3):
        rc = L.cn_op_ffn_x3(p(x), p(a1), p(b1n), p(w1), p(bb1), p(w2), p(bb2), p(a2), p(b2n), p(xn), M, dff, 1e-6, MIX, None)
        assert rc == 0, rc
    n_wg = (M + 63) // 64
    st = np.zeros((n_wg, 4, 8), dtype=np.uint64)
    assert L.cn_debug_ffn_x3_stamps(st.ctypes.data_as(C.c_void_p), n_wg) == 0
    d = np.diff(st[:, :, :4].astype(np.int64), axis=2) / 1.0  # shader cycles
    for i, name in enumerate(("prologue", "main loop", "epilogue")):
        print("%-10s median %8.0f cycles   min %8.0f   max %8.0f" % (name, np.median(d[:, :, i]), d[:, :, i].min(), d[:, :, i].max()))
    tiles = dff // 128
    for i, (name, mf) in enumerate((("W1 blocks", 8 * 384), ("bias/ReLU (serial half)", 0), ("W2 blocks", 8 * 384))):
        print("  per hidden tile: %-24s %6.0f cycles (matrix pipe: %d)" % (name, np.median(st[:, :, 4 + i].astype(np.int64)) / tiles, mf))
    print("%d workgroups; main loop per block of 12 MFMAs (384 cycles of matrix pipe): %.0f cycles" % (n_wg, np.median(d[:, :, 1]) / (dff / 128 * 16)))


if __name__ == "__main__":
    main()
