"""Error of the row-chain kernel against an fp64 reference with the kernel's roundings, per case (diagnostic):
    CASSNAT_HIP_LIB=ab/libA.so python tools/chain_err.py"""
import ctypes as C
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cassnat_asr_public_amd import hip  # noqa: E402


def ln(x, a, b, eps=1e-6):
    m = x.mean(-1, keepdim=True)
    s = x.std(-1, keepdim=True)
    return a * (x - m) / (s + eps) + b


def r16(x):
    return x.to(torch.bfloat16).double()


def main():
    d = 256
    for (M, dff, swish) in ((129, 256, 0), (129, 256, 1), (129, 128, 1), (129, 512, 1), (129, 2048, 1), (129, 2048, 0), (8000, 2048, 1), (8000, 2048, 0)):
        g = torch.Generator().manual_seed(M + dff)
        rn = lambda *s: torch.randn(*s, generator=g)
        x = rn(M, d) * 2 + 0.3
        ctx = rn(M, d)
        wo, bo = (rn(d, d) / 16).contiguous(), 0.1 * rn(d)
        a1, b1n = 1 + 0.1 * rn(d), 0.1 * rn(d)
        w1, b1 = (rn(dff, d) / 16).contiguous(), 0.1 * rn(dff)
        w2, b2 = (rn(d, dff) / math.sqrt(dff)).contiguous(), 0.1 * rn(d)
        ref = x.double() + F.linear(r16(ctx), r16(wo), bo.double())
        xn = r16(ln(ref, a1.double(), b1n.double()).float())
        pre = F.linear(xn, r16(w1), b1.double())
        h = r16((F.silu(pre) if swish else F.relu(pre)).float())
        # per hidden tile contribution, to see WHICH tile is off
        full = ref + F.linear(h, r16(w2), b2.double())
        xd = x.clone().cuda()
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        hp = lambda t: C.c_void_p(t.data_ptr())
        hip.check(hip.lib().cn_op_chain(p(xd), p(ctx.to(torch.bfloat16).cuda()), d, hp(wo), hp(bo), hp(a1), hp(b1n), hp(w1), hp(b1), hp(w2), hp(b2),
                                        None, None, None, None, None, d, M, dff, 0, 1e-6, 8 if swish else 0, hip.current_stream()))
        torch.cuda.synchronize()
        got = xd.cpu().double()
        err = (got - full).abs()
        # least-squares attribution of the residual to hidden tiles: residual ~ sum_t (delta_h_t . W2_t)
        resid = got - full
        w2t = r16(w2)  # [d][dff]
        sol = torch.linalg.lstsq(w2t, resid.T).solution.T  # [M][dff]: implied error of every hidden unit
        per_tile = sol.abs().view(M, dff // 32, 32).amax(dim=(0, 2))
        print(f"M={M} dff={dff} swish={swish}: max abs err {err.max():.5f} rel {err.max() / full.abs().max():.2e}; implied hidden-unit error per tile "
              f"{[round(float(v), 3) for v in per_tile[:12]]}", flush=True)


if __name__ == "__main__":
    main()
