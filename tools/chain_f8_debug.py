"""Debug aid for the e4m3 feed-forward form of the chain kernel (x_mode bit 32): structured weights that separate the W1 path, the
bias / activation path and the W2 path.   python tools/chain_f8_debug.py   (GPU box)"""
import ctypes as C
import math
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cassnat_asr_public_amd import hip  # noqa: E402


def e4m3(t):
    return t.clamp(-448, 448).to(torch.float8_e4m3fn).float()


def layer_norm(x, a, b, eps=1e-6):
    mean = x.mean(-1, keepdim=True)
    std = x.std(-1, keepdim=True)
    return a * (x - mean) / (std + eps) + b


def run(name, x, a1, b1n, w1, b1, w2, b2, dff):
    M, d = x.shape
    hp = lambda t: C.c_void_p(t.contiguous().data_ptr())
    pow2 = lambda w: 2.0 ** math.floor(math.log2(448.0 / max(w.abs().max().item(), 1e-30))) if w.abs().max() > 0 else 1.0
    s1, s2 = pow2(w1), pow2(w2)
    xn8 = e4m3(layer_norm(x, a1, b1n) * 16.0)
    pre = F.linear(xn8, e4m3(w1 * s1)) / (s1 * 16.0) + b1
    h8 = e4m3(F.relu(pre) * 8.0)
    ffn = F.linear(h8, e4m3(w2 * s2)) / (s2 * 8.0) + b2
    xd = x.clone().cuda()
    keep = [t.contiguous() for t in (a1, b1n, w1, b1, w2, b2)]
    hip.check(hip.lib().cn_op_chain(C.c_void_p(xd.data_ptr()), None, d, None, None, hp(keep[0]), hp(keep[1]), hp(keep[2]), hp(keep[3]),
                                    hp(keep[4]), hp(keep[5]), None, None, None, None, None, d, M, dff, 0, 1e-6, 32, None))
    torch.cuda.synchronize()
    got = xd.cpu() - x
    err = (got - ffn).abs()
    print(f"{name:40s} max|ffn| {ffn.abs().max():.4f}  max err {err.max():.5f}  mean err {err.mean():.6f}   got[0,:4] {got[0,:4].tolist()} want {ffn[0,:4].tolist()}")
    bad = (err > 0.01).nonzero()
    if len(bad):
        rows = sorted(set(bad[:, 0].tolist()))
        cols = sorted(set(bad[:, 1].tolist()))
        print(f"      {len(bad)} bad elements; rows {rows[:40]}{'...' if len(rows) > 40 else ''}; cols {cols[:70]}{'...' if len(cols) > 70 else ''}")
        r, c = bad[0].tolist()
        # is a bad element another element of x?
        val = got[r, c] + x[r, c]
        hit = ((x - val).abs() < 1e-6).nonzero()
        print(f"      first bad ({r},{c}): x_out {val:.5f}, x_in {x[r, c]:.5f}, want {x[r, c] + ffn[r, c]:.5f}; x_in holds that value at {hit[:4].tolist()}")
    return got, ffn


def main():
    g = torch.Generator().manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g)
    M, d = 128, 256
    for dff in (256, 2048):
        x = rn(M, d) * 2 + 0.3
        a1, b1n = torch.ones(d), torch.zeros(d)
        z = torch.zeros
        print("d_ff", dff)
        # 1. hidden = relu(b1) with distinct exact values, W2 random: bias order, activation, W2 order
        b1 = ((torch.arange(dff) % 7).float() * 0.25)
        run("w1=0, b1 pattern, w2 random", x, a1, b1n, z(dff, d), b1, rn(d, dff) / 16, z(d), dff)
        # 2. one hidden unit alive at a time
        for j in (0, 33):
            b1 = z(dff); b1[j] = 1.0
            w2 = z(d, dff); w2[:, j] = torch.arange(d).float() / 256
            got, want = run(f"only hidden {j}", x, a1, b1n, z(dff, d), b1, w2, z(d), dff)
        # 3. W1 path: W2 = all-ones / 64 (order-free), b1 large so that ReLU is inactive
        run("w1 random, b1=4, w2 = 1/64", x, a1, b1n, rn(dff, d) / 64, torch.full((dff,), 4.0), torch.full((d, dff), 1 / 64), z(d), dff)
        # 4. only one input channel matters
        for c in (0, 64):
            w1 = z(dff, d); w1[:, c] = 0.25
            run(f"w1 only channel {c}", x, a1, b1n, w1, torch.full((dff,), 4.0), torch.full((d, dff), 1 / 64), z(d), dff)
        run("full random", x, 1 + 0.1 * rn(d), 0.1 * rn(d), rn(dff, d) / 16, 0.1 * rn(dff), rn(d, dff) / math.sqrt(dff), 0.1 * rn(d), dff)


if __name__ == "__main__":
    main()
