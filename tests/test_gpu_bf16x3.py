"""GPU parity of the split-bf16 ("bf16x3") forms of the generic kernels, through the C ABI.

An element is a (bf16 hi, bf16 lo) pair - hi = bf16(v), lo = bf16(v - hi) - stored per group of 32 elements as 32 hi
halves then 32 lo halves (128 bytes: common.h `split_t`); a product is three bf16 MFMAs (hi.hi + hi.lo + lo.hi) with fp32
accumulation.  The references are plain torch fp32 / fp64 ops on the UNROUNDED inputs: what is compared is the claim that
this precision is parity-grade (relative error of a product ~2^-16; tolerances written at each check), not agreement with a
rounded emulation.  End-to-end: tests/test_gpu_pipeline.py::test_parity_gate[bf16x3-*].
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from cassnat_asr_public_amd import hip

pytestmark = pytest.mark.gpu
X3 = hip.PRECISION["bf16x3"]


def p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    return hip.current_stream()


def to_split(x):
    """fp32 tensor (last dim % 32 == 0) -> device buffer in the split layout (as int32 words: same byte count)."""
    x = x.contiguous().float().cuda()
    out = torch.empty(x.shape, dtype=torch.int32, device="cuda")
    hip.check(hip.lib().cn_op_convert(X3, p(x), p(out), x.numel(), 0, stream()))
    return out


def from_split(buf, shape=None):
    out = torch.empty(buf.shape if shape is None else shape, dtype=torch.float32, device="cuda")
    hip.check(hip.lib().cn_op_convert(X3, p(buf), p(out), out.numel(), 1, stream()))
    torch.cuda.synchronize()
    return out.cpu()


def relerr(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def host_split(x):
    """The definition: hi = bf16(v) (round to nearest even), lo = bf16(v - hi); bytes per 32-group: 32 hi, then 32 lo."""
    x = x.float().contiguous()
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    g = x.numel() // 32
    img = torch.stack([hi.reshape(g, 32), lo.reshape(g, 32)], 1).contiguous()  # (g, 2, 32) bf16 = 128 B per group
    return img.view(torch.int16).reshape(-1)


def test_convert_is_the_documented_layout_and_round_trips():
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.randn(1024, generator=g) * s for s in (1e-6, 1e-3, 1.0, 300.0)]).reshape(-1, 64)
    x[0, :4] = torch.tensor([0.0, -0.0, 1.0, 1.0 + 2.0 ** -9])
    dev_img = to_split(x)
    torch.cuda.synchronize()
    assert torch.equal(dev_img.cpu().view(torch.int16).reshape(-1), host_split(x))
    back = from_split(dev_img)
    # hi + lo carries >= 16 significant bits: |v - (hi + lo)| <= 2^-17 |v|
    assert ((back - x).abs() <= x.abs() * 2.0 ** -17 + 1e-38).all()


@pytest.mark.parametrize("M,N,K", [(300, 200, 256), (64, 64, 64), (1, 5000, 256), (4100, 2048, 128), (777, 256, 2048), (33, 96, 32)])
def test_gemm_bias(M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    ref = F.linear(A.double(), W.double(), bias.double())
    Ad, Wd, bd = to_split(A), to_split(W), bias.cuda()
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    hip.check(hip.lib().cn_op_gemm(X3, p(Ad), K, p(Wd), p(bd), p(out), N, 1, M, N, K, 0, None, 0, None, 1, 1.0, stream()))
    torch.cuda.synchronize()
    # three of the four partial products: the dropped lo.lo term and the two representation errors are each <= 2^-17 relative
    # to |a||w|; against the output's scale (sqrt(K) random-sign terms) that is a few 1e-6 - the bf16 path sits at 1e-2
    assert relerr(out, ref) < 3e-5
    if N % 32 == 0:  # split-format output (row stride a multiple of 32 elements)
        out2 = torch.zeros(M, N, dtype=torch.int32, device="cuda")
        hip.check(hip.lib().cn_op_gemm(X3, p(Ad), K, p(Wd), p(bd), p(out2), N, 0, M, N, K, 1, None, 0, None, 1, 1.0, stream()))
        assert relerr(from_split(out2), F.relu(ref)) < 3e-5


def test_gemm_epilogues_and_strided_operands():
    g = torch.Generator().manual_seed(3)
    M, N, K, period = 515, 256, 320, 103
    A = torch.randn(M, 2 * K, generator=g)  # the A operand is the right half of a wider buffer (as K|V / Q|K|V thirds are)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias, resid, pe = torch.randn(N, generator=g), torch.randn(M, N, generator=g), torch.randn(period, N, generator=g)
    lin = F.linear(A[:, K:].double(), W.double(), bias.double())
    Ad, Wd, bd = to_split(A), to_split(W), bias.cuda()
    a_right = C.c_void_p(Ad.data_ptr() + K * 4)  # column offset K (a multiple of 32 elements) = K * 4 bytes
    L = hip.lib()
    xres = resid.cuda()
    hip.check(L.cn_op_gemm(X3, a_right, 2 * K, p(Wd), p(bd), p(xres), N, 1, M, N, K, 0, p(xres), N, None, 1, 1.0, stream()))
    assert relerr(xres, resid.double() + lin) < 3e-5
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")
    ped = pe.cuda()
    hip.check(L.cn_op_gemm(X3, a_right, 2 * K, p(Wd), p(bd), p(out), N, 1, M, N, K, 0, None, 0, p(ped), period, 16.0, stream()))
    torch.cuda.synchronize()
    assert relerr(out, lin * 16.0 + pe.double()[torch.arange(M) % period]) < 3e-5


@pytest.mark.parametrize("M,d", [(1, 128), (1001, 256), (37, 512)])
def test_layernorm(M, d):
    from oracle.cassnat_oracle import layer_norm

    g = torch.Generator().manual_seed(d)
    x = torch.randn(M, d, generator=g) * 3 + 0.5
    a2, b2 = torch.randn(d, generator=g), torch.randn(d, generator=g)
    y = torch.zeros(M, d, dtype=torch.int32, device="cuda")
    xd, ad, bd = x.cuda(), a2.cuda(), b2.cuda()  # (named: a temporary's block would be reused by the next .cuda())
    hip.check(hip.lib().cn_op_layernorm(X3, p(xd), p(ad), p(bd), p(y), M, d, 1e-6, stream()))
    assert relerr(from_split(y), layer_norm(x, a2, b2)) < 1.6e-5  # the fp32 kernel's own 3e-6 + the 2^-17 of the element


@pytest.mark.parametrize("B,T,Fd,Cc", [(2, 61, 80, 128), (1, 8, 6, 64), (2, 61, 80, 256), (1, 9, 7, 256)])
def test_conv1_conv2(B, T, Fd, Cc):
    g = torch.Generator().manual_seed(B * T)
    x = torch.randn(B, T, Fd, generator=g)
    w1, b1 = torch.randn(Cc, 1, 3, 3, generator=g) / 3, torch.randn(Cc, generator=g) * 0.1
    w2, b2 = torch.randn(Cc, Cc, 3, 3, generator=g) / math.sqrt(9 * Cc), torch.randn(Cc, generator=g) * 0.1
    ref1 = F.relu(F.conv2d(x.unsqueeze(1), w1, b1, stride=2, padding=1))  # (B,C,T1,F1)
    T1, F1 = ref1.shape[2], ref1.shape[3]
    L = hip.lib()
    out1 = torch.zeros(B, T1, F1, Cc, dtype=torch.int32, device="cuda")
    xd, w9c, b1d, b2d = x.cuda(), w1.reshape(Cc, 9).t().contiguous().cuda(), b1.cuda(), b2.cuda()
    hip.check(L.cn_op_conv1(X3, p(xd), p(w9c), p(b1d), p(out1), B, T, Fd, Cc, stream()))
    assert relerr(from_split(out1).permute(0, 3, 1, 2), ref1) < 1e-5
    ref2 = F.relu(F.conv2d(ref1.double(), w2.double(), b2.double(), stride=2, padding=1))
    T2, F2 = ref2.shape[2], ref2.shape[3]
    out2 = torch.zeros(B, T2, F2, Cc, dtype=torch.int32, device="cuda")
    wk = to_split(w2.permute(0, 2, 3, 1).reshape(Cc, 9 * Cc))
    hip.check(L.cn_op_conv2(X3, p(out1), p(wk), p(b2d), p(out2), B, T1, F1, Cc, stream()))
    assert relerr(from_split(out2).permute(0, 3, 1, 2), ref2) < 3e-5


@pytest.mark.parametrize("B,T,Fd", [(2, 61, 80), (1, 9, 7), (3, 170, 80), (1, 1000, 80)])
def test_conv_frontend_mix_arithmetic(B, T, Fd):
    """The split-bf16 engine's conv front-end in the MIX arithmetic (conv1.hip MIXP planes, conv2.hip MIX): a product is
    half(a) half(b) + l_a q_b + q_a l_b with e4m3 cross-term factors at fixed power-of-two scales - one half-precision MFMA and two
    e4m3 MFMAs at twice the rate per 16 k where the X3 form spends three bf16 MFMAs - against the float64 convolution: the X3 form's
    tolerance (cn_op_conv2 above: 3e-5) holds."""
    Cc = 256
    g = torch.Generator().manual_seed(B * T + 7)
    x = torch.randn(B, T, Fd, generator=g)
    w1, b1 = torch.randn(Cc, 1, 3, 3, generator=g) / 3, torch.randn(Cc, generator=g) * 0.1
    w2, b2 = torch.randn(Cc, Cc, 3, 3, generator=g) / math.sqrt(9 * Cc), torch.randn(Cc, generator=g) * 0.1
    ref1 = F.relu(F.conv2d(x.unsqueeze(1), w1, b1, stride=2, padding=1))
    ref2 = F.relu(F.conv2d(ref1.double(), w2.double(), b2.double(), stride=2, padding=1))
    T2, F2 = ref2.shape[2], ref2.shape[3]
    out = torch.zeros(B, T2, F2, Cc, dtype=torch.int32, device="cuda")
    w2k = w2.permute(0, 2, 3, 1).contiguous()  # [C][3][3][C] on the host
    xd, w9c, b1d, b2d = x.cuda(), w1.reshape(Cc, 9).t().contiguous().cuda(), b1.cuda(), b2.cuda()
    hip.check(hip.lib().cn_op_conv_frontend_mix(p(xd), p(w9c), p(b1d), C.c_void_p(w2k.data_ptr()), p(b2d), p(out), None, B, T, Fd, Cc,
                                                stream()))
    got = from_split(out).permute(0, 3, 1, 2)
    err = relerr(got, ref2)
    print(f"[conv front-end, MIX arithmetic] B {B} T {T}: relative error {err:.2e}")
    assert err < 3e-5


# ----------------------------------------------------------------------------------------------- attention
def attention_reference(q, k, v, mask):
    scores = torch.einsum("bqhd,bkhd->bhqk", q.double(), k.double()) / 8.0
    scores = scores.masked_fill(mask.unsqueeze(1) == 0, float(np.finfo(np.float32).min))
    return torch.einsum("bhqk,bkhd->bqhd", torch.softmax(scores, dim=-1), v.double())


def run_attention(q, k, v, keymask=None, klen=None, intervals=None, causal=0):
    B, Lq, H, _ = q.shape
    Lk = k.shape[1]
    qd, kd, vd = to_split(q.reshape(B * Lq, H * 64)), to_split(k.reshape(B * Lk, H * 64)), to_split(v.reshape(B * Lk, H * 64))
    o = torch.zeros(B * Lq, H * 64, dtype=torch.int32, device="cuda")
    km = None if keymask is None else keymask.to(torch.uint8).cuda()
    kl = None if klen is None else klen.to(torch.int32).cuda()
    iv = None if intervals is None else intervals.to(torch.int32).cuda()
    hip.check(hip.lib().cn_op_attention(X3, p(qd), H * 64, p(kd), H * 64, p(vd), H * 64, p(o), H * 64, B, H, Lq, Lk, p(km), p(kl),
                                        p(iv), 0 if iv is None else intervals.shape[1], causal, 0.125, stream()))
    return from_split(o).reshape(B, Lq, H, 64)


ATT_TOL = 4e-5  # (fp32 kernel: 1e-5; bf16: 2e-2)


@pytest.mark.parametrize("Lq,Lk", [(250, 250), (50, 16), (171, 300), (129, 65)])
def test_attention_key_padding(Lq, Lk):
    g = torch.Generator().manual_seed(Lq * 7 + Lk)
    B, H = 3, 2
    q, k, v = (torch.randn(B, L, H, 64, generator=g) for L in (Lq, Lk, Lk))
    lens = torch.tensor([Lk, max(1, Lk * 2 // 3), max(1, Lk // 5)])
    keymask = torch.arange(Lk)[None, :] < lens[:, None]
    keymask[1, Lk // 3] = False
    assert relerr(run_attention(q, k, v, keymask=keymask), attention_reference(q, k, v, keymask[:, None, :])) < ATT_TOL


def test_attention_trigger_intervals_empty_rows_and_length_masks():
    g = torch.Generator().manual_seed(99)
    B, H, Lq, Lk = 2, 4, 70, 150
    q, k, v = (torch.randn(B, L, H, 64, generator=g) for L in (Lq, Lk, Lk))
    keymask = torch.ones(B, Lk, dtype=torch.bool)
    keymask[1, 120:] = False
    iv = torch.zeros(B, Lq + 3, 4, dtype=torch.int32)
    dense = torch.zeros(B, Lq, Lk, dtype=torch.bool)
    for b in range(B):
        edges = np.sort(np.random.default_rng(b).choice(np.arange(1, Lk), size=40, replace=False))
        lo = 0
        for u, hi in enumerate(edges):
            iv[b, u, 0], iv[b, u, 1] = lo, int(hi)
            dense[b, u, lo:hi] = True
            lo = int(hi)
        iv[b, 40, 2], iv[b, 40, 3] = 7, 8
        dense[b, 40, 7] = True
    dense &= keymask[:, None, :]
    got = run_attention(q, k, v, keymask=keymask, intervals=iv)
    assert relerr(got, attention_reference(q, k, v, dense)) < ATT_TOL
    assert relerr(got[0, 50], v[0].double().mean(0)) < ATT_TOL  # an empty row = the mean of V over every key (finite fill)
    U = 97
    q, k, v = (torch.randn(3, U, H, 64, generator=g) for _ in range(3))
    ylen = torch.tensor([97, 40, 1])
    for causal in (0, 1):
        mask = (torch.arange(U)[None, :] < ylen[:, None])[:, None, :].expand(3, U, U).clone()
        if causal:
            mask &= torch.tril(torch.ones(U, U, dtype=torch.bool))[None]
        assert relerr(run_attention(q, k, v, klen=ylen, causal=causal), attention_reference(q, k, v, mask)) < ATT_TOL


def test_attention_rescale_branch():
    """forces the online-softmax rescale: the row maximum jumps by ~80 in a late key tile"""
    g = torch.Generator().manual_seed(1)
    B, H, Lq, Lk = 1, 1, 40, 200
    q, k, v = (torch.randn(B, L, H, 64, generator=g) for L in (Lq, Lk, Lk))
    k[0, 170, 0] = q[0, 3, 0] * 10.0
    assert relerr(run_attention(q, k, v), attention_reference(q, k, v, torch.ones(B, 1, Lk, dtype=torch.bool))) < ATT_TOL


@pytest.mark.parametrize("mix", [0, 1])
@pytest.mark.parametrize("M,dff,with_next", [(8000, 2048, True), (45, 256, False), (2304, 2048, True), (64, 128, True), (1, 1024, True)])
def test_ffn_fused_x3(M, dff, with_next, mix):
    """fused_x3.hip: x += W2 relu(W1 LN(x) + b1) + b2 and the next LayerNorm, against fp64 on the unrounded operands.  mix = 1:
    the two products in the engine's mixed arithmetic (half-precision hi x hi + e4m3 cross terms, 2 MFMA units per product) -
    the same tolerance."""
    from oracle.cassnat_oracle import layer_norm

    g = torch.Generator().manual_seed(M + dff)
    x = torch.randn(M, 256, generator=g) * 2
    a1, b1n = torch.randn(256, generator=g) * 0.5 + 1, torch.randn(256, generator=g) * 0.2
    a2, b2n = torch.randn(256, generator=g) * 0.5 + 1, torch.randn(256, generator=g) * 0.2
    w1, bb1 = torch.randn(dff, 256, generator=g) / 16, torch.randn(dff, generator=g) * 0.1
    w2, bb2 = torch.randn(256, dff, generator=g) / math.sqrt(dff), torch.randn(256, generator=g) * 0.1
    xd = x.cuda()
    devs = [t.cuda() for t in (a1, b1n, bb1, bb2, a2, b2n)]
    xn = torch.zeros(M, 256, dtype=torch.int32, device="cuda")
    w1c, w2c = w1.contiguous(), w2.contiguous()
    hip.check(hip.lib().cn_op_ffn_x3(p(xd), p(devs[0]), p(devs[1]), C.c_void_p(w1c.data_ptr()), p(devs[2]), C.c_void_p(w2c.data_ptr()),
                                     p(devs[3]), p(devs[4]) if with_next else None, p(devs[5]) if with_next else None,
                                     p(xn) if with_next else None, M, dff, 1e-6, mix, stream()))
    torch.cuda.synchronize()
    xr = x.double()
    h = F.relu(F.linear(layer_norm(xr, a1.double(), b1n.double()), w1.double(), bb1.double()))
    ref = xr + F.linear(h, w2.double(), bb2.double())
    print(f"[ffn_x3 mix={mix}] M {M} dff {dff}: relative error {relerr(xd, ref):.2e}")
    assert relerr(xd, ref) < 2e-5
    if with_next:
        assert relerr(from_split(xn), layer_norm(ref, a2.double(), b2n.double())) < 3e-5


@pytest.mark.parametrize("mix", [0, 1])  # (1: the feed-forward products in the mixed arithmetic - what the engine runs)
@pytest.mark.parametrize("M,dff,with_ctx,tail_n", [(8000, 2048, True, 768), (8000, 2048, True, 0), (45, 256, True, 768), (2304, 2048, False, 256),
                                                   (65, 128, True, 512), (1, 1024, True, 768), (20031, 2048, True, 768)])
def test_x3_row_chain(M, dff, with_ctx, tail_n, mix):
    """fused_x3.hip, row-chain form: x += Wo ctx + bo; x += W2 relu(W1 LN(x) + b1) + b2; then the next LayerNorm or the next
    attention's projection of it - against fp64 on the unrounded operands (ctx as the split-bf16 values the attention kernel
    leaves).  Partial last tiles, one row, several rounds of workgroups."""
    from oracle.cassnat_oracle import layer_norm

    g = torch.Generator().manual_seed(3 * M + dff + tail_n)
    x = torch.randn(M, 256, generator=g) * 2
    ctx = torch.randn(M, 256, generator=g)
    wo, bo = torch.randn(256, 256, generator=g) / 16, torch.randn(256, generator=g) * 0.1
    a1, b1n = torch.randn(256, generator=g) * 0.5 + 1, torch.randn(256, generator=g) * 0.2
    a2, b2n = torch.randn(256, generator=g) * 0.5 + 1, torch.randn(256, generator=g) * 0.2
    w1, bb1 = torch.randn(dff, 256, generator=g) / 16, torch.randn(dff, generator=g) * 0.1
    w2, bb2 = torch.randn(256, dff, generator=g) / math.sqrt(dff), torch.randn(256, generator=g) * 0.1
    wt, bt = torch.randn(max(tail_n, 32), 256, generator=g) / 16, torch.randn(max(tail_n, 32), generator=g) * 0.1
    xd = x.cuda()
    ctx_s = to_split(ctx)
    ctx_v = from_split(ctx_s)  # the values the kernel reads
    devs = [t.cuda() for t in (bo, a1, b1n, bb1, bb2, a2, b2n, bt)]
    xn = torch.zeros(M, 256, dtype=torch.int32, device="cuda")
    tail = torch.zeros(M, max(tail_n, 32), dtype=torch.int32, device="cuda")
    hosts = [t.contiguous() for t in (wo, w1, w2, wt)]
    hp = lambda t: C.c_void_p(t.data_ptr())
    hip.check(hip.lib().cn_op_x3_chain(p(xd), p(ctx_s) if with_ctx else None, hp(hosts[0]), p(devs[0]), p(devs[1]), p(devs[2]), hp(hosts[1]),
                                       p(devs[3]), hp(hosts[2]), p(devs[4]), p(devs[5]), p(devs[6]), p(xn), hp(hosts[3]) if tail_n else None,
                                       p(devs[7]), p(tail), tail_n, M, dff, 1e-6, mix, stream()))
    torch.cuda.synchronize()
    xr = x.double()
    if with_ctx:
        xr = xr + F.linear(ctx_v.double(), wo.double(), bo.double())
    h = F.relu(F.linear(layer_norm(xr, a1.double(), b1n.double()), w1.double(), bb1.double()))
    ref = xr + F.linear(h, w2.double(), bb2.double())
    assert relerr(xd, ref) < 2e-5
    nxt = layer_norm(ref, a2.double(), b2n.double())
    if tail_n:
        assert relerr(from_split(tail), F.linear(nxt, wt.double(), bt.double())) < 4e-5
        assert int(xn.abs().max()) == 0  # LN_next(x) itself is not written in the tail form
    else:
        assert relerr(from_split(xn), nxt) < 3e-5
