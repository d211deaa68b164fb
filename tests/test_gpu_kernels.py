"""GPU parity of every hand-written kernel, driven through the C ABI (cn_op_*).

Float kernels are compared with a plain torch fp32 reference of the same op on the host; tolerances are
written next to each check.  Integer kernels (CTC alignment, greedy pack) must be bit-exact with the oracle.
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from cassnat_asr_public_amd import hip

pytestmark = pytest.mark.gpu

PRECS = ["fp32", "bf16"]
# fp32 path = exact-f32 MFMA: only summation order differs from the reference.  bf16 path: inputs are
# rounded to bf16 (8 significant bits) and the reference is computed on the same rounded inputs, so what
# is left is accumulation order and the bf16 rounding of the output.
RTOL = {"fp32": 2e-5, "bf16": 1.2e-2}


def dev(x, dtype=None):
    t = torch.as_tensor(x)
    if dtype is not None:
        t = t.to(dtype)
    return t.contiguous().cuda()


def tdtype(prec):
    return torch.float32 if prec == "fp32" else torch.bfloat16


def p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    return hip.current_stream()


def relerr(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def rounded(x, prec):
    return x.to(tdtype(prec)).float()


# ----------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("M,N,K", [(300, 200, 256), (64, 64, 64), (1, 5000, 256), (4100, 2048, 128), (777, 256, 2048)])
def test_gemm_bias(prec, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    ref = F.linear(rounded(A, prec), rounded(W, prec), bias)
    for c_f32 in (1, 0):
        out = torch.full((M, N), float("nan"), dtype=torch.float32 if c_f32 else tdtype(prec), device="cuda")
        Ad, Wd, bd = dev(A, tdtype(prec)), dev(W, tdtype(prec)), dev(bias)
        hip.check(hip.lib().cn_op_gemm(hip.PRECISION[prec], p(Ad), K, p(Wd), p(bd), p(out), N, c_f32, M, N, K, 0, None, 0,
                                       None, 1, 1.0, stream()))
        torch.cuda.synchronize()
        assert relerr(out, ref) < RTOL[prec] * (1 if c_f32 or prec == "fp32" else 1.5)


@pytest.mark.parametrize("B,T,Fd", [(5, 77, 80), (1, 1, 3), (32, 1500, 80)])
def test_cmvn_on_the_device_equals_numpy_float64(B, T, Fd):
    """cn_op_cmvn (the pipelined decoder's global CMVN, behind the host-to-device copy): float((double(x) - mean) / std) on the frames
    of each utterance - SpeechDataset's numpy arithmetic (src/data/speech_loader.py:109-115, 147-149, 340) BIT FOR BIT, padding
    frames untouched."""
    rng = np.random.default_rng(B + T)
    x = (rng.standard_normal((B, T, Fd)) * 7 + 2).astype(np.float32)
    lens = rng.integers(0, T + 1, size=B).astype(np.int32)
    lens[0] = T
    for b in range(B):
        x[b, lens[b]:] = 0.0
    mean, std = rng.standard_normal(Fd) * 3, np.abs(rng.standard_normal(Fd)) * 2 + 0.1
    want = x.copy()
    for b in range(B):
        want[b, : lens[b]] = ((x[b, : lens[b]] - mean) / std).astype(np.float32)  # float32 - float64 -> float64, as in the dataset
    xd = torch.from_numpy(x).cuda()
    hip.cmvn_(xd, torch.from_numpy(lens).cuda(), torch.from_numpy(mean).cuda(), torch.from_numpy(std).cuda())
    torch.cuda.synchronize()
    np.testing.assert_array_equal(xd.cpu().numpy(), want)


def test_quantize_fp8_matches_torch_e4m3fn():
    g = torch.Generator().manual_seed(2)
    x = torch.cat([torch.randn(4096, generator=g) * s for s in (1e-4, 1e-3, 1e-2, 0.1, 1.0, 10.0, 40.0)]).view(-1, 256)
    x[0, :8] = torch.tensor([0.0, -0.0, 28.0, -28.0, 29.0, 0.001953125 / 16, 0.0009765625 / 16, 0.0146484375 / 16])
    xb = x.to(torch.bfloat16)
    out = torch.zeros(x.shape, dtype=torch.uint8, device="cuda")
    hip.check(hip.lib().cn_op_quantize_fp8(p(xb.cuda()), 256, p(out), x.shape[0], 256, 16.0, stream()))
    torch.cuda.synchronize()
    ref = (xb.float() * 16.0).clamp(-448, 448).to(torch.float8_e4m3fn).view(torch.uint8)
    bad = (out.cpu() != ref).nonzero()
    assert len(bad) == 0, [(xb[i, j].item(), hex(out[i, j].item()), hex(ref[i, j].item())) for i, j in bad[:8].tolist()]


@pytest.mark.parametrize("M,N,K,relu", [(300, 768, 256, 0), (8000, 2048, 256, 1), (515, 256, 2048, 0), (1, 256, 128, 0)])
def test_gemm_fp8_matches_an_e4m3_emulation(M, N, K, relu):
    """BASELINE config 5's product: both operands rounded to OCP e4m3fn (torch.float8_e4m3fn on the CPU is the emulation:
    round to nearest even; the kernel side saturates at 448) at per-tensor power-of-two scales, fp32 accumulation.  Also pins
    the host-side weight quantiser and the device-side activation quantiser against torch's conversion."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g) * 1.5
    A[0, :4] = torch.tensor([40.0, -40.0, 27.9, 1e-4])  # beyond the range at scale 16 (saturates at 28), near it, tiny
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    a_scale = 16.0
    Ab = A.to(torch.bfloat16)
    Ad, bd = Ab.cuda(), dev(bias)
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")
    ws = C.c_float()
    hip.check(hip.lib().cn_op_gemm_fp8(p(Ad), K, C.c_void_p(W.data_ptr()), p(bd), p(out), M, N, K, a_scale, relu, C.byref(ws), stream()))
    torch.cuda.synchronize()
    w_scale = ws.value
    assert w_scale == 2.0 ** math.floor(math.log2(448.0 / W.abs().max().item()))
    q = lambda t: t.clamp(-448, 448).to(torch.float8_e4m3fn).float()
    ref = F.linear(q(Ab.float() * a_scale), q(W * w_scale)) / (a_scale * w_scale) + bias
    if relu:
        ref = F.relu(ref)
    # identical operands (the quantisers are byte-exact, see test_quantize_fp8_...): what is left is the matrix core's own
    # summation of the 16 products of an fp8 MFMA, which is not a chain of fp32 adds (measured 3e-5 of the output range)
    assert relerr(out, ref) < 1e-4
    # and the whole thing is a sane approximation of the unquantised product where nothing saturates
    full = F.linear(Ab.float(), W) + bias
    if relu:
        full = F.relu(full)
    if M > 1:
        assert relerr(out[1:], full[1:]) < 0.08


@pytest.mark.parametrize("prec", PRECS)
def test_gemm_epilogues(prec):
    g = torch.Generator().manual_seed(3)
    M, N, K, period = 515, 256, 320, 103
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    resid = torch.randn(M, N, generator=g)
    pe = torch.randn(period, N, generator=g)
    lin = F.linear(rounded(A, prec), rounded(W, prec), bias)
    Ad, Wd, bd = dev(A, tdtype(prec)), dev(W, tdtype(prec)), dev(bias)
    L = hip.lib()
    # ReLU
    out = torch.empty(M, N, dtype=tdtype(prec), device="cuda")
    hip.check(L.cn_op_gemm(hip.PRECISION[prec], p(Ad), K, p(Wd), p(bd), p(out), N, 0, M, N, K, 1, None, 0, None, 1, 1.0, stream()))
    assert relerr(out, F.relu(lin)) < RTOL[prec] * 1.5
    # residual, in place (C aliases resid), fp32 stream
    xres = dev(resid)
    hip.check(L.cn_op_gemm(hip.PRECISION[prec], p(Ad), K, p(Wd), p(bd), p(xres), N, 1, M, N, K, 0, p(xres), N, None, 1, 1.0, stream()))
    assert relerr(xres, resid + lin) < RTOL[prec]
    # embedding epilogue: (acc + bias) * sqrt(d) + pe[m % period]
    out = torch.empty(M, N, dtype=torch.float32, device="cuda")
    ped = dev(pe)
    hip.check(L.cn_op_gemm(hip.PRECISION[prec], p(Ad), K, p(Wd), p(bd), p(out), N, 1, M, N, K, 0, None, 0, p(ped), period, 16.0, stream()))
    ref = lin * 16.0 + pe[torch.arange(M) % period]
    assert relerr(out, ref) < RTOL[prec]


@pytest.mark.parametrize("M,K,period,lda_pad", [(515, 1024, 103, 0), (8000, 5120, 250, 0), (257, 1280, 257, 64), (31, 2048, 7, 0)])
def test_gemm_embed_deep_k(M, K, period, lda_pad):
    # linear_out shape class: bf16, N = 256, K >= 1024 goes to the LDS-DMA tile kernel (conv2.hip, LINEAR variant);
    # ragged M and a padded row stride included
    g = torch.Generator().manual_seed(M + K)
    N, lda = 256, K + lda_pad
    A = torch.randn(M, lda, generator=g)
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g)
    lin = F.linear(rounded(A[:, :K], "bf16"), rounded(W, "bf16"), bias)
    Ad, Wd, bd = dev(A, torch.bfloat16), dev(W, torch.bfloat16), dev(bias)
    out = torch.full((M + 1, N), float("nan"), dtype=torch.float32, device="cuda")
    pe = torch.randn(period, N, generator=g)
    ped = dev(pe)
    ref = lin * 16.0 + pe[torch.arange(M) % period]
    hip.check(hip.lib().cn_op_gemm(hip.PRECISION["bf16"], p(Ad), lda, p(Wd), p(bd), p(out), N, 1, M, N, K, 0, None, 0,
                                   p(ped), period, 16.0, stream()))
    torch.cuda.synchronize()
    assert relerr(out[:M], ref) < RTOL["bf16"]
    assert torch.isnan(out[M]).all()  # nothing written past M


def test_gemm_rejects_bad_k():
    a = torch.zeros(4, 100, device="cuda")
    rc = hip.lib().cn_op_gemm(0, p(a), 100, p(a), None, p(a), 4, 1, 4, 4, 100, 0, None, 0, None, 1, 1.0, stream())
    assert rc != 0 and b"multiple" in hip.lib().cn_last_error()


# ----------------------------------------------------------------------------------------------- convs
@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("B,T,Fd,Cc", [(2, 61, 80, 128), (1, 8, 6, 64), (3, 100, 83, 64), (2, 61, 80, 256), (1, 9, 7, 256),
                                       (3, 203, 80, 256)])
def test_conv1_conv2(prec, B, T, Fd, Cc):
    g = torch.Generator().manual_seed(B * T)
    x = torch.randn(B, T, Fd, generator=g)
    w1 = torch.randn(Cc, 1, 3, 3, generator=g) / 3
    b1 = torch.randn(Cc, generator=g) * 0.1
    w2 = torch.randn(Cc, Cc, 3, 3, generator=g) / math.sqrt(9 * Cc)
    b2 = torch.randn(Cc, generator=g) * 0.1
    ref1 = F.relu(F.conv2d(x.unsqueeze(1), w1, b1, stride=2, padding=1))  # (B,C,T1,F1)
    T1, F1 = ref1.shape[2], ref1.shape[3]
    L = hip.lib()
    out1 = torch.empty(B, T1, F1, Cc, dtype=tdtype(prec), device="cuda")
    w9c, xd, b1d, b2d = dev(w1.reshape(Cc, 9).t()), dev(x), dev(b1), dev(b2)
    hip.check(L.cn_op_conv1(hip.PRECISION[prec], p(xd), p(w9c), p(b1d), p(out1), B, T, Fd, Cc, stream()))
    torch.cuda.synchronize()
    assert relerr(out1.float().permute(0, 3, 1, 2), ref1) < (2e-6 if prec == "fp32" else 5e-3)
    # conv2 on the kernel's own (rounded) conv1 output so that only conv2's arithmetic is compared
    c1 = out1.float().cpu().permute(0, 3, 1, 2)
    ref2 = F.relu(F.conv2d(c1, rounded(w2, prec), b2, stride=2, padding=1))  # (B,C,T2,F2)
    T2, F2 = ref2.shape[2], ref2.shape[3]
    out2 = torch.empty(B, T2, F2, Cc, dtype=tdtype(prec), device="cuda")
    wk = dev(w2.permute(0, 2, 3, 1).reshape(Cc, 9 * Cc), tdtype(prec))
    hip.check(L.cn_op_conv2(hip.PRECISION[prec], p(out1), p(wk), p(b2d), p(out2), B, T1, F1, Cc, stream()))
    torch.cuda.synchronize()
    assert relerr(out2.float().permute(0, 3, 1, 2), ref2) < RTOL[prec]


@pytest.mark.parametrize("B,T,Fd", [(2, 67, 80), (3, 200, 80), (1, 9, 59), (5, 1000, 80), (1, 4, 83)])
def test_conv1_bordered_bf16_image_from_the_matrix_cores(B, T, Fd):
    """The bf16 engine's conv1 (embedding.py:102-104) as conv2's tile kernel reads it: [B][T1+2][F1+2][256] bf16 with a border of
    zeros.  Split-bf16 operands carry 16 significant bits in front of the bf16 rounding: the image equals bf16(relu(fp32
    convolution)) except where the fp32 value sits within 2^-16 of a rounding boundary (one code apart, a fraction of a percent)."""
    Cc = 256
    g = torch.Generator().manual_seed(B * T + Fd)
    x = torch.randn(B, T, Fd, generator=g) * 3
    w1 = torch.randn(Cc, 1, 3, 3, generator=g) / 3
    b1 = torch.randn(Cc, generator=g) * 0.1
    ref = F.relu(F.conv2d(x.unsqueeze(1), w1, b1, stride=2, padding=1)).permute(0, 2, 3, 1)  # (B,T1,F1,C)
    T1, F1 = ref.shape[1], ref.shape[2]
    want = torch.zeros(B, T1 + 2, F1 + 2, Cc)
    want[:, 1:-1, 1:-1] = ref
    want16 = want.to(torch.bfloat16)
    L = hip.lib()
    img = torch.full((B, T1 + 2, F1 + 2, Cc), 7.0, dtype=torch.bfloat16, device="cuda")
    w9c, xd, b1d = dev(w1.reshape(Cc, 9).t()), dev(x), dev(b1)
    hip.check(L.cn_op_conv1_bordered(p(xd), p(w9c), p(b1d), p(img), B, T, Fd, Cc, stream()))
    torch.cuda.synchronize()
    got = img.cpu()
    border = torch.ones(T1 + 2, F1 + 2, dtype=torch.bool)
    border[1:-1, 1:-1] = False
    assert (got[:, border] == 0).all()
    differ = got.view(torch.int16) != want16.view(torch.int16)
    assert differ.float().mean() < 5e-3
    # where they differ they are neighbours (one bf16 code apart), or a sum that cancels to almost nothing (the operands' 16 bits
    # are relative to the terms, not to the sum)
    assert bool(((got.float() - want16.float()).abs() <= want.abs() * 2.0 ** -7 + 2e-4).all())
    assert relerr(got.float(), want) < 5e-3


@pytest.mark.parametrize("B,T,Fd", [(2, 67, 80), (3, 200, 80), (1, 9, 16), (5, 1000, 80)])
def test_conv_frontend_fp8(B, T, Fd):
    """BASELINE config 5's conv front-end: conv1 + ReLU as an e4m3fn image at x8 (bordered, as conv2's LDS-DMA kernel reads it), conv2 +
    ReLU on e4m3 operands (v_mfma_scale_f32_32x32x64_f8f6f4, conv2.hip F8 form).  The image against torch's conv + e4m3 cast (a
    code may differ by one step where the two fp32 summation orders fall on different sides of a rounding boundary); the second
    convolution on the kernel's own image bytes and identically quantised weights."""
    Cc = 256
    g = torch.Generator().manual_seed(B * T)
    x = torch.randn(B, T, Fd, generator=g)
    w1 = torch.randn(Cc, 1, 3, 3, generator=g) / 3
    b1 = torch.randn(Cc, generator=g) * 0.1
    w2 = torch.randn(Cc, Cc, 3, 3, generator=g) / math.sqrt(9 * Cc)
    b2 = torch.randn(Cc, generator=g) * 0.1
    ref1 = F.relu(F.conv2d(x.unsqueeze(1), w1, b1, stride=2, padding=1))  # (B,C,T1,F1)
    T1, F1 = ref1.shape[2], ref1.shape[3]
    T2, F2 = (T1 - 1) // 2 + 1, (F1 - 1) // 2 + 1
    img = torch.empty(B, T1 + 2, F1 + 2, Cc, dtype=torch.uint8, device="cuda")
    out = torch.full((B, T2, F2, Cc), float("nan"), dtype=torch.bfloat16, device="cuda")
    w9c, xd, b1d, b2d = dev(w1.reshape(Cc, 9).t()), dev(x), dev(b1), dev(b2)
    w2k = w2.permute(0, 2, 3, 1).reshape(Cc, 9 * Cc).contiguous()
    ws = C.c_float(0)
    hip.check(hip.lib().cn_op_conv_frontend_fp8(p(xd), p(w9c), p(b1d), _hp(w2k), p(b2d), p(out), p(img), B, T, Fd, Cc, 8.0, 0.0, C.byref(ws),
                                                stream()))
    torch.cuda.synchronize()
    imgc = img.cpu()
    # border of zeros
    assert int(imgc[:, 0].max()) == 0 and int(imgc[:, -1].max()) == 0 and int(imgc[:, :, 0].max()) == 0 and int(imgc[:, :, -1].max()) == 0
    got8 = imgc[:, 1:-1, 1:-1]  # (B,T1,F1,C) bytes
    want8 = (ref1 * 8.0).clamp(max=448).to(torch.float8_e4m3fn).view(torch.uint8).permute(0, 2, 3, 1)
    diff = (got8.int() - want8.int()).abs()
    assert int(diff.max()) <= 1 and float((diff != 0).float().mean()) < 2e-3
    # second convolution on the kernel's own image
    c1 = got8.contiguous().view(torch.float8_e4m3fn).float().permute(0, 3, 1, 2) / 8.0
    wq = (w2 * ws.value).clamp(-448, 448).to(torch.float8_e4m3fn).float() / ws.value
    ref2 = F.relu(F.conv2d(c1, wq, b2, stride=2, padding=1))
    assert ws.value == 2.0 ** math.floor(math.log2(448.0 / w2.abs().max().item()))
    assert relerr(out.float().cpu().permute(0, 3, 1, 2), ref2) < 6e-3  # (bf16 output rounding: 2^-9 of the value)
    # the same with e4m3 output rows (x8): what linear_out's e4m3 form reads
    out8 = torch.zeros((B, T2, F2, Cc), dtype=torch.uint8, device="cuda")
    hip.check(hip.lib().cn_op_conv_frontend_fp8(p(xd), p(w9c), p(b1d), _hp(w2k), p(b2d), p(out8), None, B, T, Fd, Cc, 8.0, 8.0, C.byref(ws),
                                                stream()))
    torch.cuda.synchronize()
    want8 = (ref2 * 8.0).clamp(max=448).to(torch.float8_e4m3fn).view(torch.uint8).permute(0, 2, 3, 1)
    d8 = (out8.cpu().int() - want8.int()).abs()
    assert int(d8.max()) <= 1 and float((d8 != 0).float().mean()) < 5e-3  # (a code apart where the summation orders straddle a boundary)


@pytest.mark.parametrize("M", [8000, 257, 33])
def test_linear_out_fp8(M):
    """linear_out + sqrt(d) scale + positional rows on e4m3 operands (K = 5120; conv2_kernel<LINEAR, F8>) against an emulation on
    identically quantised operands."""
    K, N, Tp = 5120, 256, 250
    g = torch.Generator().manual_seed(M)
    a = torch.rand(M, K, generator=g) * 6  # (ReLU outputs)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).contiguous()
    bias, pe = 0.1 * torch.randn(N, generator=g), torch.randn(Tp, N, generator=g)
    a8 = (a * 8.0).clamp(max=448).to(torch.float8_e4m3fn)
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    ws = C.c_float(0)
    ad, bd, ped = a8.view(torch.uint8).cuda(), dev(bias), dev(pe)  # (named: a temporary would be freed before the call runs)
    hip.check(hip.lib().cn_op_linear256_fp8(p(ad), _hp(w), p(bd), p(out), M, K, 8.0, 16.0, p(ped), Tp, C.byref(ws), stream()))
    torch.cuda.synchronize()
    wq = (w * ws.value).clamp(-448, 448).to(torch.float8_e4m3fn).float() / ws.value
    ref = (F.linear(a8.float() / 8.0, wq) + bias) * 16.0 + pe[torch.arange(M) % Tp]
    assert relerr(out, ref) < 2e-5


# ----------------------------------------------------------------------------------------------- LayerNorm
@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("M,d", [(1, 128), (1001, 256), (37, 512), (5, 1024)])
def test_layernorm(prec, M, d):
    from oracle.cassnat_oracle import layer_norm

    g = torch.Generator().manual_seed(d)
    x = torch.randn(M, d, generator=g) * 3 + 0.5
    a2, b2 = torch.randn(d, generator=g), torch.randn(d, generator=g)
    y = torch.empty(M, d, dtype=tdtype(prec), device="cuda")
    xd, ad, bd = dev(x), dev(a2), dev(b2)
    hip.check(hip.lib().cn_op_layernorm(hip.PRECISION[prec], p(xd), p(ad), p(bd), p(y), M, d, 1e-6, stream()))
    torch.cuda.synchronize()
    assert relerr(y, layer_norm(x, a2, b2)) < (3e-6 if prec == "fp32" else 5e-3)


# ----------------------------------------------------------------------------------------------- attention
def attention_reference(q, k, v, mask):
    """q (B,Lq,H,64) etc; mask (B,Lq|1,Lk) bool.  src/models/modules/attention.py:13-24 semantics."""
    scores = torch.einsum("bqhd,bkhd->bhqk", q, k) / 8.0
    scores = scores.masked_fill(mask.unsqueeze(1) == 0, float(np.finfo(np.float32).min))
    pa = torch.softmax(scores, dim=-1)
    return torch.einsum("bhqk,bkhd->bqhd", pa, v)


def run_attention(prec, q, k, v, keymask=None, klen=None, intervals=None, causal=0):
    B, Lq, H, _ = q.shape
    Lk = k.shape[1]
    td = tdtype(prec)
    qd, kd, vd = dev(q.reshape(B * Lq, H * 64), td), dev(k.reshape(B * Lk, H * 64), td), dev(v.reshape(B * Lk, H * 64), td)
    o = torch.full((B * Lq, H * 64), float("nan"), dtype=td, device="cuda")
    km = None if keymask is None else dev(keymask.to(torch.uint8))
    kl = None if klen is None else dev(klen, torch.int32)
    iv = None if intervals is None else dev(intervals, torch.int32)
    hip.check(hip.lib().cn_op_attention(hip.PRECISION[prec], p(qd), H * 64, p(kd), H * 64, p(vd), H * 64, p(o), H * 64, B, H,
                                        Lq, Lk, p(km), p(kl), p(iv), 0 if iv is None else intervals.shape[1], causal,
                                        0.125, stream()))
    torch.cuda.synchronize()
    return o.float().cpu().reshape(B, Lq, H, 64)


ATT_TOL = {"fp32": 1e-5, "bf16": 2e-2}


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("Lq,Lk", [(250, 250), (50, 16), (171, 300), (128, 64), (129, 65)])
def test_attention_key_padding(prec, Lq, Lk):
    g = torch.Generator().manual_seed(Lq * 7 + Lk)
    B, H = 3, 2
    q, k, v = (torch.randn(B, L, H, 64, generator=g) for L in (Lq, Lk, Lk))
    lens = torch.tensor([Lk, max(1, Lk * 2 // 3), max(1, Lk // 5)])
    keymask = torch.arange(Lk)[None, :] < lens[:, None]
    keymask[1, Lk // 3] = False  # a hole: first feature exactly 0.0 inside an utterance
    ref = attention_reference(rounded(q, prec), rounded(k, prec), rounded(v, prec), keymask[:, None, :])
    got = run_attention(prec, q, k, v, keymask=keymask)
    assert relerr(got, ref) < ATT_TOL[prec]


@pytest.mark.parametrize("prec", PRECS)
def test_attention_trigger_intervals_and_empty_rows(prec):
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
        iv[b, 40, 2], iv[b, 40, 3] = 7, 8  # a forced single frame as the second interval
        dense[b, 40, 7] = True
        # rows 41.. stay empty -> uniform attention over all Lk keys (float-min fill, not -inf)
    dense &= keymask[:, None, :]
    ref = attention_reference(rounded(q, prec), rounded(k, prec), rounded(v, prec), dense)
    got = run_attention(prec, q, k, v, keymask=keymask, intervals=iv)
    assert relerr(got, ref) < ATT_TOL[prec]
    # an empty row really is the mean of V over every key, padded ones included
    assert relerr(got[0, 50], rounded(v, prec)[0].mean(0)) < ATT_TOL[prec]


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("causal", [0, 1])
def test_attention_target_length_mask(prec, causal):
    g = torch.Generator().manual_seed(5 + causal)
    B, H, U = 3, 4, 97
    q, k, v = (torch.randn(B, U, H, 64, generator=g) for _ in range(3))
    ylen = torch.tensor([97, 40, 1])
    mask = (torch.arange(U)[None, :] < ylen[:, None])[:, None, :].expand(B, U, U).clone()
    if causal:
        mask &= torch.tril(torch.ones(U, U, dtype=torch.bool))[None]
    ref = attention_reference(rounded(q, prec), rounded(k, prec), rounded(v, prec), mask)
    got = run_attention(prec, q, k, v, klen=ylen, causal=causal)
    assert relerr(got, ref) < ATT_TOL[prec]


def test_attention_large_scores_fp32():
    """forces the online-softmax rescale: the row maximum jumps by ~80 in a late key tile"""
    g = torch.Generator().manual_seed(1)
    B, H, Lq, Lk = 1, 1, 40, 200
    q, k, v = (torch.randn(B, L, H, 64, generator=g) for L in (Lq, Lk, Lk))
    k[0, 170, 0] = q[0, 3, 0] * 10.0
    ref = attention_reference(q, k, v, torch.ones(B, 1, Lk, dtype=torch.bool))
    got = run_attention("fp32", q, k, v)
    assert relerr(got, ref) < 1e-5


# ----------------------------------------------------------------------------------------------- generator tail
@pytest.mark.parametrize("M,V", [(37, 5000), (5, 40), (3, 1028)])
def test_logsoftmax_argmax(M, V):
    g = torch.Generator().manual_seed(V)
    x = torch.randn(M, V, generator=g)
    x[1, 7] = x[1].max() + 1.0
    x[1, 3] = x[1, 7]  # exact tie: the first index must win (torch.argmax on CPU)
    xd = dev(x)
    arg = torch.empty(M, dtype=torch.int32, device="cuda")
    mlp = torch.empty(M, dtype=torch.float32, device="cuda")
    hip.check(hip.lib().cn_op_logsoftmax_argmax(p(xd), M, V, p(arg), p(mlp), 1, stream()))
    torch.cuda.synchronize()
    ref = torch.log_softmax(x, -1)
    assert (xd.cpu() - ref).abs().max().item() < 2e-6
    assert arg.cpu().tolist() == ref.argmax(-1).tolist() and arg[1].item() == 3
    assert (mlp.cpu() - ref.max(-1).values).abs().max().item() < 2e-6


def test_topk():
    g = torch.Generator().manual_seed(8)
    x = torch.log_softmax(torch.randn(19, 5000, generator=g), -1)
    idx = torch.empty(19, 5, dtype=torch.int32, device="cuda")
    val = torch.empty(19, 5, dtype=torch.float32, device="cuda")
    xd = dev(x)
    hip.check(hip.lib().cn_op_topk(p(xd), 19, 5000, 5, p(idx), p(val), stream()))
    torch.cuda.synchronize()
    ref = torch.topk(x, 5, dim=-1)
    assert idx.cpu().tolist() == ref.indices.tolist()
    assert torch.equal(val.cpu(), ref.values)


@pytest.mark.parametrize("M,V,k,T", [(320, 5000, 15, 1.0), (7, 40, 10, 1.3), (33, 4234, 16, 0.7), (5, 1028, 1, 1.0)])
def test_logsoftmax_topk_fused_equals_the_two_kernel_form(M, V, k, T):
    """The one-pass tail of the autoregressive step must give bit-identical indices and values to log-softmax (row rewritten)
    followed by top-k, including exact ties (lower index first) and near-ties that collapse after the log-softmax rounding."""
    g = torch.Generator().manual_seed(M + V + k)
    x = torch.randn(M, V, generator=g) * 3
    x[0, 5] = x[0, 9] = x[0].max() + 0.5        # exact tie at the top
    x[1, 11] = x[1].max() + 2.0
    x[1, 3] = torch.nextafter(x[1, 11], torch.tensor(float("inf")))  # one ulp apart
    x[2] = torch.round(x[2] * 4) / 4             # many exact ties down the ranking
    xd = dev(x)
    idx_f = torch.empty(M, k, dtype=torch.int32, device="cuda")
    val_f = torch.empty(M, k, dtype=torch.float32, device="cuda")
    hip.check(hip.lib().cn_op_logsoftmax_topk(p(xd), M, V, T, k, p(idx_f), p(val_f), stream()))
    torch.cuda.synchronize()
    assert torch.equal(xd.cpu(), x)  # logits untouched
    y = (x / T).cuda() if T != 1.0 else x.clone().cuda()
    arg = torch.empty(M, dtype=torch.int32, device="cuda")
    mlp = torch.empty(M, dtype=torch.float32, device="cuda")
    hip.check(hip.lib().cn_op_logsoftmax_argmax(p(y), M, V, p(arg), p(mlp), 1, stream()))
    idx_2 = torch.empty(M, k, dtype=torch.int32, device="cuda")
    val_2 = torch.empty(M, k, dtype=torch.float32, device="cuda")
    hip.check(hip.lib().cn_op_topk(p(y), M, V, k, p(idx_2), p(val_2), stream()))
    torch.cuda.synchronize()
    assert torch.equal(idx_f.cpu(), idx_2.cpu())
    assert torch.equal(val_f.cpu(), val_2.cpu())
    ref = torch.log_softmax(x / T, -1)
    assert (val_f.cpu() - torch.topk(ref, k, dim=-1).values).abs().max().item() < 3e-6


# ----------------------------------------------------------------------------------------------- integer kernels
def run_align(best, mask, ratio, left=0, right=0):
    B, Tp = best.shape
    shift = torch.empty(B, Tp, dtype=torch.int32, device="cuda")
    src = torch.empty(B, dtype=torch.int32, device="cuda")
    ylen = torch.empty(B, dtype=torch.int32, device="cuda")
    ymax = torch.zeros(1, dtype=torch.int32, device="cuda")
    iv = torch.empty(B, Tp + 1, 4, dtype=torch.int32, device="cuda")
    bd, md, rd = dev(best, torch.int32), dev(mask.astype(np.uint8)), dev(ratio)
    hip.check(hip.lib().cn_op_ctc_align(p(bd), p(md), p(rd), B, Tp, 0, left, right, p(shift), p(src), p(ylen), p(ymax), p(iv),
                                        stream()))
    torch.cuda.synchronize()
    return shift.cpu().numpy(), src.cpu().numpy(), ylen.cpu().numpy(), int(ymax.item()), iv.cpu().numpy()


def dense_from_intervals(iv, mask, ymax):
    B, Tp = mask.shape
    t = np.arange(Tp)[None, None, :]
    s1, e1, s2, e2 = (iv[:, :ymax, i][:, :, None] for i in range(4))
    return (((t >= s1) & (t < e1)) | ((t >= s2) & (t < e2))) & mask[:, None, :]


def test_ctc_align_known_answer(golden):
    g = golden("align_kat")
    shift, src, ylen, ymax, iv = run_align(g["path"], g["mask"], np.array([1.0, 8 / 12], np.float32))
    np.testing.assert_array_equal(shift, g["aligned_seq_shift"])
    np.testing.assert_array_equal(src, g["src_size"])
    np.testing.assert_array_equal(ylen, g["ylen"])
    assert ymax == int(g["ymax"])
    np.testing.assert_array_equal(dense_from_intervals(iv, g["mask"], ymax), g["trigger"])


@pytest.mark.parametrize("left,right", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("Tp", [12, 250, 700])
def test_ctc_align_random_vs_oracle(Tp, left, right):
    from oracle import cassnat_oracle as orc

    rng = np.random.default_rng(Tp + 10 * left + right)
    B = 9
    best = rng.integers(0, 6, size=(B, Tp)) * (rng.random((B, Tp)) < 0.5)
    lens = rng.integers(1, Tp + 1, size=B)
    lens[0] = Tp
    mask = np.arange(Tp)[None, :] < lens[:, None]
    mask[2, rng.integers(0, lens[2], size=3)] = False  # holes
    best[3] = 0  # an utterance with no tokens at all
    best[4, lens[4] - 1] = 5  # a token on the very last valid frame (dropped by the shift)
    ratio = (lens / Tp).astype(np.float32)
    ratio[5] = np.float32((lens[5] - 0.6) / Tp)  # src_size one short of the mask length
    shift, src, ylen, ymax, iv = run_align(best, mask, ratio, left, right)
    o_shift, o_ylen0, o_ymax0 = orc.best_path_align(best, mask)
    o_src = orc.src_size_frames(ratio, Tp)
    o_trig, o_ylen, o_ymax = orc.align_to_intervals(o_shift, o_ylen0, o_ymax0, mask, o_src, 0, left, right)
    np.testing.assert_array_equal(shift, o_shift)
    np.testing.assert_array_equal(src, o_src)
    np.testing.assert_array_equal(ylen, o_ylen)
    assert ymax == o_ymax
    np.testing.assert_array_equal(dense_from_intervals(iv, mask, ymax), o_trig)


def test_greedy_pack_matches_oracle():
    from oracle import cassnat_oracle as orc

    rng = np.random.default_rng(4)
    B, U, V = 6, 23, 50
    att = torch.log_softmax(torch.from_numpy(rng.standard_normal((B, U, V)).astype(np.float32)), -1)
    ylen = np.array([23, 22, 10, 1, 5, 23])
    best = att.max(-1)
    hyp = torch.full((B, U + 2), -1, dtype=torch.int32, device="cuda")
    hl = torch.empty(B, dtype=torch.int32, device="cuda")
    sc = torch.empty(B, dtype=torch.float64, device="cuda")
    td, vd, yd = dev(best.indices, torch.int32), dev(best.values), dev(ylen, torch.int32)
    hip.check(hip.lib().cn_op_greedy_pack(p(td), p(vd), p(yd), B, U, 1, U + 2, p(hyp), p(hl), p(sc), stream()))
    torch.cuda.synchronize()
    o_hyp, o_sc = orc.greedy_finish(att, ylen, U)
    for b in range(B):
        assert hyp[b, : hl[b]].cpu().tolist() == o_hyp[b]
        assert sc[b].item() == o_sc[b]  # sequential double sum: bit-identical


@pytest.mark.parametrize("B,U,ld,V", [(37, 9, 12, 5000), (3, 1, 1, 40), (1600, 5, 7, 1028), (2, 16, 16, 4234)])
def test_generator_target_gather_fused_bf16(B, U, ld, V):
    """TransformerLM scoring tail: log-probability of a given target per row, from the fused generator kernel."""
    g = torch.Generator().manual_seed(B + U + V)
    h = torch.randn(B * U, 256, generator=g)
    w = (torch.randn(V, 256, generator=g) / 16).contiguous()
    b = (0.1 * torch.randn(V, generator=g)).contiguous()
    tgt = torch.randint(0, V, (B, ld), generator=g, dtype=torch.int32)
    tgt[0, 0] = V - 1
    tgt[-1, U - 1] = 0
    ref = torch.log_softmax(F.linear(rounded(h, "bf16"), rounded(w, "bf16"), b), -1).view(B, U, V)
    want = torch.gather(ref, 2, tgt[:, :U].long().unsqueeze(-1)).squeeze(-1)
    hd, td = dev(h, torch.bfloat16), dev(tgt)
    out = torch.full((B, ld), float("nan"), dtype=torch.float32, device="cuda")
    hip.check(hip.lib().cn_op_genmax_gather(p(hd), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), B, U, V, p(td), ld, p(out), stream()))
    torch.cuda.synchronize()
    assert (out.cpu()[:, :U] - want).abs().max().item() < 3e-4
    if ld > U:
        assert torch.isnan(out.cpu()[:, U:]).all()  # nothing written outside the U scored positions


def _e4m3(t):
    """round to OCP e4m3fn, saturating (torch's CPU cast is the emulation, as in test_gemm_fp8_...)"""
    return t.clamp(-448, 448).to(torch.float8_e4m3fn).float()


@pytest.mark.parametrize("M,dff,tail_n,with_ctx,x_mode", [(8000, 2048, 768, True, 32), (8000, 2048, 768, True, 32 | 19), (257, 256, 0, False, 32),
                                                           (33, 512, 256, True, 32 | 16), (4100, 1024, 1536, True, 32 | 3)])
def test_chain_fp8_feed_forward(M, dff, tail_n, with_ctx, x_mode):
    """x_mode bit 32 (BASELINE config 5): the two feed-forward products of the chain on e4m3 operands - LayerNorm output at x16,
    ReLU output at x8 (both saturating), weights at the largest power-of-two scale that keeps them in range - against an emulation
    with the same roundings; everything else of the launch as in test_chain_bf16."""
    from oracle.cassnat_oracle import layer_norm

    g = torch.Generator().manual_seed(M + dff + tail_n)
    d = 256
    rn = lambda *s: torch.randn(*s, generator=g)
    x = rn(M, d) * 2 + 0.3
    ctx = rn(M, d)
    wo, bo = (rn(d, d) / 16).contiguous(), 0.1 * rn(d)
    a1, b1n = 1 + 0.1 * rn(d), 0.1 * rn(d)
    w1, b1 = (rn(dff, d) / 16).contiguous(), 0.1 * rn(dff)
    w2, b2 = (rn(d, dff) / math.sqrt(dff)).contiguous(), 0.1 * rn(d)
    na, nb = 1 + 0.1 * rn(d), 0.1 * rn(d)
    wt, bt = (rn(max(tail_n, 1), d) / 16).contiguous(), 0.1 * rn(max(tail_n, 1))
    ref = x.clone()
    if with_ctx:
        ref = ref + F.linear(rounded(ctx, "bf16"), rounded(wo, "bf16"), bo)
    pow2 = lambda w: 2.0 ** math.floor(math.log2(448.0 / w.abs().max().item()))
    s1, s2 = pow2(w1), pow2(w2)
    xn8 = _e4m3(layer_norm(ref, a1, b1n) * 16.0)
    pre = F.linear(xn8, _e4m3(w1 * s1)) / (s1 * 16.0) + b1
    h8 = _e4m3(F.relu(pre) * 8.0)
    ref = ref + F.linear(h8, _e4m3(w2 * s2)) / (s2 * 8.0) + b2
    xd = dev(_to_blocked(x) if x_mode & 1 else x)
    if x_mode & 2 and not x_mode & 1:
        xd = torch.cat([xd, torch.zeros(32, d, device="cuda")])
    ctxd = dev(ctx, torch.bfloat16) if with_ctx else None
    with_next = True
    ldo = tail_n if tail_n else d
    rows_out = (M + 31) // 32 * 32 if x_mode & 16 else M
    out = torch.full((rows_out, ldo), float("nan"), dtype=torch.bfloat16, device="cuda")
    hip.check(hip.lib().cn_op_chain(p(xd), p(ctxd) if with_ctx else None, d, _hp(wo), _hp(bo), _hp(a1), _hp(b1n), _hp(w1),
                                    _hp(b1), _hp(w2), _hp(b2), _hp(na), _hp(nb), _hp(wt), _hp(bt), p(out), ldo, M, dff, tail_n, 1e-6,
                                    x_mode, stream()))
    torch.cuda.synchronize()
    nrb = (M + 31) // 32
    xo = xd.cpu().reshape(-1)[: nrb * 32 * d]
    xo = _from_blocked(xo.view(nrb, 32, 64, 4), M) if x_mode & 2 else xo.view(-1, d)[:M]
    # identical operand bytes except where a LayerNorm output or a pre-activation sits on a rounding boundary of e4m3 and the two
    # summation orders fall on different sides: one step there is 1/16 of the value (bf16: 1/256), in one of 256 / d_ff terms.
    # Measured 3.2e-3 of the output range at worst over these cases (most: < 1e-3)
    assert relerr(xo, ref) < 8e-3
    out = _from_blocked16(out.cpu(), M, ldo) if (x_mode & 16 and tail_n) else out[:M]
    y = layer_norm(xo, na, nb)
    if tail_n:
        assert relerr(out, F.linear(rounded(y, "bf16"), rounded(wt, "bf16"), bt)) < 6e-3
    else:
        assert relerr(out, y) < 5e-3


# ---------------------------------------------------------------------------- split-bf16 projections (proj_x3.hip)
def _unsplit(raw, M, N):
    """split-bf16 rows (per 32 columns: 64 bytes of hi halves, 64 bytes of lo halves) -> float64 hi + lo"""
    v = raw.cpu().view(torch.bfloat16).view(M, N // 32, 2, 32).double()
    return (v[:, :, 0] + v[:, :, 1]).reshape(M, N)


@pytest.mark.parametrize("M,N", [(8000, 256), (7969, 768), (33, 512), (1, 32), (32800, 256), (33000, 768), (70, 96)])
def test_projection_bf16x3(M, N):
    """The K = 256 projections of the split-bf16 engine against float64: fp32 output with and without the residual, and the
    split-bf16 output (whose hi + lo carries 16 mantissa bits)."""
    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, 256, generator=g).contiguous()
    w = (torch.randn(N, 256, generator=g) / 16).contiguous()
    b = (0.1 * torch.randn(N, generator=g)).contiguous()
    x = torch.randn(M, N, generator=g)
    ref = F.linear(a.double(), w.double(), b.double())
    hp = lambda t: C.c_void_p(t.data_ptr())
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device="cuda")
    hip.check(hip.lib().cn_op_proj_x3(hp(a), hp(w), hp(b), None, 1.0, p(out), M, N, 0, stream()))
    torch.cuda.synchronize()
    assert (out.cpu().double() - ref).abs().max().item() < 5e-5
    xd = dev(x)
    hip.check(hip.lib().cn_op_proj_x3(hp(a), hp(w), hp(b), p(xd), 0.5, p(xd), M, N, 0, stream()))
    torch.cuda.synchronize()
    assert (xd.cpu().double() - (x.double() + 0.5 * ref)).abs().max().item() < 5e-5
    raw = torch.zeros(M, N * 2, dtype=torch.int16, device="cuda")
    hip.check(hip.lib().cn_op_proj_x3(hp(a), hp(w), hp(b), None, 1.0, p(raw), M, N, 1, stream()))
    torch.cuda.synchronize()
    got = _unsplit(raw, M, N)
    assert ((got - ref).abs() / (1 + ref.abs())).max().item() < 5e-5


# ----------------------------------------------------------------------------------------------- fused FFN sublayer
@pytest.mark.parametrize("M,dff,with_next,nslice", [(8000, 2048, True, 1), (45, 256, False, 1), (2304, 2048, True, 1), (32, 128, True, 1),
                                                    (320, 2048, True, 8), (7, 2048, False, 16), (130, 1024, True, 4), (64, 256, True, 2)])
def test_ffn_fused_bf16(M, dff, with_next, nslice):
    from oracle.cassnat_oracle import layer_norm

    g = torch.Generator().manual_seed(M + dff)
    d = 256
    x = torch.randn(M, d, generator=g) * 2 + 0.3
    a, b = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    na, nb = 1 + 0.1 * torch.randn(d, generator=g), 0.1 * torch.randn(d, generator=g)
    w1 = (torch.randn(dff, d, generator=g) / math.sqrt(d)).contiguous()
    b1 = 0.1 * torch.randn(dff, generator=g)
    w2 = (torch.randn(d, dff, generator=g) / math.sqrt(dff)).contiguous()
    b2 = 0.1 * torch.randn(d, generator=g)
    # reference with the same roundings the kernel applies: LN output, weights and hidden activations in bf16
    xn = rounded(layer_norm(x, a, b), "bf16")
    h = rounded(F.relu(F.linear(xn, rounded(w1, "bf16"), b1)), "bf16")
    ref = x + F.linear(h, rounded(w2, "bf16"), b2)
    xd, ad, bd, b1d, b2d, nad, nbd = (dev(t) for t in (x, a, b, b1, b2, na, nb))
    xn_out = torch.full((M, d), float("nan"), dtype=torch.bfloat16, device="cuda") if with_next else None
    hip.check(hip.lib().cn_op_ffn_fused(p(xd), p(ad), p(bd), C.c_void_p(w1.data_ptr()), p(b1d), C.c_void_p(w2.data_ptr()),
                                        p(b2d), p(nad) if with_next else None, p(nbd) if with_next else None,
                                        p(xn_out), M, dff, 1e-6, nslice, stream()))
    torch.cuda.synchronize()
    assert relerr(xd, ref) < 2e-3  # fp32 residual stream; error = accumulation order + rare bf16 double-rounding of h
    if with_next:
        assert relerr(xn_out, layer_norm(xd.cpu(), na, nb)) < 5e-3


# ------------------------------------------------------------------------------------------------ row-chain kernel
def _hp(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _to_blocked(x):
    """[M][256] -> the chain kernel's blocked layout (include/cassnat_hip.h, cn_op_chain), rows padded to 32."""
    M = x.shape[0]
    nrb = (M + 31) // 32
    xp = torch.zeros(nrb * 32, 256, dtype=x.dtype)
    xp[:M] = x
    # row = 32 rb + r, channel = 32 nt + 8 g + 4 h + e  ->  [rb][i = 4 nt + g][lane = r + 32 h][e]
    return xp.view(nrb, 32, 8, 4, 2, 4).permute(0, 2, 3, 4, 1, 5).reshape(nrb, 32, 64, 4).contiguous()


def _from_blocked(xb, M):
    nrb = xb.shape[0]
    return xb.view(nrb, 8, 4, 2, 32, 4).permute(0, 4, 1, 2, 3, 5).reshape(nrb * 32, 256)[:M].contiguous()


def _from_blocked16(buf, M, N):
    """The chain kernel's blocked bf16 matrix (include/cassnat_hip.h, cn_op_chain, x_mode bit 16) -> row-major [M][N]."""
    nrb = (M + 31) // 32
    t = buf.view(nrb, N // 32, 2, 2, 32, 8)  # [row block][column tile][16-column half][bit 3 of the column][row][8]
    return t.permute(0, 4, 1, 2, 3, 5).reshape(nrb * 32, N)[:M]


@pytest.mark.parametrize("x_mode", [0, 3, 1, 2, 8, 11, 16, 19])  # (bit 8: Swish feed-forward; bit 16: blocked tail output)
@pytest.mark.parametrize("M,dff,tail_n,with_ctx,with_next", [
    (8000, 2048, 768, True, True),    # encoder layer at config 2: out-proj + FFN + next layer's QKV
    (8000, 2048, 0, True, True),      # last encoder layer: the stack's final LayerNorm is the output
    (300, 2048, 256, True, True),     # ragged M (3 workgroups, last one 44 rows), Q-only tail
    (129, 256, 0, True, False),       # no next norm at all; short FFN (8 tiles)
    (128, 0, 768, False, True),       # LayerNorm + QKV only (layer-0 entry)
    (77, 2048, 768, False, True),     # FFN + tail without an output projection
    (64, 128, 256, True, True),       # short stream: 3 groups
    (300, 2048, 1536, True, True),    # the last encoder layer's tail: K|V of three decoder-side layers (6 groups)
])
def test_chain_bf16(M, dff, tail_n, with_ctx, with_next, x_mode):
    from oracle.cassnat_oracle import layer_norm

    g = torch.Generator().manual_seed(M + dff + tail_n)
    d = 256
    rn = lambda *s: torch.randn(*s, generator=g)
    x = rn(M, d) * 2 + 0.3
    ctx = rn(M, d)
    wo, bo = (rn(d, d) / 16).contiguous(), 0.1 * rn(d)
    a1, b1n = 1 + 0.1 * rn(d), 0.1 * rn(d)
    w1, b1 = (rn(max(dff, 1), d) / 16).contiguous(), 0.1 * rn(max(dff, 1))
    w2, b2 = (rn(d, max(dff, 1)) / math.sqrt(max(dff, 1))).contiguous(), 0.1 * rn(d)
    na, nb = 1 + 0.1 * rn(d), 0.1 * rn(d)
    wt, bt = (rn(max(tail_n, 1), d) / 16).contiguous(), 0.1 * rn(max(tail_n, 1))
    # reference with the kernel's roundings: ctx, LN outputs, weights and hidden activations in bf16, fp32 accumulation
    ref = x.clone()
    if with_ctx:
        ref = ref + F.linear(rounded(ctx, "bf16"), rounded(wo, "bf16"), bo)
    if dff:
        xn = rounded(layer_norm(ref, a1, b1n), "bf16")
        act = F.silu if x_mode & 8 else F.relu
        h = rounded(act(F.linear(xn, rounded(w1, "bf16"), b1)), "bf16")
        ref = ref + F.linear(h, rounded(w2, "bf16"), b2)
    xd = dev(_to_blocked(x) if x_mode & 1 else x)
    if x_mode & 2 and not x_mode & 1:  # row-major in, blocked out: the buffer must hold whole 32-row blocks
        xd = torch.cat([xd, torch.zeros(32, d, device="cuda")])
    ctxd = dev(ctx, torch.bfloat16) if with_ctx else None
    ldo = tail_n if tail_n else d
    rows_out = (M + 31) // 32 * 32 if x_mode & 16 else M
    out = torch.full((rows_out, ldo), float("nan"), dtype=torch.bfloat16, device="cuda") if with_next else None
    hip.check(hip.lib().cn_op_chain(p(xd), p(ctxd) if with_ctx else None, d, _hp(wo), _hp(bo), _hp(a1), _hp(b1n), _hp(w1),
                                    _hp(b1), _hp(w2), _hp(b2), _hp(na) if with_next else None, _hp(nb) if with_next else None,
                                    _hp(wt), _hp(bt), p(out) if with_next else None, ldo, M, dff, tail_n, 1e-6, x_mode, stream()))
    torch.cuda.synchronize()
    nrb = (M + 31) // 32
    xo = xd.cpu().reshape(-1)[: nrb * 32 * d]
    xo = _from_blocked(xo.view(nrb, 32, 64, 4), M) if x_mode & 2 else xo.view(-1, d)[:M]
    if not with_ctx and not dff:
        xo = x  # nothing to store: the kernel leaves x alone (in whatever layout it came)
    assert relerr(xo, ref) < 2e-3
    if with_next and x_mode & 16 and tail_n:
        out = _from_blocked16(out.cpu(), M, ldo)
    elif with_next:
        out = out[:M]
    if with_next:
        y = layer_norm(xo, na, nb)
        if tail_n:
            assert relerr(out, F.linear(rounded(y, "bf16"), rounded(wt, "bf16"), bt)) < 6e-3
        else:
            assert relerr(out, y) < 5e-3


# ----------------------------------------------------------------------------------------------- fused generator tail
@pytest.mark.parametrize("M,V", [(8000, 5000), (37, 5000), (2336, 1028), (32, 40), (100, 4234)])
def test_generator_argmax_fused_bf16(M, V):
    g = torch.Generator().manual_seed(M + V)
    h = torch.randn(M, 256, generator=g)
    w = (torch.randn(V, 256, generator=g) / 16).contiguous()
    b = (0.1 * torch.randn(V, generator=g)).contiguous()
    logits = F.linear(rounded(h, "bf16"), rounded(w, "bf16"), b)
    ref = torch.log_softmax(logits, -1)
    hd = dev(h, torch.bfloat16)
    arg = torch.full((M,), -1, dtype=torch.int32, device="cuda")
    mlp = torch.full((M,), float("nan"), dtype=torch.float32, device="cuda")
    hip.check(hip.lib().cn_op_genmax(p(hd), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), M, V, p(arg), p(mlp), stream()))
    torch.cuda.synchronize()
    top2 = torch.topk(ref, 2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4  # rows with a clear winner must agree exactly (fp32 accumulation order differs)
    assert (arg.cpu()[clear] == ref.argmax(-1)[clear].int()).all()
    assert (arg.cpu() >= 0).all() and (arg.cpu() < V).all()
    assert (mlp.cpu() - ref.max(-1).values).abs().max().item() < 2e-4


@pytest.mark.parametrize("M,V", [(8000, 5000), (8300, 5000), (37, 5000), (2336, 1028), (32, 40), (100, 4234), (1, 6144)])
def test_generator_argmax_fused_bf16x3(M, V):
    """The split-bf16 generator tail (hi + lo operands, three MFMAs per product) against the fp32 generator in float64:
    the engine that claims the reference's tolerance must not lose it in its last kernel."""
    g = torch.Generator().manual_seed(M + V)
    h = (torch.randn(M, 256, generator=g) * 1.5).contiguous()
    w = (torch.randn(V, 256, generator=g) / 16).contiguous()
    b = (0.1 * torch.randn(V, generator=g)).contiguous()
    ref = torch.log_softmax(F.linear(h.double(), w.double(), b.double()), -1)
    arg = torch.full((M,), -1, dtype=torch.int32, device="cuda")
    mlp = torch.full((M,), float("nan"), dtype=torch.float32, device="cuda")
    hip.check(hip.lib().cn_op_genmax_x3(C.c_void_p(h.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), M, V,
                                        p(arg), p(mlp), None, 0, 0, None, stream()))
    torch.cuda.synchronize()
    top2 = torch.topk(ref, 2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4
    assert clear.float().mean() > 0.99
    assert (arg.cpu()[clear] == ref.argmax(-1)[clear].int()).all()
    assert (arg.cpu() >= 0).all() and (arg.cpu() < V).all()
    # hi + lo carries 16 mantissa bits per operand (2^-17 relative), 256 products of magnitude ~0.1 per logit: ~1e-5 expected,
    # against the gate's 1e-3 (the bf16 kernel: 5e-3)
    assert (mlp.cpu().double() - ref.max(-1).values).abs().max().item() < 1e-4
    # arg-max only (what the CTC alignment asks for)
    arg2 = torch.full((M,), -1, dtype=torch.int32, device="cuda")
    hip.check(hip.lib().cn_op_genmax_x3(C.c_void_p(h.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), M, V,
                                        p(arg2), None, None, 0, 0, None, stream()))
    torch.cuda.synchronize()
    assert torch.equal(arg2.cpu(), arg.cpu())


@pytest.mark.parametrize("B,U,ld,V", [(37, 9, 12, 5000), (3, 1, 1, 40), (1700, 5, 7, 1028), (2, 16, 16, 4234)])
def test_generator_target_gather_fused_bf16x3(B, U, ld, V):
    g = torch.Generator().manual_seed(B + U + V)
    h = torch.randn(B * U, 256, generator=g).contiguous()
    w = (torch.randn(V, 256, generator=g) / 16).contiguous()
    b = (0.1 * torch.randn(V, generator=g)).contiguous()
    tgt = torch.randint(0, V, (B, ld), generator=g, dtype=torch.int32)
    tgt[0, 0] = V - 1
    tgt[-1, U - 1] = 0
    ref = torch.log_softmax(F.linear(h.double(), w.double(), b.double()), -1).view(B, U, V)
    want = torch.gather(ref, 2, tgt[:, :U].long().unsqueeze(-1)).squeeze(-1)
    td = dev(tgt)
    out = torch.full((B, ld), float("nan"), dtype=torch.float32, device="cuda")
    hip.check(hip.lib().cn_op_genmax_x3(C.c_void_p(h.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), B * U, V,
                                        None, None, p(td), U, ld, p(out), stream()))
    torch.cuda.synchronize()
    assert (out.cpu()[:, :U].double() - want).abs().max().item() < 1e-4
    if ld > U:
        assert torch.isnan(out.cpu()[:, U:]).all()


@pytest.mark.parametrize("M,V", [(700, 5000), (40, 1028)])
def test_generator_argmax_ties_go_to_the_lower_index(M, V):
    """genmax.hip walks its vocabulary tiles in a rotation that differs per workgroup; among EXACTLY equal logits the lower
    index wins whatever the order (torch.argmax: first maximum).  Every vocabulary row appears twice, far apart, with the same
    bias: all products are bit-identical pairs (same operands, same accumulation order)."""
    g = torch.Generator().manual_seed(V)
    h = torch.randn(M, 256, generator=g)
    half = V // 2
    w0 = torch.randn(half, 256, generator=g) / 16
    b0 = 0.1 * torch.randn(half, generator=g)
    w = torch.cat([w0, w0, torch.zeros(V - 2 * half, 256)]).contiguous()
    b = torch.cat([b0, b0, torch.full((V - 2 * half,), -30.0)]).contiguous()
    ref = F.linear(rounded(h, "bf16"), rounded(w0, "bf16"), b0)
    hd = dev(h, torch.bfloat16)
    arg = torch.full((M,), -1, dtype=torch.int32, device="cuda")
    mlp = torch.full((M,), float("nan"), dtype=torch.float32, device="cuda")
    hip.check(hip.lib().cn_op_genmax(p(hd), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), M, V, p(arg), p(mlp), stream()))
    torch.cuda.synchronize()
    a = arg.cpu()
    assert (a >= 0).all() and (a < half).all(), "a tie between v and v + V/2 must resolve to v"
    top2 = torch.topk(ref, 2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4
    assert (a[clear] == ref.argmax(-1)[clear].int()).all()


@pytest.mark.parametrize("F,with_cmvn", [(80, True), (80, False), (7, True), (83, False)])
def test_unpack_rows_is_collate_and_cmvn_on_the_device(F, with_cmvn):
    """cn_op_unpack_rows: the packed archive rows of a pass -> the padded batch, bit for bit what SuperviseLoader.collate_fn
    (src/data/speech_loader.py:327-356) builds from the dataset's float64-normalised matrices (:109-115, 147-149)."""
    from cassnat_asr_public_amd.pipeline import PackedBatch

    rng = np.random.default_rng(F)
    lens = [61, 1, 33, 64, 32, 47, 96]
    T = 96
    views = [(rng.standard_normal((n, F)) * 3 + 0.5).astype(np.float32) for n in lens]
    mean, std = rng.standard_normal(F), rng.random(F) + 0.5
    want = PackedBatch(views).padded(0.0, (mean, std) if with_cmvn else None).numpy()
    packed = torch.from_numpy(np.concatenate(views, 0)).cuda()
    off = torch.tensor(np.concatenate([[0], np.cumsum(lens)[:-1]]), dtype=torch.int32, device="cuda")
    ln = torch.tensor(lens, dtype=torch.int32, device="cuda")
    out = torch.full((len(lens), T, F), 7.0, device="cuda")
    hip.unpack_rows(packed, off, ln, out, 0.0, torch.from_numpy(mean).cuda() if with_cmvn else None, torch.from_numpy(std).cuda() if with_cmvn else None)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), want)
