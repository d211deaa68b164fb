"""End-to-end GPU parity of the CASS-NAT hot path through the C ABI and through the drop-in Python API.

fp32 engine (exact-f32 MFMA) is the parity gate named by BASELINE.json's north_star:
  * token-for-token on the CTC alignment / indexing (frames whose reference top-2 margin is < 1e-4 may flip
    and are reported; every other frame must agree),
  * |logit error| <= 1e-3 on the encoder log-posteriors (the tolerance north_star states).
The bf16 engine is the throughput mode: it reports its agreement rate and logit error against the same goldens
and is gated only loosely (with random weights bf16 cannot meet the fp32 gates).
"""
import numpy as np
import pytest
import torch

from conftest import config1_case, config2_b8_case, config2_b32_case, config5_shape_case, load_golden, tiny_case
from cassnat_asr_public_amd import hip, synth
from cassnat_asr_public_amd.models.cassnat import make_model
from cassnat_asr_public_amd.utils.agreement import assert_flips_explained, flips_by_margin

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-3  # north_star: "within 1e-3 on encoder logits"
# A frame's arg-max can only change when the error on its two best log-posteriors exceeds the reference's margin between them:
# margin < 2 max|d log-posterior| over ALL logits.  The fixtures hold a strided sample of the logits (every st-th frame, every
# sv-th label), whose maximum error is a little below the full maximum; FLIP_K x the sampled error is the gate (the tests print
# the measured ratio beside the flip counts).
FLIP_K = 1.5  # (measured, GPUTEST r04: largest flip margin = 0.22 .. 0.89 x the sampled error, bf16 and every fp8 scope)


class Vocab:
    word2index = {"blank": 0, "sos": 1, "eos": 2, "unk": 3}


def build(args, state, precision, capture=False):
    args.hip_precision = precision
    args.hip_capture = capture
    model = make_model(args.input_size, args).cuda()
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(torch.from_numpy(state[k]))
    return model


def decode(model, args, feats, sizes):
    src = torch.from_numpy(feats)
    mask = (src[:, :, 0] != args.padding_idx).unsqueeze(1)
    with torch.no_grad():
        out, _ = model.beam_decode(src.cuda(), mask.cuda(), torch.from_numpy(sizes).cuda(), Vocab, args)
    return out


def maxerr(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


def test_parameter_names_match_reference_checkpoint_keys():
    args = synth.make_args("tiny")
    model = make_model(80, args)
    assert [k for k, _ in model.named_parameters()] == list(synth.param_shapes(args).keys())
    assert {k: tuple(v.shape) for k, v in model.named_parameters()} == dict(synth.param_shapes(args))


def test_tiny_fp32_every_stage():
    g = load_golden("tiny_stages")
    args, state, feats, sizes = tiny_case()
    model = build(args, state, "fp32", capture=True)
    out = decode(model, args, feats, sizes)
    eng = model._engine
    # channels-last on the device: (B,T1,F1,C) / (B,T',F2,C)  vs reference (B,C,T,F)
    assert maxerr(eng.fetch("conv1").transpose(0, 3, 1, 2)[:, ::8], g["conv1_c8"]) < 1e-5
    assert maxerr(eng.fetch("conv2").transpose(0, 3, 1, 2), g["conv2"]) < 1e-4
    for name, tol in [("x_embed", 2e-4), ("enc_layer0", 5e-4), ("enc_layer1", 5e-4), ("enc_h", 1e-4), ("ctc_out", LOGIT_TOL),
                      ("ac_embed", 5e-4), ("pred_embed", 5e-4), ("dec_h", 1e-4), ("att_out", LOGIT_TOL)]:
        assert maxerr(eng.fetch(name), g[name]) < tol, name
    np.testing.assert_array_equal(eng.fetch("aligned_seq_shift"), g["aligned_seq_shift"])
    np.testing.assert_array_equal(eng.fetch("ylen"), g["ylen"])
    assert int(eng.fetch("ymax")[0]) == int(g["ymax"])
    for b, seqs in enumerate(out):
        assert seqs[0]["hyp"] == g["hyp"][b, : g["hyp_len"][b]].tolist()
        assert abs(seqs[0]["score"] - g["score"][b]) < 1e-3
    assert maxerr(eng.fetch("ctc_out"), g["ctc_out"]) < 2e-5  # what the fp32 path actually achieves here


@pytest.mark.parametrize("name,ov", [
    ("dilate", dict(left_trigger=1, right_trigger=1)),
    ("srctrig", dict(src_trigger=True)),
    ("unimask", dict(use_unimask=True)),
    ("beam3", dict(beam_width=3, length_penalty=0.1)),
])
def test_tiny_fp32_option_variants(name, ov):
    g = load_golden("tiny_" + name)
    args, state, feats, sizes = tiny_case(**ov)
    model = build(args, state, "fp32", capture=True)
    out = decode(model, args, feats, sizes)
    eng = model._engine
    np.testing.assert_array_equal(eng.fetch("ylen"), g["ylen"])
    assert maxerr(eng.fetch("dec_h"), g["dec_h"]) < 1e-4
    assert maxerr(eng.fetch("att_out"), g["att_out"]) < LOGIT_TOL
    for b, seqs in enumerate(out):
        assert seqs[0]["hyp"] == g["hyp"][b, : g["hyp_len"][b]].tolist()
        assert abs(seqs[0]["score"] - g["score"][b]) < 1e-3
        if "beam_hyp" in g:
            for j, s in enumerate(seqs):
                assert s["hyp"] == g["beam_hyp"][b, j, : g["beam_len"][b, j]].tolist()
                assert abs(s["score"] - g["beam_score"][b, j]) < 1e-3


def check_against_golden(model, args, feats, sizes, g, st, sv, dt, fp32):
    out = decode(model, args, feats, sizes)
    eng = model._engine
    best = eng.fetch("best_paths")
    margin = g["margin"].astype(np.float32)
    flips = best != g["best_paths"]
    ctc_err = maxerr(eng.fetch("ctc_out")[:, ::st, ::sv], g["ctc_sample"])
    report = dict(frames=int(flips.size), flips=int(flips.sum()), ctc_logit_err=ctc_err)
    report["margins"] = flips_by_margin(best, g["best_paths"], margin)
    if fp32:
        assert (margin[flips] < 1e-4).all(), f"argmax differs on a clear-margin frame: {report}"
        assert ctc_err < LOGIT_TOL, report
        if not flips.any():
            np.testing.assert_array_equal(eng.fetch("aligned_seq_shift"), g["aligned_seq_shift"])
            np.testing.assert_array_equal(eng.fetch("ylen"), g["ylen"])
            U = int(g["ymax"])
            if "att_argmax" in g:
                tok = eng.fetch("tok")
                aflip = tok != g["att_argmax"][:, :U]
                assert (g["att_margin"].astype(np.float32)[:, :U][aflip] < 1e-4).all()
            att_err = maxerr(eng.fetch("att_out")[:, ::dt, ::sv], g["att_sample"])
            assert att_err < LOGIT_TOL, att_err
            report["att_logit_err"] = att_err
            n_ok = sum(seqs[0]["hyp"] == g["hyp"][b, : g["hyp_len"][b]].tolist() for b, seqs in enumerate(out))
            report["hyp_exact"] = n_ok
            if "att_argmax" not in g or not aflip.any():
                assert n_ok == len(out)
                np.testing.assert_allclose([s[0]["score"] for s in out], g["score"], atol=2e-2)
    return report


@pytest.mark.parametrize("case,name,strides", [
    (config1_case, "config1", (7, 13, 5)),
    (config2_b8_case, "config2_b8", (10, 50, 4)),
    (config2_b32_case, "config2_b32", (25, 100, 8)),
    (config5_shape_case, "config5_shape", (10, 50, 4)),
])
@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_fp32_parity_gate(case, name, strides, prec, capsys):
    """north_star's gate - token-for-token alignment / indexing, 1e-3 on the log-posteriors - for BOTH engines that claim it:
    the exact-f32 MFMA engine and the split-bf16 engine (three bf16 MFMAs per product; what bench.py times as `parity_engine`)."""
    g = load_golden(name)
    args, state, feats, sizes = case()
    model = build(args, state, prec, capture=True)
    rep = check_against_golden(model, args, feats, sizes, g, *strides, fp32=True)
    with capsys.disabled():
        print(f"\n[parity {prec}] {name}: {rep}")


@pytest.mark.parametrize("case,name", [(config1_case, "config1"), (config2_b8_case, "config2_b8"), (config2_b32_case, "config2_b32")])
def test_bf16x3_fused_path_matches_the_goldens(case, name):
    """The gate above runs with the capture mode on, which keeps the generators on the GEMM + log-softmax kernels (full rows).
    The throughput path of the bf16x3 engine - fused generator kernels (`genmax_x3_kernel`: no logits tensor), projection
    kernel - must give the same integers: best path, shifted alignment, token counts, hypotheses; scores to 2e-2."""
    g = load_golden(name)
    args, state, feats, sizes = case()
    model = build(args, state, "bf16x3", capture=False)
    out = decode(model, args, feats, sizes)
    eng = model._engine
    # the encoder output of this path (conv2 on the LDS-DMA kernel's split form, projection kernel, fused FFN) against the
    # same engine's capture run (generic GEMM kernels throughout): two roundings of the same fp32-grade arithmetic
    ref_model = build(args, state, "bf16x3", capture=True)
    decode(ref_model, args, feats, sizes)
    enc_ref = ref_model._engine.fetch("enc_h")
    assert maxerr(eng.fetch("enc_h_live"), enc_ref) < 2e-4 * max(1.0, float(np.abs(enc_ref).max()))
    flips = eng.fetch("best_paths") != g["best_paths"]
    assert (g["margin"].astype(np.float32)[flips] < 1e-4).all()
    if not flips.any():
        np.testing.assert_array_equal(eng.fetch("aligned_seq_shift"), g["aligned_seq_shift"])
        np.testing.assert_array_equal(eng.fetch("ylen"), g["ylen"])
        U = int(g["ymax"])
        aflip = eng.fetch("tok") != g["att_argmax"][:, :U] if "att_argmax" in g else np.zeros(1, bool)
        if "att_argmax" in g:
            assert (g["att_margin"].astype(np.float32)[:, :U][aflip] < 1e-4).all()
        if not aflip.any():
            assert [s[0]["hyp"] for s in out] == [g["hyp"][b, : g["hyp_len"][b]].tolist() for b in range(len(out))]
            np.testing.assert_allclose([s[0]["score"] for s in out], g["score"], atol=2e-2)


@pytest.mark.parametrize("case,name,strides", [
    (config2_b8_case, "config2_b8", (10, 50, 4)),
    (config2_b32_case, "config2_b32", (25, 100, 8)),
    (config5_shape_case, "config5_shape", (10, 50, 4)),
])
def test_bf16_agreement_report(case, name, strides, capsys):
    g = load_golden(name)
    args, state, feats, sizes = case()
    model = build(args, state, "bf16", capture=True)
    rep = check_against_golden(model, args, feats, sizes, g, *strides, fp32=False)
    with capsys.disabled():
        print(f"\n[agreement bf16] {name}: {rep}")
    # gates at 2x what this build measures (GPUTEST_r02: 0.65-0.67 % flips, 5.2e-3 / 5.8e-3): a regression of the bf16 kernels'
    # rounding (an accumulation moved to bf16, a dropped fp32 residual) shows here; SURVEY 7 predicts 0.65-2.8 % for bf16 alone
    assert rep["flips"] / rep["frames"] < 0.015
    assert rep["ctc_logit_err"] < 1.2e-2
    # the statement that transfers to a trained model's peaked posteriors (src/models/cassnat.py:378-389 is an arg-max per frame):
    # every flip sits on a frame whose fp32 margin is within FLIP_K x the measured logit error, none on a clear-margin frame
    m = rep["margins"]
    assert_flips_explained(m, rep["ctc_logit_err"], FLIP_K, f"bf16 {name}")
    assert m["flips_margin_ge_0.05"] == 0 and m["flips_margin_ge_0.2"] == 0, m


@pytest.mark.parametrize("case,name,strides", [
    (config2_b8_case, "config2_b8", (10, 50, 4)),
    (config2_b32_case, "config2_b32", (25, 100, 8)),
    (config5_shape_case, "config5_shape", (10, 50, 4)),
])
def test_fp16_engine_meets_the_logit_tolerance(case, name, strides, capsys):
    """The fp16 engine = the bf16 engine's kernels with IEEE half operands (libcassnat_hip_f16.so: csrc/common.h, -DCN_OP16_F16):
    same schedules, same speed, 11 significant bits per operand instead of 8.  Against the reference's fp32 goldens its CTC
    log-posteriors stay within north_star's 1e-3; what may still flip is a frame whose fp32 top-2 margin is of the size of that
    error (a random-weight model has such frames, a trained one's posteriors are peaked)."""
    g = load_golden(name)
    args, state, feats, sizes = case()
    model = build(args, state, "fp16", capture=True)
    rep = check_against_golden(model, args, feats, sizes, g, *strides, fp32=False)
    assert model._engine.L.cn_operand16() == b"fp16"
    with capsys.disabled():
        print(f"\n[agreement fp16] {name}: {rep}")
    assert rep["ctc_logit_err"] < 1e-3, rep
    assert rep["flips"] / rep["frames"] < 0.003, rep  # (measured: 4 / 2000 - all four on frames of margin < 1e-4 -, 3 / 8000, 0 / 600)
    m = rep["margins"]
    assert_flips_explained(m, rep["ctc_logit_err"], FLIP_K, f"fp16 {name}")
    assert m["max_flip_margin"] < 2e-3 and m["flips_margin_ge_0.05"] == 0 and m["flips_margin_ge_0.2"] == 0, m


def test_the_two_libraries_refuse_each_others_engines():
    args, state, feats, sizes = tiny_case()
    assert hip.lib().cn_operand16() == b"bf16" and hip.lib("f16").cn_operand16() == b"fp16"
    cfg = hip.Engine(args, precision="fp16", max_batch=2, max_frames=64).cfg
    h = hip.C.c_void_p()
    assert hip.lib().cn_model_create(hip.C.byref(cfg), hip.C.byref(h)) != 0 and b"libcassnat_hip_f16.so" in hip.lib().cn_last_error()
    cfg.precision = hip.PRECISION["bf16"]
    assert hip.lib("f16").cn_model_create(hip.C.byref(cfg), hip.C.byref(h)) != 0 and b"fp16 engine only" in hip.lib("f16").cn_last_error()


def test_full_size_properties_bf16():
    """BASELINE config 2 at full size: size-independent properties of the integer outputs."""
    args, state, feats, sizes = config2_b32_case()
    model = build(args, state, "bf16")
    out = decode(model, args, feats, sizes)
    eng = model._engine
    shift, ylen, best, iv = (eng.fetch(n) for n in ("aligned_seq_shift", "ylen", "best_paths", "intervals"))
    assert (shift[:, 0] == 0).all()
    np.testing.assert_array_equal((shift != 0).sum(1) + 1, ylen)
    # no two adjacent frames of the collapsed path carry the same non-blank token
    col = shift[:, 1:]
    assert not ((col[:, 1:] == col[:, :-1]) & (col[:, 1:] != 0) & (best[:, 1:-1] == best[:, :-2])).any()
    # trigger rows tile the frame axis: row u ends where row u+1 starts, rows beyond ylen are empty
    for b in range(shift.shape[0]):
        n = ylen[b]
        assert iv[b, 0, 0] == 0 and iv[b, n - 1, 1] == shift.shape[1]
        np.testing.assert_array_equal(iv[b, : n - 1, 1], iv[b, 1:n, 0])
        assert (iv[b, n:, :2] == 0).all()
        assert len(out[b][0]["hyp"]) == 1 + min(n + 1, ylen.max())
    # idempotence: decoding the same batch again gives identical hypotheses
    again = decode(model, args, feats, sizes)
    assert [s[0]["hyp"] for s in again] == [s[0]["hyp"] for s in out]


def test_decode_rejects_oversize_batch():
    args, state, feats, sizes = tiny_case()
    eng = hip.Engine(args, precision="fp32", max_batch=2, max_frames=64)
    eng.load_state(state, torch.zeros(5000, args.d_model))
    f = torch.from_numpy(feats).cuda()
    hyp = torch.empty(3, 40, dtype=torch.int32, device="cuda")
    with pytest.raises(hip.HipError, match="workspace buffer .* needs"):
        eng.decode(f, torch.from_numpy(sizes).cuda(), hip.Engine.make_opts(args), hyp,
                   torch.empty(3, dtype=torch.int32, device="cuda"), torch.empty(3, dtype=torch.float64, device="cuda"))


def test_weight_blob_handoff_reproduces_rank0_engine():
    """The multi-GPU weight path minus the RCCL call: a layout-only engine that receives rank 0's packed blob
    byte-for-byte must decode identically."""
    from cassnat_asr_public_amd import dist as cdist

    args, state, feats, sizes = config1_case()
    src = build(args, state, "bf16")
    out_a = decode(src, args, feats, sizes)
    dst = make_model(args.input_size, args).cuda()  # xavier-initialised, never loaded
    eng = dst.build_engine(1, feats.shape[1], with_weights=False)
    pa, na = src._engine.weight_blob()
    pb, nb = eng.weight_blob()
    assert na == nb and na > 4_000_000
    ta = torch.as_tensor(cdist._CudaBlob(pa, na), device="cuda")
    tb = torch.as_tensor(cdist._CudaBlob(pb, nb), device="cuda")
    assert ta.data_ptr() == pa and tb.data_ptr() == pb  # zero-copy views
    tb.copy_(ta)
    torch.cuda.synchronize()
    out_b = decode(dst, args, feats, sizes)
    assert [s[0]["hyp"] for s in out_a] == [s[0]["hyp"] for s in out_b]
    assert [s[0]["score"] for s in out_a] == [s[0]["score"] for s in out_b]


@pytest.mark.parametrize("prec", ["fp32", "fp16"])
@pytest.mark.parametrize("batch_size", [3, 1])
def test_decode_asr_cli_end_to_end(tmp_path, batch_size, prec):
    """decode_asr.py --task cassnat on a synthetic Kaldi table: result file == oracle hypotheses as text.  batch_size 3: one
    batch, the plain loop; batch_size 1: three batches through the decode pipelines (pipeline.DecodePipelines).  fp16: the engine
    of the second library (half-precision operands) behind the same command line - on this model (no near-tie frame) the same text."""
    import yaml

    from cassnat_asr_public_amd.bin import decode_asr
    from cassnat_asr_public_amd.data import kaldi_io
    from oracle import cassnat_oracle as orc

    args, state, feats, sizes = tiny_case()
    lengths = [61, 50, 37]
    mats = [(f"spk-utt{b}", feats[b, :n]) for b, n in enumerate(lengths)]
    scp = str(tmp_path / "feats.scp")
    kaldi_io.write_ark_scp(str(tmp_path / "feats.ark"), scp, mats)
    vocab_file = tmp_path / "vocab.txt"
    vocab_file.write_text("".join(f"w{i}\n" for i in range(args.vocab_size - 4)))
    ckpt = str(tmp_path / "model.mdl")
    torch.save({"model_state": {"module." + k: torch.from_numpy(v) for k, v in state.items()}}, ckpt)
    conf = {k: getattr(args, k) for k in ("input_size", "d_model", "n_head", "d_encff", "d_decff", "d_ff", "N_enc", "N_extra",
                                          "N_self_dec", "N_mix_dec", "model_type", "n_features", "left_ctx", "right_ctx",
                                          "skip_frame", "padding_idx", "beam_width", "length_penalty", "use_trigger")}
    conf.update(vocab_file=str(vocab_file), use_gpu=True)
    cfg = tmp_path / "decode.yaml"
    cfg.write_text(yaml.safe_dump(conf))
    result = str(tmp_path / "token_results.txt")
    rc = decode_asr.main(["--task", "cassnat", "--test_config", str(cfg), "--data_path", scp, "--resume_model", ckpt,
                          "--result_file", result, "--batch_size", str(batch_size), "--hip_precision", prec,
                          "--load_data_workers", "0"])
    assert rc == 0
    index2word = {i + 4: f"w{i}" for i in range(args.vocab_size - 4)}
    if batch_size == 3:
        hyps = orc.decode_nast(state, feats, sizes, args)["hyps"]
    else:  # the reference's greedy finish reads min(ylen + 1, U of the BATCH) rows: an utterance decoded alone ends one token earlier
        hyps = [orc.decode_nast(state, feats[b : b + 1, :n], np.ones(1, np.float32), args)["hyps"][0] for b, n in enumerate(lengths)]
    expect = [f"spk-utt{b} " + " ".join(orc.hyp_to_text(h, index2word)) for b, h in enumerate(hyps)]
    assert open(result).read().splitlines() == expect


def test_decode_asr_cli_with_global_cmvn_on_the_device(tmp_path):
    """decode_asr with `use_cmvn`: the pipelined decoder takes raw rows from the archive and normalises them on the device
    (hip.cmvn_) - the result file equals the one with the CMVN in the dataset (--hip_device_cmvn 0), line for line, and the
    oracle's on the normalised features."""
    import yaml

    from cassnat_asr_public_amd.bin import decode_asr
    from cassnat_asr_public_amd.data import kaldi_io
    from oracle import cassnat_oracle as orc

    args, state, feats, sizes = tiny_case()
    lengths = [61, 50, 37, 61, 12]
    rng = np.random.default_rng(1)
    raw = [(rng.standard_normal((n, feats.shape[2])) * 2.5 + 0.7).astype(np.float32) for n in lengths]
    mats = [(f"spk-utt{b}", m) for b, m in enumerate(raw)]
    scp = str(tmp_path / "feats.scp")
    kaldi_io.write_ark_scp(str(tmp_path / "feats.ark"), scp, mats)
    allf = np.vstack(raw).astype(np.float64)
    stats = np.zeros((2, feats.shape[2] + 1))
    stats[0, :-1], stats[0, -1], stats[1, :-1] = allf.sum(0), len(allf), (allf ** 2).sum(0)
    kaldi_io.write_ark_scp(str(tmp_path / "cmvn.ark"), str(tmp_path / "cmvn.scp"), [("global", stats)])
    cmvn_spec = kaldi_io.read_scp(str(tmp_path / "cmvn.scp"))[0][1]
    vocab_file = tmp_path / "vocab.txt"
    vocab_file.write_text("".join(f"w{i}\n" for i in range(args.vocab_size - 4)))
    ckpt = str(tmp_path / "model.mdl")
    torch.save({"model_state": {"module." + k: torch.from_numpy(v) for k, v in state.items()}}, ckpt)
    conf = {k: getattr(args, k) for k in ("input_size", "d_model", "n_head", "d_encff", "d_decff", "d_ff", "N_enc", "N_extra",
                                          "N_self_dec", "N_mix_dec", "model_type", "n_features", "left_ctx", "right_ctx",
                                          "skip_frame", "padding_idx", "beam_width", "length_penalty", "use_trigger")}
    conf.update(vocab_file=str(vocab_file), use_gpu=True, use_cmvn=True, global_cmvn=cmvn_spec)
    cfg = tmp_path / "decode.yaml"
    cfg.write_text(yaml.safe_dump(conf))
    out = {}
    for dev in (1, 0):
        result = str(tmp_path / f"token_results_{dev}.txt")
        rc = decode_asr.main(["--task", "cassnat", "--test_config", str(cfg), "--data_path", scp, "--resume_model", ckpt,
                              "--result_file", result, "--batch_size", "1", "--hip_precision", "fp32", "--load_data_workers", "0",
                              "--hip_device_cmvn", str(dev)])
        assert rc == 0
        out[dev] = open(result).read().splitlines()
    assert out[1] == out[0]
    mean = stats[0, :-1] / stats[0, -1]  # (as SpeechDataset._load_cmvn derives them from the Kaldi stats matrix)
    std = np.sqrt(stats[1, :-1] / stats[0, -1] - mean ** 2)
    index2word = {i + 4: f"w{i}" for i in range(args.vocab_size - 4)}
    hyps = [orc.decode_nast(state, ((m - mean) / std).astype(np.float32)[None], np.ones(1, np.float32), args)["hyps"][0] for m in raw]
    assert out[1] == [f"spk-utt{b} " + " ".join(orc.hyp_to_text(h, index2word)) for b, h in enumerate(hyps)]


@pytest.mark.parametrize("prec", ["bf16", "fp16", "bf16x3"])
def test_bf16_production_path_against_golden(prec, capsys):
    """The path bench.py times (bf16, fused FFN sublayer, fused generator arg-max, no capture) on the benchmark workload:
    CTC arg-max agreement with the reference and hypothesis agreement on utterances whose alignment did not flip.  fp16: the same
    kernels with half-precision operands (the second library) - an order of magnitude fewer flips, every one on a frame whose fp32
    margin is below twice the 1e-3 tolerance, and most utterances' hypotheses are the reference's token for token."""
    g = load_golden("config2_b32")
    args, state, feats, sizes = config2_b32_case()
    model = build(args, state, prec, capture=False)
    out = decode(model, args, feats, sizes)
    eng = model._engine
    best, shift, ylen = eng.fetch("best_paths"), eng.fetch("aligned_seq_shift"), eng.fetch("ylen")
    flips = best != g["best_paths"]
    same_align = (shift == g["aligned_seq_shift"]).all(1)
    tok_agree = []
    for b in np.flatnonzero(same_align):
        ref = g["hyp"][b, : g["hyp_len"][b]].tolist()
        got = out[b][0]["hyp"]
        assert len(got) == len(ref)
        tok_agree.append(np.mean([x == y for x, y in zip(got, ref)]))
    rep = dict(ctc_flip_rate=float(flips.mean()), utts_with_identical_alignment=int(same_align.sum()),
               token_agreement_on_those=float(np.mean(tok_agree)) if tok_agree else None,
               ylen_max_abs_diff=int(np.abs(ylen - g["ylen"]).max()))
    # the production path keeps no log-posteriors (its generator is fused with the arg-max), but they follow from what it does
    # keep: the encoder output as the generator's 16-bit operand (`enc_h_live`) times the weight rounded to the same type,
    # accumulated here in float64 (the kernel: fp32, another order) - against the reference's log-posteriors (sampled in the fixture)
    enc = torch.from_numpy(eng.fetch("enc_h_live")).double()
    t16 = {"fp16": torch.float16, "bf16": torch.bfloat16}.get(prec, torch.float32)  # (split-bf16: hi + lo carry the fp32 value to 2^-17)
    w = torch.from_numpy(state["ctc_generator.proj.weight"]).to(t16).double()
    logp = torch.log_softmax(enc @ w.T + torch.from_numpy(state["ctc_generator.proj.bias"]).double(), -1)
    rep["ctc_logit_err"] = float((logp[:, ::25, ::100].float() - torch.from_numpy(g["ctc_sample"])).abs().max())
    assert (logp.argmax(-1).numpy() != best).mean() < 2e-4  # (the reconstruction IS the kernel's arithmetic up to summation order)
    with capsys.disabled():
        print(f"\n[{prec} production path] {rep}")
    assert rep["ctc_flip_rate"] < 0.02            # 0.65-2.8 % expected from bf16 rounding alone
    assert rep["ylen_max_abs_diff"] <= 3
    if tok_agree:
        assert rep["token_agreement_on_those"] > 0.9
    if prec == "bf16x3":  # the parity engine on its production kernels (fused row chain, conv2 in the MIX arithmetic, fused generators)
        assert not flips.any() and same_align.all() and rep["token_agreement_on_those"] == 1.0 and rep["ctc_logit_err"] < 5e-5, rep
    if prec == "fp16":  # measured: 3 flips of 8000 (margins <= 3.7e-4), 29 of 32 alignments identical
        assert rep["ctc_flip_rate"] < 0.001 and rep["utts_with_identical_alignment"] >= 26 and rep["token_agreement_on_those"] > 0.99
        assert (g["margin"].astype(np.float32)[flips] < 2e-3).all()
        assert rep["ctc_logit_err"] < 1e-3, rep  # north_star's tolerance on the encoder logits, on the path the benchmark times


# ------------------------------------------------------------------------------------------- ESA + LM ranking (8f rank 3)
@pytest.mark.parametrize("which,prec", [("esa_tiny", "fp32"), ("esa_config2", "fp32"), ("esa_config2", "bf16"),
                                        ("esa_tiny", "bf16x3"), ("esa_config2", "bf16x3"),  # the split-bf16 engine: gated like fp32
                                        ("esa_config2", "fp8"),  # (the ranking LM of an fp8 recogniser runs bf16)
                                        ("esa_config2", "fp16"),  # (recogniser and ranking LM as engines of the half-precision library)
                                        ("esa_conf_tiny", "fp32")])  # (conformer blocks under ESA: the shipped YAML's combination)
def test_esa_sampling_with_lm_ranking(which, prec, capsys):
    """sample_num = 4 alignments per utterance (random draws = the fixture's, i.e. the reference's torch.randint stream),
    TransformerLM ranking on the device.  fp32 and bf16x3: hypotheses (up to the position the reference reads from a masked row, see
    tests/test_oracle_golden.py) and scores equal the reference's; bf16: reported."""
    from conftest import esa_case
    from cassnat_asr_public_amd.models.lm import make_model as make_lm

    g = load_golden(which)
    args, lm_args, state, lm_state, feats, sizes = esa_case(which)
    args.esa_select = g["select"]
    lm_args.hip_precision = prec
    model = build(args, state, prec)
    lm = make_lm(lm_args).cuda()
    with torch.no_grad():
        for k, p in lm.named_parameters():
            p.copy_(torch.from_numpy(lm_state[k]))
    src = torch.from_numpy(feats)
    with torch.no_grad():
        out, _ = model.beam_decode(src.cuda(), (src[:, :, 0] != 0).unsqueeze(1).cuda(), torch.from_numpy(sizes).cuda(), Vocab, args, lm)
    same = 0
    for b, seqs in enumerate(out):
        h, ref = seqs[0]["hyp"], g["hyp"][b, : g["hyp_len"][b]].tolist()
        n = len(h) - 1 if len(h) == len(ref) and h[-1] == 0 and ref[-1] != 0 else len(h)  # the masked-row tie token
        same += h[:n] == ref[:n] and len(h) == len(ref)
    with capsys.disabled():
        print(f"\n[ESA {prec}] {which}: {same}/{len(out)} hypotheses identical, scores {[round(s[0]['score'], 3) for s in out]} vs {np.round(g['score'], 3).tolist()}")
    if prec in ("fp32", "bf16x3"):
        assert same == len(out)
        np.testing.assert_allclose([s[0]["score"] for s in out], g["score"], rtol=1e-5, atol=2e-3)
    elif prec == "fp8":  # e4m3 encoder products: the path runs with its LM beside it, the ranking score stays of the reference's order
        assert lm.hip_precision == "bf16"
        np.testing.assert_allclose([s[0]["score"] for s in out], g["score"], rtol=0.2, atol=2.0)
    else:  # bf16: another of the sampled alignments may win the ranking; its score stays within a few percent (measured: 1.6 %)
        np.testing.assert_allclose([s[0]["score"] for s in out], g["score"], rtol=0.05, atol=0.5)
    if which in ("esa_tiny", "esa_conf_tiny"):  # the samples go through the decoder side in groups: any group size gives the same answer
        for group in (1, 3):
            args.hip_esa_group = group
            with torch.no_grad():
                out_g, _ = model.beam_decode(src.cuda(), (src[:, :, 0] != 0).unsqueeze(1).cuda(), torch.from_numpy(sizes).cuda(), Vocab, args, lm)
            assert [s[0]["hyp"] for s in out_g] == [s[0]["hyp"] for s in out]
            np.testing.assert_allclose([s[0]["score"] for s in out_g], [s[0]["score"] for s in out], rtol=1e-6, atol=1e-5)


def test_config5_fp8_encoder_products(capsys):
    """BASELINE config 5 (Aishell-1-sized vocabulary, e4m3fn encoder products with per-tensor scales): no 1e-3 gate exists for
    it - the run reports its alignment agreement and logit error against the reference's fp32 golden, next to
    the bf16 engine's on the same weights, and is gated loosely (finite, most frames agree, error of the fp8 order).
    ``hip_fp8_scope`` chooses which products take e4m3 operands: every one adds its ~5 % of relative noise, so fewer products
    = fewer flips (and less of the speed-up: DESIGN.md 5d)."""
    g = load_golden("config5_shape")
    args, state, feats, sizes = config5_shape_case()
    st, sv = 10, 50  # strides of the golden's log-posterior sample
    rows = {}
    for prec, scope in (("bf16", "all"), ("fp8", "all"), ("fp8", "conv2+ffn"), ("fp8", "conv2+ffn:8"), ("fp8", "conv2"), ("fp8", "ffn")):
        args.hip_fp8_scope = scope
        model = build(args, state, prec, capture=True)
        out = decode(model, args, feats, sizes)
        eng = model._engine
        best, ctc = eng.fetch("best_paths"), eng.fetch("ctc_out")
        assert np.isfinite(ctc).all() and all(np.isfinite(s[0]["score"]) for s in out)
        rows[prec if prec == "bf16" else f"fp8[{scope}]"] = dict(
            flips=float((best != g["best_paths"]).mean()), err=maxerr(ctc[:, ::st, ::sv], g["ctc_sample"]),
            hyp=sum(s[0]["hyp"] == g["hyp"][b, : g["hyp_len"][b]].tolist() for b, s in enumerate(out)),
            margins=flips_by_margin(best, g["best_paths"], g["margin"]))
    args.hip_fp8_scope = "all"
    with capsys.disabled():
        print(f"\n[config 5] V={args.vocab_size}, {g['best_paths'].size} frames, against the fp32 reference: "
              + "; ".join(f"{k}: argmax flips {v['flips']:.4f}, max |d log-posterior| {v['err']:.4f}, hypotheses identical "
                          f"{v['hyp']}/{len(g['hyp'])}, largest flip margin {v['margins']['max_flip_margin']:.4f} "
                          f"(= {v['margins']['max_flip_margin'] / max(v['err'], 1e-9):.2f} x err), flips at margin >= 0.05: "
                          f"{v['margins']['flips_margin_ge_0.05']}/{v['margins']['frames_margin_ge_0.05']}, >= 0.2: "
                          f"{v['margins']['flips_margin_ge_0.2']}/{v['margins']['frames_margin_ge_0.2']}" for k, v in rows.items()))
    # gates at about 2x the measured values
    assert rows["fp8[all]"]["flips"] < 0.07 and rows["fp8[all]"]["err"] < 0.065
    assert rows["fp8[conv2]"]["flips"] < 0.04 and rows["fp8[conv2]"]["err"] < 0.04
    assert rows["bf16"]["flips"] < 0.015 and rows["bf16"]["err"] < 1.2e-2
    for k, v in rows.items():  # no scope is worse than all of them together (beyond the 600-frame fixture's resolution)
        if k.startswith("fp8["):
            assert v["flips"] <= rows["fp8[all]"]["flips"] + 0.012, (k, v)
    # margin-conditioned: whatever the rate on this near-flat fixture, no engine flips a frame whose fp32 margin exceeds
    # FLIP_K x its own measured logit error, and none flips a frame of margin >= 0.2 (a trained model's typical frame)
    for k, v in rows.items():
        assert_flips_explained(v["margins"], v["err"], FLIP_K, k)
        assert v["margins"]["flips_margin_ge_0.2"] == 0 and v["margins"]["flips_margin_ge_0.05"] == 0, (k, v["margins"])


def test_fp8_scope_strings():
    assert hip.parse_fp8_scope("all") == (0, 0) and hip.parse_fp8_scope(None) == (0, 0)
    assert hip.parse_fp8_scope("conv2") == (1, 0) and hip.parse_fp8_scope("conv2+linear+ffn") == (7, 0)
    assert hip.parse_fp8_scope("conv2 + ffn:8") == (5, 8) and hip.parse_fp8_scope("ffn") == (4, 0)
    for bad in ("linear", "ffn+linear", "conv", "conv2:3", "ffn:-1"):
        with pytest.raises(ValueError):
            hip.parse_fp8_scope(bad)
    # the library refuses what the parser refuses
    args = synth.make_args("config2")
    args.fp8_scope = 2
    with pytest.raises(hip.HipError, match="fp8_scope"):
        hip.Engine(args, precision="fp8", max_batch=2, max_frames=64)


def test_fp8_engine_without_row_chains_keeps_the_unfused_layer():
    """d_ff not a multiple of 256: the e4m3 feed-forward form of the chain kernel does not apply and the fp8 engine runs round 2's
    layer - four e4m3 products per encoder layer as separate launches (run_enc_layer_fp8) - behind the same e4m3 conv front-end.
    Held against the bf16 engine on the same weights, loosely (the path must stay alive and of the fp8 order)."""
    # 40-dim features as well: conv1's VALU form of the e4m3 image (F1 + 2 < 32), conv2's e4m3 form with bf16 output rows, bf16
    # linear_out (K = 10 x 256, not the 5120 its e4m3 form is unrolled for)
    args = synth.make_args("config2", d_encff=384, N_enc=3, input_size=40, n_features=40)
    state = synth.make_state(args, seed=11, blank_bias=0.35)
    lens = synth.ragged_lengths(3, 400, 200, seed=2)
    feats, sizes = synth.make_feats(3, 400, 40, lengths=lens, seed=5)
    got = {}
    for prec in ("bf16", "fp8"):
        model = build(args, state, prec, capture=True)
        out = decode(model, args, feats, sizes)
        eng = model._engine
        got[prec] = (eng.fetch("best_paths"), eng.fetch("ctc_out"))
        assert np.isfinite(got[prec][1]).all() and all(np.isfinite(s[0]["score"]) for s in out)
    flips = float((got["fp8"][0] != got["bf16"][0]).mean())
    err = maxerr(got["fp8"][1][:, ::7, ::50], got["bf16"][1][:, ::7, ::50])
    # measured 15.7 % flips, 0.060 (three layers of random weights, every product of a layer in e4m3: arg-max is fragile here)
    assert flips < 0.3 and err < 0.12, (flips, err)


def test_esa_group_rows_beyond_one_alignment_per_utterance():
    """ESA at a size where the decoder side of one group (B x 16 alignments x U rows) has many times the B x (T' + 1) rows
    the logits buffer holds: the engines without the fused generator kernel (fp32, bf16x3) take the rows through it in chunks
    (they used to write past it).  No golden at this size: the two parity-grade engines must agree with each other - same
    draws, same hypotheses, scores to fp32 accuracy."""
    from cassnat_asr_public_amd import synth as sy
    from cassnat_asr_public_amd.models.cassnat import make_model
    from cassnat_asr_public_amd.models.lm import make_model as make_lm

    B, T, samples = 4, 400, 16
    got = {}
    for prec in ("fp32", "bf16x3"):
        args = sy.make_args("config2", sample_num=samples, rank_model="lm", threshold=0.9)
        args.hip_precision = prec
        args.hip_max_batch, args.hip_max_frames = B, T
        lm_args = sy.make_args_lm("lm_small", vocab_size=args.vocab_size)
        lm_args.hip_precision = prec
        state = sy.make_state(args, seed=0, blank_bias=sy.BENCH_BLANK_BIAS)
        lm_state = sy.make_state(lm_args, seed=9, gain=2.0)
        model = make_model(args.input_size, args).cuda()
        lm = make_lm(lm_args).cuda()
        with torch.no_grad():
            for k, p in model.named_parameters():
                p.copy_(torch.from_numpy(state[k]))
            for k, p in lm.named_parameters():
                p.copy_(torch.from_numpy(lm_state[k]))
        fh, sh = sy.make_feats(B, T, args.input_size, seed=77)
        src, sizes = torch.from_numpy(fh).cuda(), torch.from_numpy(sh).cuda()
        torch.manual_seed(3)
        with torch.no_grad():
            out, _ = model.beam_decode(src, (src[:, :, 0] != args.padding_idx).unsqueeze(1), sizes, Vocab, args, lm)
        torch.cuda.synchronize()
        got[prec] = ([o[0]["hyp"] for o in out], [o[0]["score"] for o in out])
        assert max(len(h) for h in got[prec][0]) * 16 > T // 4 + 1  # the group's rows do exceed one alignment's capacity
    assert got["fp32"][0] == got["bf16x3"][0]
    np.testing.assert_allclose(got["fp32"][1], got["bf16x3"][1], rtol=1e-4, atol=5e-3)


def _task_on_synthetic_table(tmp_path, args, state, mats, batch_size, **conf_over):
    """A CassNATTask over a synthetic Kaldi table, built the way decode_asr.main builds it."""
    import yaml

    from cassnat_asr_public_amd.data import kaldi_io
    from cassnat_asr_public_amd.tasks import CassNATTask
    from cassnat_asr_public_amd.utils.parser import DecodeParser

    scp = str(tmp_path / "feats.scp")
    kaldi_io.write_ark_scp(str(tmp_path / "feats.ark"), scp, mats)
    vocab_file = tmp_path / "vocab.txt"
    vocab_file.write_text("".join(f"w{i}\n" for i in range(args.vocab_size - 4)))
    ckpt = str(tmp_path / "model.mdl")
    torch.save({"model_state": {"module." + k: torch.from_numpy(v) for k, v in state.items()}}, ckpt)
    conf = {k: getattr(args, k) for k in ("input_size", "d_model", "n_head", "d_encff", "d_decff", "d_ff", "N_enc", "N_extra",
                                          "N_self_dec", "N_mix_dec", "model_type", "n_features", "left_ctx", "right_ctx",
                                          "skip_frame", "padding_idx", "beam_width", "length_penalty", "use_trigger")}
    conf.update(vocab_file=str(vocab_file), use_gpu=True, **conf_over)
    cfg = tmp_path / "decode.yaml"
    cfg.write_text(yaml.safe_dump(conf))
    a = DecodeParser().get_args(["--task", "cassnat", "--test_config", str(cfg), "--data_path", scp, "--resume_model", ckpt,
                                 "--result_file", str(tmp_path / "res.txt"), "--batch_size", str(batch_size), "--hip_precision", "fp32",
                                 "--load_data_workers", "0"])
    for k, v in conf.items():
        setattr(a, k, v)
    a.test_paths = [{"name": "test", "scp_path": scp}]
    a.rank = 0
    return CassNATTask("test", a), a


def test_cached_pipelines_follow_the_models_weights(tmp_path):
    """A task keeps its decode pipelines (engines with their own packed weights) across decode() calls.  After the parameters
    change - in place, through .data + invalidate_engine, or by load_state_dict - the next decode() must not run on the old
    weights: its result file equals the plain loop's on the new ones (the cache key ignored the weights)."""
    args, state, feats, sizes = tiny_case()
    lengths = [61, 50, 37, 44]
    mats = [(f"spk-utt{b}", feats[b % 3, :n]) for b, n in enumerate(lengths)]
    task, a = _task_on_synthetic_table(tmp_path, args, state, mats, batch_size=1)

    def run(pipelines, name):
        a.hip_pipelines, a.result_file = pipelines, str(tmp_path / name)
        task.decode(a)
        return open(a.result_file).read().splitlines()

    first = run(2, "p0.txt")
    assert first == run(1, "l0.txt")
    pipes0 = task._pipes
    assert run(2, "p0b.txt") == first and task._pipes is pipes0  # unchanged weights: the pipelines are kept
    rng = np.random.default_rng(5)
    w = task.model.att_generator.proj.weight
    with torch.no_grad():  # 1. an in-place edit (bumps the tensor version)
        w.copy_(torch.from_numpy(rng.standard_normal(tuple(w.shape)).astype(np.float32)))
    second = run(2, "p1.txt")
    assert task._pipes is not pipes0
    assert second == run(1, "l1.txt") and second != first
    pipes1 = task._pipes
    w.data = torch.from_numpy(rng.standard_normal(tuple(w.shape)).astype(np.float32))  # 2. through .data: invisible to versions
    task.model.invalidate_engine()
    third = run(2, "p2.txt")
    assert task._pipes is not pipes1
    assert third == run(1, "l2.txt") and third != second
    sd = {k: torch.from_numpy(v) for k, v in state.items()}  # 3. load_state_dict: back to the checkpoint
    task.model.load_state_dict(sd)
    assert run(2, "p3.txt") == first
    task.close()


def test_decode_asr_cli_cmvn_on_a_float64_archive(tmp_path):
    """A `DM` (float64) archive: the reference normalises the float64 values and rounds once.  The device form would round the
    features to float32 BEFORE normalising, so it must not be chosen: the dataset keeps the CMVN on the host, and the result
    equals the oracle's on features normalised in float64."""
    from cassnat_asr_public_amd.data import kaldi_io
    from oracle import cassnat_oracle as orc

    args, state, feats, sizes = tiny_case()
    lengths = [61, 50, 37, 12]
    rng = np.random.default_rng(2)
    raw = [rng.standard_normal((n, feats.shape[2])) * 2.5 + 0.7 for n in lengths]  # float64 matrices
    allf = np.vstack(raw)
    stats = np.zeros((2, feats.shape[2] + 1))
    stats[0, :-1], stats[0, -1], stats[1, :-1] = allf.sum(0), len(allf), (allf ** 2).sum(0)
    kaldi_io.write_ark_scp(str(tmp_path / "cmvn.ark"), str(tmp_path / "cmvn.scp"), [("global", stats)])
    cmvn_spec = kaldi_io.read_scp(str(tmp_path / "cmvn.scp"))[0][1]
    task, a = _task_on_synthetic_table(tmp_path, args, state, [(f"spk-utt{b}", m) for b, m in enumerate(raw)], batch_size=1,
                                       use_cmvn=True, global_cmvn=cmvn_spec)
    assert kaldi_io.mat_dtype(kaldi_io.read_scp(str(tmp_path / "feats.scp"))[0][1]) == np.float64
    assert not task.test_loader.dataset.can_defer_cmvn()
    a.hip_pipelines = 2
    task.decode(a)
    assert task._pipes.cmvn is None  # the pipelines received normalised batches
    got = open(a.result_file).read().splitlines()
    task.close()
    mean = stats[0, :-1] / stats[0, -1]
    std = np.sqrt(stats[1, :-1] / stats[0, -1] - mean ** 2)
    index2word = {i + 4: f"w{i}" for i in range(args.vocab_size - 4)}
    hyps = [orc.decode_nast(state, ((m - mean) / std).astype(np.float32)[None], np.ones(1, np.float32), args)["hyps"][0] for m in raw]
    assert got == [f"spk-utt{b} " + " ".join(orc.hyp_to_text(h, index2word)) for b, h in enumerate(hyps)]


# ------------------------------------------------------------------------------------------- branches closed in round 4
@pytest.mark.parametrize("prec", ["fp32", "bf16x3", "bf16"])
@pytest.mark.parametrize("which", ["tiny_notrigger", "config2_notrigger"])
def test_use_trigger_false(which, prec):
    """args.use_trigger == False (src/models/cassnat.py:469-473): the extractor attends over every valid frame (trigger_mask =
    src_mask) and the row counts are best_path_align's own.  fp32 / bf16x3: stages, hypotheses and scores equal the reference's;
    bf16 (config-2 size: row chains and the fused kernels on this option): runs, same row counts unless the alignment flipped."""
    from conftest import notrigger_case

    g = load_golden(which)
    args, state, feats, sizes = notrigger_case(which)
    model = build(args, state, prec, capture=(prec != "bf16"))
    out = decode(model, args, feats, sizes)
    eng = model._engine
    if prec == "bf16":
        if (eng.fetch("aligned_seq_shift") == g["aligned_seq_shift"]).all():
            np.testing.assert_array_equal(eng.fetch("ylen"), g["ylen0"])
            assert [len(s[0]["hyp"]) for s in out] == g["hyp_len"].tolist()
        return
    np.testing.assert_array_equal(eng.fetch("aligned_seq_shift"), g["aligned_seq_shift"])
    np.testing.assert_array_equal(eng.fetch("ylen"), g["ylen0"])
    assert int(eng.fetch("ymax")[0]) == int(g["ylen0"].max())
    if which == "tiny_notrigger":
        for name, tol in (("ac_embed", 5e-4), ("pred_embed", 5e-4), ("dec_h", 1e-4), ("att_out", LOGIT_TOL)):
            assert maxerr(eng.fetch(name), g[name]) < tol, name
    else:
        assert maxerr(eng.fetch("att_out")[:, ::3, ::25], g["att_sample"]) < LOGIT_TOL
    for b, seqs in enumerate(out):
        assert seqs[0]["hyp"] == g["hyp"][b, : g["hyp_len"][b]].tolist()
        assert abs(seqs[0]["score"] - g["score"][b]) < 2e-2


@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_esa_finished_with_beam_width_3(prec):
    """ESA (sample_num 4, TransformerLM ranking) with beam_width 3 (src/models/cassnat.py:556-561, 574-637): every beam of every
    utterance up to the position the reference reads from an all-zero row (an implementation-defined torch.topk of equal
    values), scores in full."""
    from conftest import esa_beam3_case
    from cassnat_asr_public_amd.models.lm import make_model as make_lm

    g = load_golden("esa_beam3_tiny")
    args, lm_args, state, lm_state, feats, sizes = esa_beam3_case()
    args.esa_select = g["select"]
    lm_args.hip_precision = prec
    model = build(args, state, prec)
    lm = make_lm(lm_args).cuda()
    with torch.no_grad():
        for k, p in lm.named_parameters():
            p.copy_(torch.from_numpy(lm_state[k]))
    src = torch.from_numpy(feats)
    with torch.no_grad():
        out, _ = model.beam_decode(src.cuda(), (src[:, :, 0] != 0).unsqueeze(1).cuda(), torch.from_numpy(sizes).cuda(), Vocab, args, lm)
    ylen = model._engine.fetch("ylen")
    for b, beams in enumerate(out):
        assert len(beams) == 3
        n = int(ylen[b]) + 1
        for j, s in enumerate(beams):
            assert len(s["hyp"]) == g["beam_len"][b, j]
            assert s["hyp"][:n] == g["beam_hyp"][b, j, :n].tolist(), (b, j)
            assert abs(s["score"] - g["beam_score"][b, j]) < 2e-3
