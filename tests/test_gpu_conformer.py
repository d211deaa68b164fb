"""GPU parity of the conformer CASS-NAT variants (use_conv_enc / use_conv_dec, relative positions) through
the drop-in ``models.cassnat`` API.  Goldens are the reference's own CassNAT.beam_decode outputs with forward hooks
(tests/golden/conf_*.npz, oracle/make_goldens.py).  fp32 engine = gate (every captured stage, alignment and hypotheses
exact); bf16 engine = reported agreement."""
import numpy as np
import pytest
import torch

from conftest import conf_small_case, conf_tiny_case, load_golden
from cassnat_asr_public_amd import synth
from cassnat_asr_public_amd.models.cassnat import make_model
from test_gpu_pipeline import LOGIT_TOL, Vocab, build, decode, maxerr

pytestmark = pytest.mark.gpu


def test_conformer_parameter_names_match_reference():
    for ov in (dict(), dict(use_conv_enc=False)):
        args = synth.make_args("tiny_conf", **ov)
        model = make_model(80, args)
        assert {k: tuple(v.shape) for k, v in model.named_parameters()} == dict(synth.param_shapes_conformer(args))
        assert [k for k, _ in model.named_parameters()] == list(synth.param_shapes_conformer(args).keys())


def test_conformer_tiny_fp32_every_stage():
    g = load_golden("conf_tiny")
    args, state, feats, sizes = conf_tiny_case()
    model = build(args, state, "fp32", capture=True)
    out = decode(model, args, feats, sizes)
    eng = model._engine
    for name in ["x_embed", "enc_layer0", "enc_layer1", "enc_h", "ctc_out", "ac_embed", "pred_embed", "dec_h", "att_out"]:
        scale = max(1.0, float(np.abs(g[name]).max()))
        assert maxerr(eng.fetch(name), g[name]) < 1e-4 * scale, name
    np.testing.assert_array_equal(eng.fetch("aligned_seq_shift"), g["aligned_seq_shift"])
    np.testing.assert_array_equal(eng.fetch("ylen"), g["ylen"])
    for b, seqs in enumerate(out):
        assert seqs[0]["hyp"] == g["hyp"][b, : g["hyp_len"][b]].tolist()
        assert abs(seqs[0]["score"] - g["score"][b]) < 1e-3


@pytest.mark.parametrize("name,seed,ov", [("conf_tiny_dec_only", 4, dict(use_conv_enc=False)),
                                          ("conf_tiny_beam3", 4, dict(beam_width=3, length_penalty=0.1))])
def test_conformer_tiny_variants_fp32(name, seed, ov):
    g = load_golden(name)
    args, state, feats, sizes = conf_tiny_case(seed=seed, **ov)
    model = build(args, state, "fp32", capture=True)
    out = decode(model, args, feats, sizes)
    assert maxerr(model._engine.fetch("att_out"), g["att_out"]) < LOGIT_TOL
    for b, seqs in enumerate(out):
        assert seqs[0]["hyp"] == g["hyp"][b, : g["hyp_len"][b]].tolist()
        if "beam_hyp" in g:
            for j, s in enumerate(seqs):
                assert s["hyp"] == g["beam_hyp"][b, j, : g["beam_len"][b, j]].tolist(), (b, j)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_conformer_shipped_shape(prec, capsys):
    g = load_golden("conf_small")
    args, state, feats, sizes = conf_small_case()
    model = build(args, state, prec, capture=True)
    out = decode(model, args, feats, sizes)
    eng = model._engine
    best = eng.fetch("best_paths")
    clear = g["margin"] > 1e-4
    flips = int((best[clear] != g["best_paths"][clear]).sum())
    ctc_err = maxerr(eng.fetch("ctc_out")[:, ::5, ::13], g["ctc_sample"])
    exact = sum(seqs[0]["hyp"] == g["hyp"][b, : g["hyp_len"][b]].tolist() for b, seqs in enumerate(out))
    with capsys.disabled():
        print(f"\n[conformer {prec}] conf_small: frames {int(clear.sum())}, flips {flips}, ctc_logit_err {ctc_err:.3g}, hyp_exact {exact}/{len(out)}")
    if prec == "fp32":
        assert flips == 0 and ctc_err < LOGIT_TOL
        np.testing.assert_array_equal(eng.fetch("ylen"), g["ylen"])
        assert maxerr(eng.fetch("dec_h")[:, ::3, ::8], g["dec_sample"]) < 5e-4
        assert exact == len(out)
    else:
        assert flips <= 0.1 * clear.sum() and ctc_err < 0.2
