"""Edge cases of the NAST hot path, HIP fp32 engine against the oracle on the same seeded inputs: smallest legal inputs,
frame counts around the subsampling boundaries, very ragged batches, utterances that emit no token at all, a batch
smaller / shorter than the workspace it runs in, and repeated calls on one engine.  Integer results (alignment, token
counts, hypotheses) must be identical; float stages within the tolerances of test_gpu_pipeline.py.
"""
import numpy as np
import pytest
import torch

from cassnat_asr_public_amd import hip, synth
from cassnat_asr_public_amd.models.cassnat import make_model
from oracle.cassnat_oracle import decode_nast

pytestmark = pytest.mark.gpu


class Vocab:
    word2index = {"blank": 0, "sos": 1, "eos": 2, "unk": 3}


def build(args, state, capture=True, prec="fp32"):
    args.hip_precision = prec
    args.hip_capture = capture
    model = make_model(args.input_size, args).cuda()
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(torch.from_numpy(state[k]))
    return model


def run_both(model, state, args, feats, sizes):
    src = torch.from_numpy(feats)
    mask = (src[:, :, 0] != args.padding_idx).unsqueeze(1)
    with torch.no_grad():
        out, _ = model.beam_decode(src.cuda(), mask.cuda(), torch.from_numpy(sizes).cuda(), Vocab, args)
    ref = decode_nast(state, feats, sizes, args, stages=True)
    return out, ref


def assert_same(model, out, ref, clear_margin=True):
    eng = model._engine
    ctc = ref["ctc_out"].numpy()
    top2 = np.sort(ctc, -1)[..., -2:]
    margin = top2[..., 1] - top2[..., 0]
    best = eng.fetch("best_paths")
    flips = best != ref["best_paths"]
    assert (margin[flips] < 1e-4).all()
    assert np.abs(eng.fetch("ctc_out") - ctc).max() < 1e-3
    if flips.any():  # a near-tie frame went the other way: everything downstream may legitimately differ
        return False
    np.testing.assert_array_equal(eng.fetch("aligned_seq_shift"), ref["aligned_seq_shift"])
    np.testing.assert_array_equal(eng.fetch("ylen"), ref["ylen"])
    assert int(eng.fetch("ymax")[0]) == int(ref["ymax"])
    att = ref["att_out"].numpy()
    assert np.abs(eng.fetch("att_out") - att).max() < 1e-3
    a2 = np.sort(att, -1)[..., -2:]
    if ((a2[..., 1] - a2[..., 0]) > 1e-4).all():
        for b, seqs in enumerate(out):
            assert seqs[0]["hyp"] == list(ref["hyps"][b]), b
            assert abs(seqs[0]["score"] - ref["scores"][b]) < 1e-3
    return True


@pytest.mark.parametrize("T", [1, 2, 3, 4, 5, 7, 8, 9, 62, 63, 64, 65])
def test_frame_counts_around_the_subsampling_boundaries(T):
    # T' = ((T - 1) // 2 + 1 - 1) // 2 + 1: one subsampled frame up to T = 4, two up to 8, ...
    args = synth.make_args("tiny")
    state = synth.make_state(args, seed=0, gain=2.0)
    feats, sizes = synth.make_feats(2, T, 80, lengths=[T, max(1, T - 2)], seed=5 + T)
    model = build(args, state)
    out, ref = run_both(model, state, args, feats, sizes)
    assert_same(model, out, ref)


def test_single_utterance_single_frame():
    args = synth.make_args("tiny")
    state = synth.make_state(args, seed=0, gain=2.0)
    feats, sizes = synth.make_feats(1, 1, 80, seed=3)
    model = build(args, state)
    out, ref = run_both(model, state, args, feats, sizes)
    assert_same(model, out, ref)
    assert len(out) == 1 and out[0][0]["hyp"][0] == 1


@pytest.mark.parametrize("lengths", [[61, 9, 5, 1], [61, 61, 2, 2], [61, 4, 4, 3]])
def test_very_ragged_batch(lengths):
    args = synth.make_args("tiny")
    state = synth.make_state(args, seed=0, gain=2.0)
    feats, sizes = synth.make_feats(len(lengths), 61, 80, lengths=lengths, seed=17)
    model = build(args, state)
    out, ref = run_both(model, state, args, feats, sizes)
    assert_same(model, out, ref)


def test_utterances_that_emit_no_token():
    # a huge blank bias makes every frame blank: the token count is 0, ylen = 1 (the EOS row only), one decoder row
    args = synth.make_args("tiny")
    state = synth.make_state(args, seed=0, gain=2.0, blank_bias=60.0)
    feats, sizes = synth.make_feats(3, 61, 80, lengths=[61, 50, 37], seed=11)
    model = build(args, state)
    out, ref = run_both(model, state, args, feats, sizes)
    assert (ref["best_paths"] == 0).all() and int(ref["ymax"]) == 1
    assert assert_same(model, out, ref)
    for seqs in out:
        assert len(seqs[0]["hyp"]) == 1 + min(2, int(ref["ymax"]))


def test_mixed_silent_and_speaking_utterances():
    # utterance 1 is all blank because its features are tiny (the bias decides); the others emit tokens: U comes from them
    args = synth.make_args("tiny")
    state = synth.make_state(args, seed=0, gain=2.0, blank_bias=2.5)
    feats, sizes = synth.make_feats(3, 61, 80, lengths=[61, 61, 40], seed=23)
    feats[1] *= 1e-3
    model = build(args, state)
    out, ref = run_both(model, state, args, feats, sizes)
    assert_same(model, out, ref)
    assert int(ref["ylen"].min()) >= 1


def test_small_call_inside_a_large_workspace_and_repeated_calls():
    # one engine sized for (8, 200) serves a (3, 61) call, then a (8, 200) call, then the first again: same answers
    args = synth.make_args("tiny")
    args.hip_max_batch, args.hip_max_frames = 8, 200
    state = synth.make_state(args, seed=0, gain=2.0)
    model = build(args, state)
    f1, s1 = synth.make_feats(3, 61, 80, lengths=[61, 50, 37], seed=11)
    f2, s2 = synth.make_feats(8, 200, 80, lengths=list(synth.ragged_lengths(8, 200, 20, seed=3)), seed=12)
    out1, ref1 = run_both(model, state, args, f1, s1)
    eng = model._engine
    assert_same(model, out1, ref1)
    out2, ref2 = run_both(model, state, args, f2, s2)
    assert model._engine is eng  # no rebuild
    assert_same(model, out2, ref2)
    out1b, _ = run_both(model, state, args, f1, s1)
    assert [s[0]["hyp"] for s in out1b] == [s[0]["hyp"] for s in out1]
    assert [s[0]["score"] for s in out1b] == [s[0]["score"] for s in out1]


def test_leading_padding_frames_are_masked_like_the_reference():
    # the reference derives the mask from feats[:, :, 0] != 0, wherever the zeros are: a zeroed frame in the middle of an
    # utterance is masked as a key, and size_ratio still counts it
    args = synth.make_args("tiny")
    state = synth.make_state(args, seed=0, gain=2.0)
    feats, sizes = synth.make_feats(2, 61, 80, lengths=[61, 45], seed=29)
    feats[0, 8:16] = 0.0
    feats[1, 0:4] = 0.0
    model = build(args, state)
    out, ref = run_both(model, state, args, feats, sizes)
    assert_same(model, out, ref)


@pytest.mark.parametrize("B,T,lengths", [(1, 1, [1]), (2, 5, [5, 2]), (3, 64, [64, 9, 1]), (5, 130, [130, 129, 77, 8, 4]),
                                          (2, 1530, [1530, 1100])])  # (T' = 383 > 256: keys no longer resident in one LDS tile set)
def test_bf16_fast_path_on_small_and_ragged_batches(B, T, lengths):
    """The bf16 engine at d_model 256 runs the fused kernels (row chain, LDS-DMA conv, fused generator); their row tiles are
    128 / 256 rows, so these batches exercise partial tiles, single rows and masked tails.  Checked against the fp32 engine
    on the same weights: encoder output within bf16 accumulation error, and - when both engines pick the same CTC path -
    identical token counts and a decoder output within the same error."""
    args = synth.make_args("config2", N_enc=2)
    state = synth.make_state(args, seed=4, blank_bias=0.3)
    feats, sizes = synth.make_feats(B, T, 80, lengths=lengths, seed=31)
    outs = {}
    for prec in ("fp32", "bf16"):
        args.hip_precision, args.hip_capture = prec, True
        model = make_model(args.input_size, args).cuda()
        with torch.no_grad():
            for k, p in model.named_parameters():
                p.copy_(torch.from_numpy(state[k]))
            src = torch.from_numpy(feats)
            out, _ = model.beam_decode(src.cuda(), (src[:, :, 0] != 0).unsqueeze(1).cuda(), torch.from_numpy(sizes).cuda(), Vocab, args)
        e = model._engine
        outs[prec] = dict(out=out, enc_h=e.fetch("enc_h"), best=e.fetch("best_paths"), ylen=e.fetch("ylen"), dec_h=e.fetch("dec_h"))
    a, b = outs["fp32"], outs["bf16"]
    assert np.isfinite(b["enc_h"]).all() and np.isfinite(b["dec_h"]).all()
    scale = np.abs(a["enc_h"]).max()
    assert np.abs(a["enc_h"] - b["enc_h"]).max() < 0.06 * scale
    if (a["best"] == b["best"]).all():
        np.testing.assert_array_equal(a["ylen"], b["ylen"])
        # rows u < ylen[b] only: the rest of dec_h is padding the reference never reads
        for i in range(B):
            n = int(a["ylen"][i])
            d = np.abs(a["dec_h"][i, :n] - b["dec_h"][i, :n]).max()
            assert d < 0.08 * np.abs(a["dec_h"][i, :n]).max(), (i, d)
    # the production call (no capture: blocked activation layout between chain launches, fused generator + arg-max)
    args.hip_precision, args.hip_capture = "bf16", False
    model = make_model(args.input_size, args).cuda()
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(torch.from_numpy(state[k]))
        src = torch.from_numpy(feats)
        prod, _ = model.beam_decode(src.cuda(), (src[:, :, 0] != 0).unsqueeze(1).cuda(), torch.from_numpy(sizes).cuda(), Vocab, args)
    Tp = ((T - 1) // 2 + 1 - 1) // 2 + 1
    for i, seqs in enumerate(prod):
        h = seqs[0]["hyp"]
        assert h[0] == 1 and 2 <= len(h) <= Tp + 2 and all(0 <= t < args.vocab_size for t in h) and np.isfinite(seqs[0]["score"])
    same = sum(p_[0]["hyp"] == c[0]["hyp"] for p_, c in zip(prod, b["out"]))
    assert same >= B - 1, (same, B)  # summation order differs between the fused and the captured generator: a near-tie may flip


def test_decode_pipelines_equal_the_plain_loop():
    """pipeline.DecodePipelines (what bench.py measures and CassNATTask.decode uses): 11 ragged batches of different shapes
    through 3 pipelines give, in order, exactly the hypotheses and scores of one beam_decode call per batch."""
    from cassnat_asr_public_amd.pipeline import DecodePipelines

    args = synth.make_args("tiny")
    args.hip_max_batch, args.hip_max_frames = 4, 90
    state = synth.make_state(args, seed=0, gain=2.0)
    model = build(args, state, capture=False)
    rng = np.random.default_rng(5)
    data = []
    for k in range(11):
        B, T = int(rng.integers(1, 5)), int(rng.integers(9, 91))
        lens = sorted((int(x) for x in rng.integers(1, T + 1, size=B)), reverse=True)
        lens[0] = T
        data.append(synth.make_feats(B, T, 80, lengths=lens, seed=100 + k))
    want = []
    for feats, sizes in data:
        src = torch.from_numpy(feats)
        with torch.no_grad():
            out, _ = model.beam_decode(src.cuda(), (src[:, :, 0] != 0).unsqueeze(1).cuda(), torch.from_numpy(sizes).cuda(), Vocab, args)
        want.append(([s[0]["hyp"] for s in out], [s[0]["score"] for s in out]))
    pipes = DecodePipelines(model, 3, 4, 90)
    got = list(pipes.decode(((torch.from_numpy(f), torch.from_numpy(s), k) for k, (f, s) in enumerate(data)), args, sos=1))
    pipes.close()
    assert [t for t, _, _ in got] == list(range(11))
    for (tag, hyps, scores), (wh, ws) in zip(got, want):
        assert hyps == wh, tag
        assert list(scores) == ws, tag
    # coalescing: equal-shaped neighbours share an engine pass; each batch keeps the hypotheses (incl. the batch-dependent
    # trailing token of the reference's greedy finish) and scores of a pass of its own
    same = []
    for k in range(9):  # runs of equal shapes: 3 x (3, 61), 1 x (2, 40), 4 x (4, 77), 1 x (1, 9)
        B, T = ([(3, 61)] * 3 + [(2, 40)] + [(4, 77)] * 4 + [(1, 9)])[k]
        lens = sorted((int(x) for x in rng.integers(1, T + 1, size=B)), reverse=True)
        lens[0] = T
        same.append(synth.make_feats(B, T, 80, lengths=lens, seed=300 + k))
    want2 = []
    for feats, sizes in same:
        src = torch.from_numpy(feats)
        with torch.no_grad():
            out, _ = model.beam_decode(src.cuda(), (src[:, :, 0] != 0).unsqueeze(1).cuda(), torch.from_numpy(sizes).cuda(), Vocab, args)
        want2.append(([s[0]["hyp"] for s in out], [s[0]["score"] for s in out]))
    pipes2 = DecodePipelines(model, 2, 4, 90, coalesce=2)
    got2 = list(pipes2.decode(((torch.from_numpy(f), torch.from_numpy(s), k) for k, (f, s) in enumerate(same)), args, sos=1))
    pipes2.close()
    assert [t for t, _, _ in got2] == list(range(9))
    for (tag, hyps, scores), (wh, ws) in zip(got2, want2):
        assert hyps == wh, tag
        assert list(scores) == ws, tag
    # bench.py's width: up to ten batches per pass (a list of 13 equal shapes on two pipelines goes as 7 + 6)
    wide, want3 = [], []
    for k in range(13):
        lens = sorted((int(x) for x in rng.integers(1, 78, size=4)), reverse=True)
        lens[0] = 77
        wide.append(synth.make_feats(4, 77, 80, lengths=lens, seed=500 + k))
    for feats, sizes in wide:
        src = torch.from_numpy(feats)
        with torch.no_grad():
            out, _ = model.beam_decode(src.cuda(), (src[:, :, 0] != 0).unsqueeze(1).cuda(), torch.from_numpy(sizes).cuda(), Vocab, args)
        want3.append(([s[0]["hyp"] for s in out], [s[0]["score"] for s in out]))
    pipes3 = DecodePipelines(model, 2, 4, 90, coalesce=10)
    got3 = list(pipes3.decode([(torch.from_numpy(f), torch.from_numpy(s), k) for k, (f, s) in enumerate(wide)], args, sos=1))
    pipes3.close()
    assert [t for t, _, _ in got3] == list(range(13))
    for (tag, hyps, scores), (wh, ws) in zip(got3, want3):
        assert hyps == wh, tag
        assert list(scores) == ws, tag


def test_lm_scoring_bf16_fast_path_against_the_fp32_engine():
    """TransformerLM scoring (ESA ranking): the bf16 engine runs attention + row chain per layer and the fused generator with
    a target gather; the fp32 engine runs the generic kernels and a full log-softmax.  Same tokens, ragged lengths, rows that
    do not fill a 128-row tile: scores of the positions the ranking reads (u < length) agree within bf16 error."""
    from cassnat_asr_public_amd.models.lm import make_model as make_lm

    lm_args = synth.make_args_lm("lm_small")
    lm_state = synth.make_state(lm_args, seed=9, gain=2.0)
    g = torch.Generator().manual_seed(3)
    N, U, ld = 37, 23, 26
    tok = torch.randint(4, lm_args.vocab_size, (N, ld), generator=g, dtype=torch.int32)
    tok[:, 0] = 1
    tgt = torch.randint(4, lm_args.vocab_size, (N, ld), generator=g, dtype=torch.int32)
    length = torch.randint(1, U + 1, (N,), generator=g, dtype=torch.int32)
    length[0], length[1] = U, 1
    got = {}
    for prec in ("fp32", "bf16"):
        lm_args.hip_precision = prec
        lm = make_lm(lm_args).cuda()
        with torch.no_grad():
            for k, p in lm.named_parameters():
                p.copy_(torch.from_numpy(lm_state[k]))
        got[prec] = lm.score_tokens(tok.cuda(), tgt.cuda(), length.cuda(), U, max_frames=256).cpu()  # (workspace: rows <= N * (frames / 4 + 1))
    mask = torch.arange(ld).view(1, ld) < length.view(N, 1).long()
    a, b = got["fp32"][mask], got["bf16"][mask]
    assert torch.isfinite(b).all()
    assert (a - b).abs().max().item() < 0.12 and (a - b).abs().mean().item() < 0.02


def test_bf16_engine_is_stateless_across_calls_of_different_shapes():
    """The bf16 engine reuses buffers between calls (the zero halo of the conv1 image is only rewritten when the batch shape
    changes; K|V of the decoder side come from the encoder's last launch): a batch decoded again after a batch of another
    shape, and again right after itself, gives bit-identical encoder outputs and hypotheses."""
    args = synth.make_args("config2", N_enc=2)
    args.hip_precision, args.hip_capture = "bf16", False
    args.hip_max_batch, args.hip_max_frames = 4, 200
    state = synth.make_state(args, seed=4, blank_bias=0.3)
    model = make_model(args.input_size, args).cuda()
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(torch.from_numpy(state[k]))
    a = synth.make_feats(2, 64, 80, lengths=[64, 41], seed=1)
    b = synth.make_feats(4, 200, 80, lengths=[200, 150, 90, 33], seed=2)
    c = synth.make_feats(3, 64, 80, lengths=[64, 64, 10], seed=3)  # same frame count as `a`, another batch size

    def run(fs):
        src = torch.from_numpy(fs[0])
        with torch.no_grad():
            out, _ = model.beam_decode(src.cuda(), (src[:, :, 0] != 0).unsqueeze(1).cuda(), torch.from_numpy(fs[1]).cuda(), Vocab, args)
        return [s[0]["hyp"] for s in out], [s[0]["score"] for s in out], model._engine.fetch("enc_h_live").copy()

    first = run(a)
    for other in (b, c, a):
        got = run(other)
        if other is a:
            assert got[0] == first[0] and got[1] == first[1] and np.array_equal(got[2], first[2])
    again = run(a)
    assert again[0] == first[0] and again[1] == first[1] and np.array_equal(again[2], first[2])


def test_randomised_shapes_and_options_against_the_oracle():
    """30 random cases on the tiny model, fp32 engine vs the oracle: batch 1-5, 1-140 frames, ragged lengths (down to one
    frame), trigger dilation, src_trigger, use_unimask, blank bias from 'every frame a token' to 'almost none', one engine
    reused throughout.  Integer results identical whenever no near-tie frame flipped; log-posteriors within 1e-3 always."""
    rng = np.random.default_rng(20260930)
    base = synth.make_args("tiny")
    base.hip_max_batch, base.hip_max_frames = 5, 140
    models = {}
    flipped = 0
    for case in range(30):
        B, T = int(rng.integers(1, 6)), int(rng.integers(1, 141))
        lens = sorted((int(v) for v in rng.integers(1, T + 1, size=B)), reverse=True)
        lens[0] = T
        ov = dict(left_trigger=int(rng.integers(0, 2)), right_trigger=int(rng.integers(0, 2)), src_trigger=bool(rng.integers(0, 2)),
                  use_unimask=bool(rng.integers(0, 2)))
        bias = float(rng.choice([0.0, 1.0, 3.0, 8.0]))
        args = synth.make_args("tiny", **ov)
        args.hip_max_batch, args.hip_max_frames = 5, 140
        args.hip_precision, args.hip_capture = "fp32", True  # (the capture switch is read per call)
        state = synth.make_state(args, seed=0, gain=2.0, blank_bias=bias)
        if bias not in models:  # one engine per weight set, reused across shapes and options
            models[bias] = build(args, state)
        model = models[bias]
        feats, sizes = synth.make_feats(B, T, 80, lengths=lens, seed=1000 + case)
        out, ref = run_both(model, state, args, feats, sizes)
        flipped += not assert_same(model, out, ref)
    assert flipped <= 3  # near-tie frames are rare


def test_fetch_refuses_captures_of_an_earlier_call():
    from cassnat_asr_public_amd import hip as _hip

    args = synth.make_args("tiny")
    state = synth.make_state(args, seed=0, gain=2.0)
    model = build(args, state, capture=True)
    feats, sizes = synth.make_feats(2, 40, 80, lengths=[40, 31], seed=2)
    run_both(model, state, args, feats, sizes)
    assert model._engine.fetch("ctc_out").shape[0] == 2
    args.hip_capture = False
    run_both(model, state, args, feats[:1], sizes[:1])
    with pytest.raises(_hip.HipError, match="earlier call"):
        model._engine.fetch("ctc_out")
    assert model._engine.fetch("best_paths").shape[0] == 1  # live buffers are always served


def test_randomised_shapes_bf16_fast_path_against_the_fp32_engine():
    """16 random batch shapes (1-6 utterances, 1-330 frames, ragged) through ONE bf16 engine (fused kernels: tiles of 128 / 256
    rows, so M = B * T' lands on and around tile boundaries) and ONE fp32 engine on the same weights: encoder output within bf16
    error for every shape, decoder output too whenever both pick the same CTC path."""
    args = synth.make_args("config2", N_enc=2)
    args.hip_max_batch, args.hip_max_frames = 6, 330
    state = synth.make_state(args, seed=4, blank_bias=0.3)
    eng = {}
    for prec in ("fp32", "bf16"):
        a = synth.make_args("config2", N_enc=2)
        a.hip_max_batch, a.hip_max_frames, a.hip_precision, a.hip_capture = 6, 330, prec, True
        m = make_model(a.input_size, a).cuda()
        with torch.no_grad():
            for k, p in m.named_parameters():
                p.copy_(torch.from_numpy(state[k]))
        eng[prec] = (m, a)
    rng = np.random.default_rng(77)
    same_path = 0
    for case in range(16):
        B = int(rng.integers(1, 7))
        T = int(rng.choice([int(rng.integers(1, 331)), 4 * int(rng.choice([32, 64, 128 // B + 1])) - int(rng.integers(0, 4))]))
        T = max(1, min(T, 330))
        lens = sorted((int(v) for v in rng.integers(1, T + 1, size=B)), reverse=True)
        lens[0] = T
        feats, sizes = synth.make_feats(B, T, 80, lengths=lens, seed=500 + case)
        got = {}
        for prec, (m, a) in eng.items():
            src = torch.from_numpy(feats)
            with torch.no_grad():
                m.beam_decode(src.cuda(), (src[:, :, 0] != 0).unsqueeze(1).cuda(), torch.from_numpy(sizes).cuda(), Vocab, a)
            e = m._engine
            got[prec] = dict(enc_h=e.fetch("enc_h"), best=e.fetch("best_paths"), ylen=e.fetch("ylen"), dec_h=e.fetch("dec_h"))
        x, y = got["fp32"], got["bf16"]
        assert np.isfinite(y["enc_h"]).all() and np.isfinite(y["dec_h"]).all(), (case, B, T)
        assert np.abs(x["enc_h"] - y["enc_h"]).max() < 0.06 * np.abs(x["enc_h"]).max(), (case, B, T)
        if (x["best"] == y["best"]).all():
            same_path += 1
            np.testing.assert_array_equal(x["ylen"], y["ylen"])
            for i in range(B):
                n = int(x["ylen"][i])
                assert np.abs(x["dec_h"][i, :n] - y["dec_h"][i, :n]).max() < 0.08 * np.abs(x["dec_h"][i, :n]).max(), (case, B, T, i)
    assert same_path >= 4


@pytest.mark.parametrize("prec", ["bf16", "bf16x3", "fp32", "fp16"])
def test_merged_pass_equals_separate_passes_when_rows_change_workgroups(prec):
    """The config-2 model at a size where a merged engine pass puts an utterance's rows into other workgroups than a pass of
    its own does (400 frames -> 100 rows per utterance, 128-row workgroups, three ragged batches of 4): hypotheses AND scores
    are bitwise those of the separate passes.  (A per-workgroup rotation of the FFN / vocabulary tile order made the fp32
    accumulation order depend on the workgroup: identical tokens, scores differing from the sixth digit on.)"""
    from cassnat_asr_public_amd.pipeline import DecodePipelines

    args = synth.make_args("config2")
    args.hip_max_batch, args.hip_max_frames = 4, 400
    state = synth.make_state(args, seed=0, blank_bias=synth.BENCH_BLANK_BIAS)
    model = build(args, state, capture=False, prec=prec)  # (bf16: row chain / generator kernels; bf16x3: FFN, projection, generator, conv2 kernels)
    rng = np.random.default_rng(21)
    data = []
    for k in range(3):
        lens = sorted((int(x) for x in rng.integers(150, 401, size=4)), reverse=True)
        lens[0] = 400
        data.append(synth.make_feats(4, 400, 80, lengths=lens, seed=700 + k))
    want = []
    for feats, sizes in data:
        src = torch.from_numpy(feats)
        with torch.no_grad():
            out, _ = model.beam_decode(src.cuda(), (src[:, :, 0] != 0).unsqueeze(1).cuda(), torch.from_numpy(sizes).cuda(), Vocab, args)
        want.append(([s[0]["hyp"] for s in out], [s[0]["score"] for s in out]))
    pipes = DecodePipelines(model, 1, 4, 400, coalesce=3)
    got = list(pipes.decode([(torch.from_numpy(f), torch.from_numpy(s), k) for k, (f, s) in enumerate(data)], args, sos=1))
    pipes.close()
    for (tag, hyps, scores), (wh, ws) in zip(got, want):
        assert hyps == wh, tag
        assert list(scores) == ws, tag


# ------------------------------------------------------------------------------- merged passes of different frame counts
def _separate(model, args, data):
    want = []
    for feats, sizes in data:
        src = torch.from_numpy(feats)
        with torch.no_grad():
            out, _ = model.beam_decode(src.cuda(), (src[:, :, 0] != 0).unsqueeze(1).cuda(), torch.from_numpy(sizes).cuda(), Vocab, args)
        want.append(([s[0]["hyp"] for s in out], [s[0]["score"] for s in out]))
    return want


def _ragged_batches(rng, shapes, seed0, holes=False):
    data = []
    for k, (B, T) in enumerate(shapes):
        lens = sorted((int(x) for x in rng.integers(1, T + 1, size=B)), reverse=True)
        lens[0] = T  # collate pads a batch to ITS longest utterance (speech_loader.py:327-356)
        f, s = synth.make_feats(B, T, 80, lengths=lens, seed=seed0 + k)
        if holes and T > 24:  # zeroed frames inside an utterance: masked keys in the middle, possibly a trigger row without any key
            f[0, 8:21] = 0.0
        data.append((f, s))
    return data


@pytest.mark.parametrize("prec", ["fp32", "bf16x3", "bf16", "fp16"])
def test_merged_pass_of_different_frame_counts_equals_separate_passes(prec):
    """cn_decode_nast_merged: batches of DIFFERENT frame counts (every residue of the two stride-2 subsamplings, different
    utterance counts, very short utterances, zeroed frames inside an utterance) through ONE engine pass give, per batch, exactly
    the hypotheses and scores of a pass of their own - the reference collates every batch to its own longest utterance
    (speech_loader.py:327-356) and decodes batch after batch (cassnat_task.py:317-356).  Also with the decoder side launched on a
    predicted row count (no mid-pass sync), including a prediction that falls short (the pass is decoded again)."""
    from cassnat_asr_public_amd.pipeline import DecodePipelines

    args = synth.make_args("tiny")
    args.hip_max_batch, args.hip_max_frames = 4, 96
    state = synth.make_state(args, seed=0, gain=2.0)
    model = build(args, state, capture=False, prec=prec)
    rng = np.random.default_rng(31)
    shapes = [(3, 61), (4, 64), (2, 62), (1, 63), (3, 65), (4, 57), (2, 41), (3, 44), (1, 43), (4, 42), (3, 9), (2, 12), (4, 10), (1, 1),
              (2, 5), (3, 96), (2, 90), (4, 77), (3, 80)]
    data = _ragged_batches(rng, shapes, 1200, holes=True)
    want = _separate(model, args, data)
    items = [(torch.from_numpy(f), torch.from_numpy(s), k) for k, (f, s) in enumerate(data)]
    for predict in (False, True):
        with DecodePipelines(model, 2, 4, 96, coalesce=5, ragged=0.5, predict_rows=predict) as pipes:
            got = list(pipes.decode(items, args, sos=1))
            assert pipes.stats["merged_ragged"] >= 3, pipes.stats
            if predict:
                assert pipes.stats["predicted"] >= 1, pipes.stats
        assert [t for t, _, _ in got] == list(range(len(shapes)))
        for (tag, hyps, scores), (wh, ws) in zip(got, want):
            assert hyps == wh, (predict, tag, shapes[tag])
            assert list(scores) == ws, (predict, tag, shapes[tag])
    # a prediction that is too small: detected when the pass has drained, the pass runs again exactly
    with DecodePipelines(model, 1, 4, 96, coalesce=5, ragged=0.5) as pipes:
        pipes._rows["ratio"] = 0.01
        pipes._learn = lambda ymax, T: None  # (keep the predictor wrong for the whole list)
        got = list(pipes.decode(items, args, sos=1))
        assert pipes.stats["missed"] >= 1, pipes.stats
    for (tag, hyps, scores), (wh, ws) in zip(got, want):
        assert hyps == wh and list(scores) == ws, tag


@pytest.mark.parametrize("prec", ["bf16", "bf16x3", "fp8", "fp16"])
def test_merged_ragged_pass_config2_size(prec):
    """The same equality on the config-2 model at sizes where the fast kernels run (row chain, LDS-DMA convolution, fused
    generator; 128-/256-row tiles cut across batches of 100-, 91- and 78-row utterances).  fp8: the e4m3 forms of the front end
    (conv1 on the matrix cores with per-utterance frame counts, conv2, linear_out) and of the chain's feed-forward products."""
    from cassnat_asr_public_amd.pipeline import DecodePipelines

    args = synth.make_args("config2")
    args.hip_max_batch, args.hip_max_frames = 4, 400
    state = synth.make_state(args, seed=0, blank_bias=synth.BENCH_BLANK_BIAS)
    model = build(args, state, capture=False, prec=prec)
    rng = np.random.default_rng(41)
    shapes = [(4, 400), (3, 363), (4, 310), (2, 397), (4, 333), (1, 301)]
    data = []
    for k, (B, T) in enumerate(shapes):
        lens = sorted((int(x) for x in rng.integers(120, T + 1, size=B)), reverse=True)
        lens[0] = T
        data.append(synth.make_feats(B, T, 80, lengths=lens, seed=1500 + k))
    want = _separate(model, args, data)
    items = [(torch.from_numpy(f), torch.from_numpy(s), k) for k, (f, s) in enumerate(data)]
    with DecodePipelines(model, 1, 4, 400, coalesce=6, ragged=0.7) as pipes:
        got = list(pipes.decode(items, args, sos=1, plan=[3, 3]))
        assert pipes.stats["merged_ragged"] == 2 and pipes.stats["passes"] == 2, pipes.stats
    for (tag, hyps, scores), (wh, ws) in zip(got, want):
        assert hyps == wh, tag
        assert list(scores) == ws, tag


def test_workspace_is_an_area_and_every_buffer_is_checked():
    """The engine's workspace is max_batch x max_frames of AREA: a call with more, shorter utterances runs (and gives the plain
    call's results); a call that does not fit is refused with the name of the buffer that is too small, before anything is
    launched - capacities used to be implicit per buffer (round 2's GPU fault: decoder rows written past a logits buffer)."""
    from cassnat_asr_public_amd import hip

    args = synth.make_args("tiny")
    args.hip_max_batch, args.hip_max_frames = 2, 64
    state = synth.make_state(args, seed=0, gain=2.0)
    model = build(args, state, capture=False)
    f, s = synth.make_feats(5, 20, 80, lengths=[20, 17, 11, 5, 1], seed=77)
    eng = model.engine(2, 64)
    assert (eng.cfg.max_batch, eng.cfg.max_frames) == (2, 64)
    hyp, hyp_len, score = model.decode_device(torch.from_numpy(f), torch.from_numpy(s), args, engine=eng)  # 5 x 20 fits 2 x 64
    args2 = synth.make_args("tiny")
    args2.hip_max_batch, args2.hip_max_frames = 8, 64
    big = build(args2, state, capture=False)
    h2, l2, s2 = big.decode_device(torch.from_numpy(f), torch.from_numpy(s), args2)
    assert torch.equal(hyp_len, l2) and torch.equal(score, s2)
    assert all(torch.equal(hyp[b, : int(l2[b])], h2[b, : int(l2[b])]) for b in range(5))
    f3, s3 = synth.make_feats(3, 64, 80, seed=78)
    with pytest.raises(hip.HipError, match="workspace buffer '"):
        model.decode_device(torch.from_numpy(f3), torch.from_numpy(s3), args, engine=eng)  # 3 x 64 does not
    f4, s4 = synth.make_feats(40, 4, 80, seed=79)
    with pytest.raises(hip.HipError, match="workspace buffer '"):
        model.decode_device(torch.from_numpy(f4), torch.from_numpy(s4), args, engine=eng)  # 40 one-row utterances do not either


def test_bench_size_merged_pass_and_esa_50_in_every_precision(capsys):
    """Once, at the benchmark's size (32 x 1000 frames, config-2 model): a ten-batch merged engine pass - the bench's launch
    width, 80,000 encoder rows - and ESA with sample_num 50 (the shipped decode YAML) in bf16, bf16x3 and fp32.  No golden exists
    at this size; the checks are the ones that found round 2's bugs: every buffer's capacity is checked on the host (ws_check),
    the merged pass equals a plain call bitwise, and the two parity-grade engines agree with each other."""
    from cassnat_asr_public_amd.models.lm import make_model as make_lm
    from cassnat_asr_public_amd.pipeline import DecodePipelines

    B, T = 32, 1000
    hyps = {}
    for prec in ("bf16", "bf16x3", "fp32"):
        args = synth.make_args("config2")
        args.hip_max_batch, args.hip_max_frames = B, T
        state = synth.make_state(args, seed=0, blank_bias=synth.BENCH_BLANK_BIAS)
        model = build(args, state, capture=False, prec=prec)
        rng = np.random.default_rng(3)
        data = []
        for k in range(10):
            lens = sorted((int(x) for x in rng.integers(400, T + 1, size=B)), reverse=True)
            lens[0] = T
            data.append(synth.make_feats(B, T, 80, lengths=lens, seed=2000 + k))
        want = _separate(model, args, data[:2])
        model._engine.close()
        model._engine = None
        with DecodePipelines(model, 1, B, T, coalesce=10) as pipes:
            got = list(pipes.decode([(torch.from_numpy(f), torch.from_numpy(s), k) for k, (f, s) in enumerate(data)], args, sos=1, plan=[10]))
            assert pipes.stats["passes"] == 1 and pipes.stats["batches"] == 10, pipes.stats
        for (tag, h, sc), (wh, ws) in zip(got[:2], want):
            assert h == wh and list(sc) == ws, (prec, tag)
        hyps[prec] = [h for _, h, _ in got]
        # ESA, sample_num 50 in groups of 16: B x 16 x U decoder rows per group against every buffer's capacity
        a2 = synth.make_args("config2", sample_num=50, rank_model="lm", threshold=0.9)
        a2.hip_precision, a2.hip_max_batch, a2.hip_max_frames = prec, B, T
        lm_args = synth.make_args_lm("lm_small", vocab_size=a2.vocab_size)
        lm_args.hip_precision = prec
        lm_state = synth.make_state(lm_args, seed=9, gain=2.0)
        m2, lm = build(a2, state, capture=False, prec=prec), make_lm(lm_args).cuda()
        with torch.no_grad():
            for k, p in lm.named_parameters():
                p.copy_(torch.from_numpy(lm_state[k]))
        src, sizes = torch.from_numpy(data[0][0]).cuda(), torch.from_numpy(data[0][1]).cuda()
        torch.manual_seed(3)
        with torch.no_grad():
            out, _ = m2.beam_decode(src, (src[:, :, 0] != 0).unsqueeze(1), sizes, Vocab, a2, lm)
        torch.cuda.synchronize()
        assert len(out) == B and all(np.isfinite(o[0]["score"]) for o in out)
        hyps[prec + "/esa"] = [o[0]["hyp"] for o in out]
        for x in (m2, lm, model):
            if getattr(x, "_engine", None) is not None:
                x._engine.close()
                x._engine = None
    same = sum(a == b for x, y in zip(hyps["fp32"], hyps["bf16x3"]) for a, b in zip(x, y))
    same_esa = sum(a == b for a, b in zip(hyps["fp32/esa"], hyps["bf16x3/esa"]))
    same_bf = sum(a == b for x, y in zip(hyps["fp32"], hyps["bf16"]) for a, b in zip(x, y))
    with capsys.disabled():
        print(f"\n[bench size] merged ten-batch pass: bf16x3 == fp32 on {same}/320 hypotheses, bf16 == fp32 on {same_bf}/320; "
              f"ESA 50: bf16x3 == fp32 on {same_esa}/32")
    assert same >= 316 and same_esa >= 30


def test_merged_pass_whose_conv_image_exceeds_4_gib():
    """13 batches of 32 x 1000 frames in one engine pass: the bordered conv1 image is 416 x 502 x 42 x 512 B = 4.5 GB.  The LDS-DMA
    convolution used absolute 32-bit lane offsets into it (they wrapped silently past 4 GiB: wrong hypotheses, no fault); they are
    relative to the tile's first row now.  The last batch - the one whose rows lie past 4 GiB - must decode exactly as it does alone."""
    from cassnat_asr_public_amd.pipeline import DecodePipelines

    B, T, n = 32, 1000, 13
    args = synth.make_args("config2")
    args.hip_max_batch, args.hip_max_frames = B, T
    state = synth.make_state(args, seed=0, blank_bias=synth.BENCH_BLANK_BIAS)
    model = build(args, state, capture=False, prec="bf16")
    first = synth.make_feats(B, T, 80, seed=2100)
    rng = np.random.default_rng(9)
    lens = sorted((int(x) for x in rng.integers(500, T + 1, size=B)), reverse=True)
    lens[0] = T
    last = synth.make_feats(B, T, 80, lengths=lens, seed=2101)
    want = _separate(model, args, [first, last])
    model._engine.close()
    model._engine = None
    f0, s0 = torch.from_numpy(first[0]).cuda(), torch.from_numpy(first[1]).cuda()
    items = [(f0, s0, k) for k in range(n - 1)] + [(torch.from_numpy(last[0]).cuda(), torch.from_numpy(last[1]).cuda(), n - 1)]
    with DecodePipelines(model, 1, B, T, coalesce=n) as pipes:
        assert pipes.fits(n * B, T)
        got = list(pipes.decode(items, args, sos=1, plan=[n]))
        assert pipes.stats["passes"] == 1
    assert (n * B) * (T // 2 + 2) * 42 * 512 > 1 << 32
    for k in (0, n - 2):
        assert got[k][1] == want[0][0] and list(got[k][2]) == want[0][1], k
    assert got[n - 1][1] == want[1][0] and list(got[n - 1][2]) == want[1][1]


def test_a_ticket_expires_instead_of_returning_another_passes_counts():
    """A ticket names the pass by its sequence number on the handle; its counts live in one of four page-locked words.  After four
    further decode calls on the handle (plain cn_decode_nast calls included) the word belongs to another pass: cn_decode_ticket
    fails loudly instead of handing out that pass's row counts (a caller with more than three outstanding passes
    could accept a short prediction as covered and return truncated hypotheses)."""
    args = synth.make_args("tiny")
    args.hip_max_batch, args.hip_max_frames = 4, 96
    state = synth.make_state(args, seed=0, gain=2.0)
    model = build(args, state, capture=False, prec="fp32")
    f, s = synth.make_feats(3, 61, 80, lengths=[61, 50, 37], seed=11)
    feats, ratio = torch.from_numpy(f).cuda(), torch.from_numpy(s).cuda()
    eng = model.engine(3, 61)
    tickets = []
    for _ in range(4):
        *_, t = model.decode_device(feats, ratio, args, 1, engine=eng, u_hint=40, want_ticket=True)
        tickets.append(t)
    torch.cuda.synchronize()
    assert tickets == list(range(tickets[0], tickets[0] + 4))
    counts = [eng.ticket(t) for t in tickets]  # four outstanding passes: all still valid, all the same counts
    assert len(set(counts)) == 1 and counts[0][1] == 17 and 1 <= counts[0][0] <= 17  # (rows used: min(u_hint, T' + 1), T' = 16)
    model.decode_device(feats, ratio, args, 1, engine=eng)  # a plain call takes the oldest ticket's word over
    torch.cuda.synchronize()
    with pytest.raises(hip.HipError, match="expired"):
        eng.ticket(tickets[0])
    assert eng.ticket(tickets[1]) == counts[1]


@pytest.mark.parametrize("prec", ["fp32", "bf16", "fp16"])
def test_merged_pass_without_trigger_rows_equals_separate_passes(prec):
    """use_trigger = False (cassnat.py:469-473) through merged engine passes and the packed reader's batch objects: per batch the
    hypotheses and scores of a pass of its own (the extractor's mask is the utterance's own frame range, the row counts carry no
    EOS row).  Batches without a CTC token are left out: the reference itself raises on them (a scatter at index -1, :477)."""
    from cassnat_asr_public_amd.pipeline import DecodePipelines, PackedBatch

    args = synth.make_args("tiny", use_trigger=False)
    args.hip_max_batch, args.hip_max_frames = 4, 96
    state = synth.make_state(args, seed=0, gain=2.0)
    model = build(args, state, capture=False, prec=prec)
    rng = np.random.default_rng(33)
    shapes = [(3, 61), (4, 64), (2, 62), (3, 65), (4, 57), (2, 41), (3, 44), (4, 42), (3, 96), (2, 90), (4, 77), (3, 80)]
    data = []
    for k, (B, T) in enumerate(shapes):
        lens = sorted((int(x) for x in rng.integers(T // 2, T + 1, size=B)), reverse=True)
        lens[0] = T
        data.append(synth.make_feats(B, T, 80, lengths=lens, seed=2100 + k))
    want = _separate(model, args, data)
    assert all(len(h) >= 2 for hyps, _ in want for h in hyps)
    items = [(torch.from_numpy(f), torch.from_numpy(s), k) for k, (f, s) in enumerate(data)]
    # the same batches as the packed reader hands them over: unpadded rows per utterance
    packed = []
    for k, (f, s) in enumerate(data):
        T = f.shape[1]
        pb = PackedBatch([np.ascontiguousarray(f[b, : int(round(float(s[b]) * T))]) for b in range(f.shape[0])])
        assert pb.shape == f.shape and torch.equal(pb.ratios(), torch.from_numpy(s))
        packed.append((pb, pb.ratios(), k))
    for batches in (items, packed):
        with DecodePipelines(model, 2, 4, 96, coalesce=5, ragged=0.5, copy_threads=2) as pipes:
            got = list(pipes.decode(batches, args, sos=1))
            assert pipes.stats["merged_ragged"] >= 2, pipes.stats
        for (tag, hyps, scores), (wh, ws) in zip(got, want):
            assert hyps == wh and list(scores) == ws, (tag, shapes[tag])


def test_fp16_engine_fails_loudly_outside_the_half_range():
    """The fp16 engine's MFMA operands have a range (+-65504) and do not saturate.  What drives magnitudes from outside is the scale
    of the features, so every pass checks them against the bound under which neither subsampling convolution's output can pass
    65504 / 2 (from the convolutions' weight row sums; csrc/model.hip op16_feat_limit): `beam_decode` and the decode pipelines
    raise instead of returning hypotheses of such a pass; inside the bound the engine decodes, and the same features decode on the
    bf16 engine (range of fp32)."""
    from cassnat_asr_public_amd.pipeline import DecodePipelines

    args = synth.make_args("tiny")
    state = synth.make_state(args, seed=0, gain=2.0)
    feats, sizes = synth.make_feats(3, 61, 80, lengths=[61, 50, 37], seed=11)
    ratio = torch.from_numpy(sizes).cuda()
    model = build(args, state, capture=False, prec="fp16")
    src0 = torch.from_numpy(feats).cuda()
    mask = (src0[:, :, 0] != args.padding_idx).unsqueeze(1)
    ok, _ = model.beam_decode(src0, mask, ratio, Vocab, args)  # in range: decodes
    assert all(np.isfinite(o[0]["score"]) for o in ok)
    fault, limit = hip.C.c_int32(), hip.C.c_float()
    eng = model._engine
    hip.check(eng.L.cn_take_range_fault(eng.handle, hip.C.byref(fault), hip.C.byref(limit)), "cn_take_range_fault", eng.L)
    # the bound follows from the weights: |conv1| <= fmax A1 + B1, |conv2| <= |conv1| A2 + B2, both <= 32752
    w1, b1 = state["src_embed.conv.0.weight"], state["src_embed.conv.0.bias"]
    w2, b2 = state["src_embed.conv.2.weight"], state["src_embed.conv.2.bias"]
    A1, A2 = np.abs(w1).reshape(w1.shape[0], -1).sum(1).max(), np.abs(w2).reshape(w2.shape[0], -1).sum(1).max()
    c1 = min(32752.0, (32752.0 - np.abs(b2).max()) / A2)
    want = (c1 - np.abs(b1).max()) / A1
    assert fault.value == 0 and abs(limit.value - want) < 1e-3 * want, (limit.value, want)
    scale = 1.05 * limit.value / np.abs(feats).max()
    just_in = torch.from_numpy((feats * (0.95 / 1.05 * scale)).astype(np.float32)).cuda()
    model.beam_decode(just_in, mask, ratio, Vocab, args)
    big = torch.from_numpy((feats * scale).astype(np.float32)).cuda()
    with pytest.raises(hip.HipError, match="half-precision range"):
        model.beam_decode(big, mask, ratio, Vocab, args)
    model.beam_decode(src0, mask, ratio, Vocab, args)  # (the flag was taken: the next call is judged on its own features)
    pipes = DecodePipelines(model, 1, 3, 61, coalesce=2)
    try:
        with pytest.raises(hip.HipError, match="half-precision range"):
            list(pipes.decode([(src0, ratio, 0), (big, ratio, 1)], args))
    finally:
        pipes.close()
    wide = build(args, state, capture=False, prec="bf16")
    out, _ = wide.beam_decode(big, mask, ratio, Vocab, args)
    assert all(np.isfinite(o[0]["score"]) for o in out)
    # a batch that does not start on a 16-byte boundary (a view into a larger buffer): the guard reads it all the same
    base = torch.empty(src0.numel() + 1, device="cuda")
    odd = base[1:].view_as(src0)
    assert odd.data_ptr() % 16 == 4
    odd.copy_(src0)
    got, _ = model.beam_decode(odd, mask, ratio, Vocab, args)
    assert [o[0]["hyp"] for o in got] == [o[0]["hyp"] for o in ok] and [o[0]["score"] for o in got] == [o[0]["score"] for o in ok]
    odd.copy_(big)
    with pytest.raises(hip.HipError, match="half-precision range"):
        model.beam_decode(odd, mask, ratio, Vocab, args)
    # an engine that received its weights as a blob (a non-zero rank) carries the bound too: it is a word of the blob
    eng2 = hip.Engine(args, precision="fp16", max_batch=3, max_frames=61)
    eng2.load_state(state, torch.zeros(5000, args.d_model))
    hip.check(eng2.L.cn_take_range_fault(eng2.handle, hip.C.byref(fault), hip.C.byref(limit)), "cn_take_range_fault", eng2.L)
    assert abs(limit.value - want) < 1e-3 * want


def test_bf16x3_engine_guards_the_range_of_its_mixed_convolution():
    """The split-bf16 engine's second convolution runs the mixed arithmetic of csrc/conv2.hip (half-precision hi x hi + e4m3 cross
    terms at fixed scales): its e4m3 operands hold conv1 outputs up to 448.  Features that could push conv1 beyond that would not
    break the decode - the cross terms saturate - but take the engine below its tolerance, silently; the feature-range guard
    (cn_take_range_fault, the bound from conv1's weight row sums) makes `beam_decode` raise instead."""
    args = synth.make_args("config2")
    state = synth.make_state(args, seed=0, blank_bias=synth.BENCH_BLANK_BIAS)
    feats, sizes = synth.make_feats(2, 120, 80, lengths=[120, 93], seed=5)
    model = build(args, state, capture=False, prec="bf16x3")
    src = torch.from_numpy(feats).cuda()
    ratio = torch.from_numpy(sizes).cuda()
    mask = (src[:, :, 0] != args.padding_idx).unsqueeze(1)
    model.beam_decode(src, mask, ratio, Vocab, args)
    eng = model._engine
    fault, limit = hip.C.c_int32(), hip.C.c_float()
    hip.check(eng.L.cn_take_range_fault(eng.handle, hip.C.byref(fault), hip.C.byref(limit)), "cn_take_range_fault", eng.L)
    w1, b1 = state["src_embed.conv.0.weight"], state["src_embed.conv.0.bias"]
    want = (448.0 - np.abs(b1).max()) / np.abs(w1).reshape(w1.shape[0], -1).sum(1).max()
    assert fault.value == 0 and abs(limit.value - want) < 1e-3 * want, (limit.value, want)
    big = torch.from_numpy((feats * (1.05 * limit.value / np.abs(feats).max())).astype(np.float32)).cuda()
    with pytest.raises(hip.HipError, match="mixed-arithmetic convolution"):
        model.beam_decode(big, mask, ratio, Vocab, args)
    model.beam_decode(src, mask, ratio, Vocab, args)
