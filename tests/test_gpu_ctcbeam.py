"""decode_type ctc_only / ctc_att and the at_baseline / n-gram ESA rankers on the device (second half).

  * kernels on the reference's known-answer vectors (tests/golden/ctc_kat.npz: outputs of the reference's own ctc_beam_decode
    and viterbi_align): hypotheses and the aligned path bit-exact, float64 scores to 1e-9;
  * end to end through the drop-in API with the fp32 engine: `utils.beam_decode.ctc_beam_decode` and
    `CassNAT.beam_decode(..., ctc_top_seqs)` against the reference's outputs (tests/golden/ctcbeam_*.npz) and, as the exact
    check of the integer / float64 part, against the oracle run on the engine's OWN log-posteriors;
  * ESA ranked by the autoregressive baseline against the reference (esa_at_tiny.npz); the n-gram ranker's wiring.
"""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import ast_tiny_case, ctcbeam_case, load_golden, tiny_case
from cassnat_asr_public_amd import hip, synth
from cassnat_asr_public_amd.models.cassnat import make_model
from cassnat_asr_public_amd.utils.beam_decode import ctc_beam_decode
from oracle import cassnat_oracle as orc

pytestmark = pytest.mark.gpu


class Vocab:
    word2index = {"blank": 0, "sos": 1, "eos": 2, "unk": 3}
    index2word = {i: f"▁w{i}" for i in range(64)}


def p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def run_beam_kernel(ctc, ratio, W, P, lp):
    B, Tp, V = ctc.shape
    logp, rd = torch.from_numpy(ctc).float().cuda(), torch.from_numpy(ratio).float().cuda()
    hyp = torch.zeros(B, W, Tp + 1, dtype=torch.int32, device="cuda")
    hlen, nb = torch.zeros(B, W, dtype=torch.int32, device="cuda"), torch.zeros(B, dtype=torch.int32, device="cuda")
    sc, pb, pnb = (torch.zeros(B, W, dtype=torch.float64, device="cuda") for _ in range(3))
    hip.check(hip.lib().cn_op_ctc_prefix_beam(p(logp), p(rd), B, Tp, V, W, P, lp, 0, p(hyp), Tp + 1, p(hlen), p(sc), p(pb), p(pnb),
                                              p(nb), hip.current_stream()))
    torch.cuda.synchronize()
    return [t.cpu().numpy() for t in (hyp, hlen, sc, pb, pnb, nb)]


def assert_beams(got, g, prefix=""):
    hyp, hlen, sc, pb, pnb, nb = got
    np.testing.assert_array_equal(nb, g[prefix + "beam_n"])
    for b in range(hyp.shape[0]):
        n = int(nb[b])
        np.testing.assert_array_equal(hlen[b, :n], g[prefix + "beam_len"][b, :n])
        for j in range(n):
            assert hyp[b, j, : hlen[b, j]].tolist() == g[prefix + "beam_hyp"][b, j, : hlen[b, j]].tolist(), (b, j)
        for a, name in ((sc, "beam_score"), (pb, "beam_p_blk"), (pnb, "beam_p_nblk")):
            np.testing.assert_allclose(a[b, :n], g[prefix + name][b, :n], rtol=0, atol=1e-9)


def test_prefix_beam_and_viterbi_kernels_on_the_reference_vectors():
    g = load_golden("ctc_kat")
    for tag in "abc":
        W, P, lp = g[tag + "_cfg"]
        assert_beams(run_beam_kernel(g["ctc"], g["ratio"], int(W), int(P), float(lp)), g, tag + "_")
    B, Tp, V = g["ctc"].shape
    logp, km = torch.from_numpy(g["ctc"]).cuda(), torch.from_numpy(g["mask"].astype(np.uint8)).cuda()
    rd = torch.from_numpy(g["ratio"]).float().cuda()
    ys, yl = torch.from_numpy(g["ys"].astype(np.int32)).cuda(), torch.from_numpy(g["ylens"].astype(np.int32)).cuda()
    path = torch.full((B, Tp), -1, dtype=torch.int32, device="cuda")
    hip.check(hip.lib().cn_op_ctc_viterbi(p(logp), p(km), p(rd), p(ys), p(yl), B, Tp, V, ys.shape[1], int(g["ylens"].max()), 0, p(path),
                                          hip.current_stream()))
    torch.cuda.synchronize()
    # collapse + shift of the per-frame labels (src/models/cassnat.py:344-352) = the reference's aligned_seq_shift
    lab = path.cpu().numpy().astype(np.int64)
    prev = np.zeros_like(lab)
    prev[:, 1:] = lab[:, :-1]
    col = np.where(lab == prev, 0, lab)
    shift = np.zeros_like(lab)
    shift[:, 1:] = col[:, :-1]
    np.testing.assert_array_equal(shift, g["viterbi_shift"])


def test_prefix_beam_kernel_random_vs_oracle():
    """Larger random cases (beam 32 x pruning 32 candidates, many frames, blank-heavy rows that get skipped) against the oracle."""
    rng = np.random.default_rng(3)
    for (B, Tp, V, W, P, lp) in [(4, 97, 50, 32, 32, 0.1), (2, 260, 300, 20, 30, 0.0), (3, 40, 9, 7, 0, 0.0)]:
        logits = rng.standard_normal((B, Tp, V)).astype(np.float32) * 2
        logits[:, ::3, 0] += 6.0  # blank probability > 0.95 on a third of the frames
        ctc = torch.log_softmax(torch.from_numpy(logits), -1).numpy()
        ratio = np.linspace(1.0, 0.6, B).astype(np.float32)
        got = run_beam_kernel(ctc, ratio, W, P, lp)
        ref = orc.ctc_prefix_beam(ctc, orc.src_size_frames(ratio, Tp), W, P, lp)
        hyp, hlen, sc, pb, pnb, nb = got
        for b in range(B):
            assert int(nb[b]) == len(ref[b])
            for j, s in enumerate(ref[b]):
                assert hyp[b, j, : hlen[b, j]].tolist() == s["hyp"], (b, j)
            np.testing.assert_allclose(sc[b, : len(ref[b])], [s["score_ctc"] for s in ref[b]], rtol=0, atol=1e-8)


def build(args, state, prec="fp32", capture=False):
    args.hip_precision, args.hip_capture = prec, capture
    model = make_model(args.input_size, args).cuda()
    with torch.no_grad():
        for k, q in model.named_parameters():
            q.copy_(torch.from_numpy(state[k]))
    return model


@pytest.mark.parametrize("name", ["ctcbeam_tiny", "ctcbeam_config2"])
@pytest.mark.parametrize("prec", ["fp32", "bf16x3"])
def test_ctc_only_and_ctc_att_against_the_reference(name, prec, capsys):
    g = load_golden(name)
    args, state, feats, sizes = ctcbeam_case(name)
    model = build(args, state, prec, capture=True)
    src = torch.from_numpy(feats).cuda()
    mask = (src[:, :, 0] != 0).unsqueeze(1)
    ratio = torch.from_numpy(sizes).cuda()
    with torch.no_grad():
        top = ctc_beam_decode(model, src, mask, ratio, Vocab, args, None)                       # decode_type ctc_only
        out, _ = model.beam_decode(src, mask, ratio, Vocab, args, None, top)                    # decode_type ctc_att
    eng = model._engine
    ctc_dev = eng.fetch("ctc_out")
    # (1) exact check of the integer / float64 part: the oracle on the engine's own log-posteriors
    ref = orc.decode_nast_ctc(state, feats, sizes, args, ctc_out_override=ctc_dev)
    for b, seqs in enumerate(top):
        assert [s["hyp"] for s in seqs] == [s["hyp"] for s in ref["beams"][b]]
        np.testing.assert_allclose([s["score_ctc"] for s in seqs], [s["score_ctc"] for s in ref["beams"][b]], rtol=0, atol=1e-8)
    np.testing.assert_array_equal(eng.fetch("aligned_seq_shift"), ref["aligned_seq_shift"])
    # (2) against the reference's own run: the best hypothesis, the alignment and the attention-decoder output must be the
    # reference's; lower beams may swap where two float64 scores differ by less than the engines' 1e-5 logit error
    same_beams = sum(s["hyp"] == g["beam_hyp"][b, j, : g["beam_len"][b, j]].tolist() for b, seqs in enumerate(top) for j, s in enumerate(seqs))
    total = sum(len(seqs) for seqs in top)
    with capsys.disabled():
        print(f"\n[ctc beam {prec}] {name}: {same_beams}/{total} beam entries identical to the reference's")
    for b, seqs in enumerate(top):
        assert seqs[0]["hyp"] == g["beam_hyp"][b, 0, : g["beam_len"][b, 0]].tolist()
        assert abs(seqs[0]["score_ctc"] - g["beam_score"][b, 0]) < 1e-2
    np.testing.assert_array_equal(eng.fetch("aligned_seq_shift"), g["aligned_seq_shift"])
    for b, seqs in enumerate(out):
        assert seqs[0]["hyp"] == g["hyp"][b, : g["hyp_len"][b]].tolist()
        assert abs(seqs[0]["score"] - g["score"][b]) < 2e-2
    assert same_beams >= total - 2


def test_esa_ranked_by_the_autoregressive_baseline():
    from cassnat_asr_public_amd.models.transformer import make_model as make_ast

    g = load_golden("esa_at_tiny")
    args = synth.make_args("tiny", sample_num=4, threshold=0.9, rank_model="at_baseline")
    aa = synth.make_args_ast("tiny_ast")
    state, ast_state = synth.make_state(args, seed=0, gain=2.0), synth.make_state(aa, seed=3, gain=2.0)
    feats, sizes = synth.make_feats(3, 61, 80, lengths=[61, 50, 37], seed=11)
    args.esa_select = g["select"]
    aa.hip_precision = "fp32"
    model = build(args, state, "fp32")
    ast = make_ast(aa.input_size, aa).cuda()
    with torch.no_grad():
        for k, q in ast.named_parameters():
            q.copy_(torch.from_numpy(ast_state[k]))
    src = torch.from_numpy(feats).cuda()
    with torch.no_grad():
        out, _ = model.beam_decode(src, (src[:, :, 0] != 0).unsqueeze(1), torch.from_numpy(sizes).cuda(), Vocab, args, ast)
    for b, seqs in enumerate(out):
        h, ref = seqs[0]["hyp"], g["hyp"][b, : g["hyp_len"][b]].tolist()
        n = len(h) - 1 if len(h) == len(ref) and h[-1] == 0 and ref[-1] != 0 else len(h)  # the masked-row tie token
        assert len(h) == len(ref) and h[:n] == ref[:n]
    np.testing.assert_allclose([s[0]["score"] for s in out], g["score"], rtol=1e-5, atol=2e-3)


def test_esa_ngram_ranker_wiring():
    """rank_model 'n-gram' (kenlm in the reference; any object with score(str) here): with all draws zero every sample is
    the best path, so the pick cannot matter and the result is the greedy hypothesis; the stub sees B * sample_num sentences."""
    args, state, feats, sizes = tiny_case(sample_num=4, threshold=0.9, rank_model="n-gram")
    args.esa_select = np.zeros((3 * 4, 16, 1), np.uint8)
    model = build(args, state, "fp32")

    class Stub:
        seen = []

        def score(self, text):
            self.seen.append(text)
            return -float(len(text.split()))

    lm = Stub()
    src = torch.from_numpy(feats).cuda()
    with torch.no_grad():
        out, _ = model.beam_decode(src, (src[:, :, 0] != 0).unsqueeze(1), torch.from_numpy(sizes).cuda(), Vocab, args, lm)
    plain = orc.decode_nast(state, feats, sizes, synth.make_args("tiny"))
    assert len(lm.seen) == 12 and all(" " in t or t == "" or t.startswith("w") for t in lm.seen)
    for b, seqs in enumerate(out):
        n = min(len(seqs[0]["hyp"]), len(plain["hyps"][b])) - 1
        assert seqs[0]["hyp"][:n] == plain["hyps"][b][:n]


@pytest.mark.parametrize("decode_type", ["ctc_only", "ctc_att"])
def test_decode_asr_cli_with_ctc_decode_types(tmp_path, decode_type):
    """decode_asr.py --task cassnat with `decode_type: ctc_only / ctc_att` in the YAML (src/tasks/cassnat_task.py:335-341):
    the result file is the oracle's hypotheses as text."""
    from test_gpu_multirank import _write_case
    from cassnat_asr_public_amd.bin import decode_asr

    args, state, feats, sizes = ctcbeam_case("ctcbeam_tiny")
    lengths = [61, 50, 37]
    scp, ckpt, cfg = _write_case(tmp_path, args, state, feats, lengths,
                                 extra_conf=dict(decode_type=decode_type, sample_num=1, ctc_beam=5, ctc_pruning=8, ctc_lp=0.2, ctc_lm_weight=0))
    result = str(tmp_path / "result.txt")
    rc = decode_asr.main(["--task", "cassnat", "--test_config", cfg, "--data_path", scp, "--resume_model", ckpt, "--result_file", result,
                          "--batch_size", "3", "--hip_precision", "fp32", "--load_data_workers", "0"])
    assert rc == 0
    args.decode_type = decode_type
    ref = orc.decode_nast_ctc(state, feats, sizes, args)
    hyps = [b[0]["hyp"] for b in ref["beams"]] if decode_type == "ctc_only" else ref["hyps"]
    index2word = {i + 4: f"w{i}" for i in range(args.vocab_size - 4)}
    index2word[3] = "unk"  # (a CTC hypothesis may carry the unk id: data.vocab.Vocab maps ids 0..3 to blank/sos/eos/unk)
    expect = [f"spk-utt{b:02d} " + " ".join(orc.hyp_to_text(h, index2word)) for b, h in enumerate(hyps)]
    assert open(result).read().splitlines() == expect
