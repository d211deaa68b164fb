"""Full-size exercise (B = 32 x 1000 frames, config 2 model) of options the tests only run at fixture sizes: NAT beam search
(beam_width > 1), ragged lengths, the capture mode, in several precisions.  Prints one line per case; any failure raises.
    python tools/exercise_full_size.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cassnat_asr_public_amd import synth  # noqa: E402
from cassnat_asr_public_amd.models.cassnat import make_model  # noqa: E402


class Vocab:
    word2index = {"blank": 0, "sos": 1, "eos": 2, "unk": 3}


def run(prec, lengths=None, capture=False, **over):
    args = synth.make_args("config2", **over)
    args.hip_precision, args.hip_capture = prec, capture
    args.hip_max_batch, args.hip_max_frames = 32, 1000
    state = synth.make_state(args, seed=0, blank_bias=synth.BENCH_BLANK_BIAS)
    model = make_model(args.input_size, args).cuda()
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(torch.from_numpy(state[k]))
    fh, sh = synth.make_feats(32, 1000, args.input_size, lengths=lengths, seed=1234)
    src, sizes = torch.from_numpy(fh).cuda(), torch.from_numpy(sh).cuda()
    mask = (src[:, :, 0] != args.padding_idx).unsqueeze(1)
    best = None
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            out, _ = model.beam_decode(src, mask, sizes, Vocab, args, None)
        torch.cuda.synchronize()
        best = time.perf_counter() - t0
    assert len(out) == 32 and all(np.isfinite(o[0]["score"]) for o in out)
    return best, out


def main():
    rng = np.random.default_rng(7)
    ragged = sorted((int(x) for x in rng.integers(400, 1001, size=32)), reverse=True)
    ragged[0] = 1000
    cases = [("greedy, ragged lengths 400..1000", dict(lengths=ragged)), ("beam_width 4", dict(beam_width=4)),
             ("beam_width 4, ragged", dict(beam_width=4, lengths=ragged)), ("capture mode", dict(capture=True))]
    ref = {}
    for name, kw in cases:
        for prec in ("bf16", "bf16x3", "fp32"):
            t, out = run(prec, **kw)
            hyps = [o[0]["hyp"] for o in out]
            if prec == "fp32":
                ref[name] = hyps
            print("%-36s %-7s %7.2f ms   beams %d   max tokens %d" % (name, prec, t * 1e3, len(out[0]), max(len(h) for h in hyps) - 1), flush=True)
        # the two parity-grade engines agree hypothesis for hypothesis
        t, out = run("bf16x3", **kw)
        same = sum(o[0]["hyp"] == r for o, r in zip(out, ref[name]))
        print("%-36s bf16x3 vs fp32: %d / 32 best hypotheses identical" % (name, same), flush=True)
        assert same >= 31, (name, same)


def pipelines_ragged():
    """Twelve ragged batches of the same padded shape through 2 pipelines x up to 10 batches per pass against the plain loop."""
    from cassnat_asr_public_amd.pipeline import DecodePipelines

    args = synth.make_args("config2")
    args.hip_precision = "bf16"
    args.hip_max_batch, args.hip_max_frames = 32, 1000
    state = synth.make_state(args, seed=0, blank_bias=synth.BENCH_BLANK_BIAS)
    model = make_model(args.input_size, args).cuda()
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(torch.from_numpy(state[k]))
    rng = np.random.default_rng(11)
    data = []
    for k in range(12):
        lens = sorted((int(x) for x in rng.integers(300, 1001, size=32)), reverse=True)
        lens[0] = 1000
        data.append(synth.make_feats(32, 1000, args.input_size, lengths=lens, seed=900 + k))
    want = []
    for fh, sh in data:
        src = torch.from_numpy(fh).cuda()
        with torch.no_grad():
            out, _ = model.beam_decode(src, (src[:, :, 0] != 0).unsqueeze(1), torch.from_numpy(sh).cuda(), Vocab, args, None)
        want.append(([o[0]["hyp"] for o in out], [o[0]["score"] for o in out]))
    with DecodePipelines(model, 2, 32, 1000, coalesce=10) as pipes:
        got = list(pipes.decode([(torch.from_numpy(f), torch.from_numpy(s), k) for k, (f, s) in enumerate(data)], args, sos=1))
    bad = 0
    for (tag, hyps, scores), (wh, ws) in zip(got, want):
        for b in range(32):
            if hyps[b] != wh[b] or scores[b] != ws[b]:
                bad += 1
                if bad <= 6:
                    d = [i for i in range(min(len(hyps[b]), len(wh[b]))) if hyps[b][i] != wh[b][i]]
                    print("batch %d utt %d: len %d vs %d, first token diffs at %s, score %.6f vs %.6f" % (tag, b, len(hyps[b]), len(wh[b]), d[:5], scores[b], ws[b]), flush=True)
    assert bad == 0, "%d utterances differ" % bad
    print("12 ragged batches of 32 x 1000 through 2 pipelines x 10 per pass: hypotheses and scores equal the plain loop's", flush=True)


if __name__ == "__main__":
    if "--pipelines-only" not in sys.argv:
        main()
    pipelines_ragged()
