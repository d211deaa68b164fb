"""Accuracy of the throughput engines on a batch large enough to resolve a percent of arg-max flips.

The config-5 fixture holds 600 frames (one flip = 0.17 %); this tool decodes B ragged utterances of up to T frames with the
fp32 engine (the parity engine: within 2e-6 of the reference, tests/test_gpu_pipeline.py) and with each engine named on the
command line, and prints the share of frames whose CTC arg-max differs, the largest log-posterior difference and WHERE the flips
sit by the fp32 engine's top-2 margin (utils/agreement.py).  The experiment switches (CASSNAT_NO_CONV2_F8, CASSNAT_NO_LINEAR_F8,
CASSNAT_FP8_LAYERS, ...) are read only by a -DCASSNAT_EXPERIMENTS build of the library (CASSNAT_HIP_LIB=ab/libcassnat_hip_exp.so,
`python -m cassnat_asr_public_amd.build --experiments`); `--hip_fp8_scope` is the product's setting for the same question.

    python tools/fp8_accuracy.py --engines "bf16,fp8,fp8[conv2+ffn:8],fp8[conv2]" --batch 48 --frames 1000
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cassnat_asr_public_amd import synth  # noqa: E402
from cassnat_asr_public_amd.models.cassnat import make_model  # noqa: E402
from cassnat_asr_public_amd.utils.agreement import flips_by_margin  # noqa: E402


class Vocab:
    word2index = {"blank": 0, "sos": 1, "eos": 2, "unk": 3}


def run(args, state, feats, sizes, precision):
    precision, _, scope = precision.partition("[")
    args.hip_precision = precision
    args.hip_fp8_scope = scope.rstrip("]") or "all"
    args.hip_capture = True
    model = make_model(args.input_size, args).cuda()
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(torch.from_numpy(state[k]))
        src = torch.from_numpy(feats)
        mask = (src[:, :, 0] != args.padding_idx).unsqueeze(1)
        out, _ = model.beam_decode(src.cuda(), mask.cuda(), torch.from_numpy(sizes).cuda(), Vocab, args)
    eng = model._engine
    return eng.fetch("best_paths"), eng.fetch("ctc_out"), [s[0]["hyp"] for s in out]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--engines", default="bf16,fp8")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--vocab", type=int, default=4234)
    ap.add_argument("--seed", type=int, default=5)
    a = ap.parse_args()
    args = synth.make_args("config2", vocab_size=a.vocab)
    state = synth.make_state(args, seed=a.seed, blank_bias=0.35)
    lens = synth.ragged_lengths(a.batch, a.frames, a.frames // 2, seed=9)
    feats, sizes = synth.make_feats(a.batch, a.frames, 80, lengths=lens, seed=77)
    ref_best, ref_ctc, ref_hyp = run(args, state, feats, sizes, "fp32")
    # frames of the padding carry no posterior of interest: count the utterances' own frames only
    tp = ref_best.shape[1]
    own = np.zeros(ref_best.shape, bool)
    for b, n in enumerate(lens):
        own[b, : min(tp, ((n - 1) // 2 - 1) // 2)] = True
    top2 = np.sort(np.partition(ref_ctc, -2, axis=-1)[..., -2:], axis=-1)
    margin = top2[..., 1] - top2[..., 0]  # the fp32 engine's top-2 margin per frame
    tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("CASSNAT_"))
    for prec in a.engines.split(","):
        best, ctc, hyp = run(args, state, feats, sizes, prec)
        flips = float((best != ref_best)[own].mean())
        err = float(np.abs(ctc.astype(np.float64) - ref_ctc.astype(np.float64))[own].max())
        rms = float(np.sqrt(np.mean((ctc.astype(np.float64) - ref_ctc.astype(np.float64))[own] ** 2)))
        same = sum(h == r for h, r in zip(hyp, ref_hyp))
        mg = flips_by_margin(best, ref_best, margin, own)
        print(f"{prec:18s} [{tag}] frames {int(own.sum())}: argmax flips {flips:.4f}, max |d log-posterior| {err:.4f}, rms {rms:.5f}, "
              f"hypotheses identical {same}/{len(hyp)}; largest flip margin {mg['max_flip_margin']:.4f} = "
              f"{mg['max_flip_margin'] / max(err, 1e-12):.2f} x max err (bound: 2); flips at margin >= 0.05: "
              f"{mg['flips_margin_ge_0.05']}/{mg['frames_margin_ge_0.05']}, >= 0.2: {mg['flips_margin_ge_0.2']}/{mg['frames_margin_ge_0.2']}; "
              f"[flips, frames] by margin {mg['flips_frames_by_margin']}", flush=True)


if __name__ == "__main__":
    main()
