"""Timing of ESA decoding (sample_num alignments per utterance + TransformerLM ranking: the shipped cassnat_decode.yaml's
mode) on the bench shape: config 2 model, B utterances x 1000 frames, LM preset lm_small.  Not the headline bench; prints
one JSON line.
    python tools/time_esa.py [--batch 32] [--frames 1000] [--samples 50] [--precision bf16]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cassnat_asr_public_amd import synth  # noqa: E402
from cassnat_asr_public_amd.models.cassnat import make_model  # noqa: E402
from cassnat_asr_public_amd.models.lm import make_model as make_lm  # noqa: E402


class Vocab:
    word2index = {"blank": 0, "sos": 1, "eos": 2, "unk": 3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--samples", type=int, default=50)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--same-seed", action="store_true", help="the same random draws in every repetition")
    ap.add_argument("--rank", default="lm", choices=["lm", "at_baseline"],
                    help="ranker: the TransformerLM (lm_small) or the autoregressive baseline (config 4 model, teacher-forced)")
    ap.add_argument("--group", type=int, default=0, help="args.hip_esa_group: samples per decoder-side pass (0: the package's default)")
    a = ap.parse_args()
    args = synth.make_args("config2", sample_num=a.samples, rank_model=a.rank, threshold=0.9)
    if a.group > 0:
        args.hip_esa_group = a.group
    args.hip_precision = a.precision
    args.hip_max_batch, args.hip_max_frames = a.batch, a.frames
    if a.rank == "lm":
        lm_args = synth.make_args_lm("lm_small", vocab_size=args.vocab_size)
    else:
        lm_args = synth.make_args_ast("config4", vocab_size=args.vocab_size)
    lm_args.hip_precision = a.precision
    state = synth.make_state(args, seed=0, blank_bias=synth.BENCH_BLANK_BIAS)
    lm_state = synth.make_state(lm_args, seed=9, gain=2.0)
    model = make_model(args.input_size, args).cuda()
    if a.rank == "lm":
        lm = make_lm(lm_args).cuda()
    else:
        from cassnat_asr_public_amd.models.transformer import make_model as make_ast

        lm = make_ast(lm_args.input_size, lm_args).cuda()
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(torch.from_numpy(state[k]))
        for k, p in lm.named_parameters():
            p.copy_(torch.from_numpy(lm_state[k]))
    fh, sh = synth.make_feats(a.batch, a.frames, args.input_size, seed=1234)
    src, sizes = torch.from_numpy(fh).cuda(), torch.from_numpy(sh).cuda()
    mask = (src[:, :, 0] != args.padding_idx).unsqueeze(1)
    times = []
    for r in range(a.reps + 1):
        torch.manual_seed(0 if a.same_seed else r)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, _ = model.beam_decode(src, mask, sizes, Vocab, args, lm)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    best = min(times[1:])
    print(json.dumps({"workload": f"ESA: config 2 model, sample_num {a.samples}, " + ("TransformerLM lm_small" if a.rank == "lm" else "autoregressive-baseline (config 4 model)") + " ranking",
                      "batch": a.batch, "frames": a.frames, "precision": a.precision, "esa_group": a.group or None, "sec_per_batch": round(best, 4),
                      "utt_per_sec": round(a.batch / best, 2), "rtf": round(best / (a.batch * a.frames * 0.01), 6),
                      "tokens_max": max(len(o[0]["hyp"]) for o in out) - 1, "all_runs_sec": [round(t, 4) for t in times]}))


if __name__ == "__main__":
    main()
