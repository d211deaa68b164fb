"""Timing (and a full-size exercise) of `decode_type: ctc_only` (CTC prefix beam search) and `ctc_att` (its best hypothesis
forced through the NAT decoder) on the bench shape: config 2 model, B utterances x T frames.  Prints one JSON line.
    python tools/time_ctc_modes.py [--batch 32] [--frames 1000] [--precision bf16] [--ctc-beam 10]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cassnat_asr_public_amd import synth  # noqa: E402
from cassnat_asr_public_amd.models.cassnat import make_model  # noqa: E402
from cassnat_asr_public_amd.utils.beam_decode import ctc_beam_decode  # noqa: E402


class Vocab:
    word2index = {"blank": 0, "sos": 1, "eos": 2, "unk": 3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--ctc-beam", type=int, default=10)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    args = synth.make_args("config2", decode_type="ctc_att", sample_num=1, ctc_lm_weight=0, ctc_beam=a.ctc_beam, ctc_pruning=15, ctc_lp=0.0)
    args.hip_precision = a.precision
    args.hip_max_batch, args.hip_max_frames = a.batch, a.frames
    state = synth.make_state(args, seed=0, blank_bias=synth.BENCH_BLANK_BIAS)
    model = make_model(args.input_size, args).cuda()
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(torch.from_numpy(state[k]))
    fh, sh = synth.make_feats(a.batch, a.frames, args.input_size, seed=1234)
    src, sizes = torch.from_numpy(fh).cuda(), torch.from_numpy(sh).cuda()
    mask = (src[:, :, 0] != args.padding_idx).unsqueeze(1)
    t_ctc, t_att = [], []
    for _ in range(a.reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            top = ctc_beam_decode(model, src, mask, sizes, Vocab, args, None)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        with torch.no_grad():
            out, _ = model.beam_decode(src, mask, sizes, Vocab, args, None, top)
        torch.cuda.synchronize()
        t_ctc.append(t1 - t0)
        t_att.append(time.perf_counter() - t1)
    print(json.dumps({"workload": f"decode_type ctc_only / ctc_att: config 2 model, ctc_beam {a.ctc_beam}, pruning 15",
                      "batch": a.batch, "frames": a.frames, "precision": a.precision,
                      "ctc_only_sec_per_batch": round(min(t_ctc[1:]), 4), "ctc_att_extra_sec_per_batch": round(min(t_att[1:]), 4),
                      "ctc_only_utt_per_sec": round(a.batch / min(t_ctc[1:]), 1), "beams": len(top[0]),
                      "tokens_max": max(len(o[0]["hyp"]) for o in out) - 1}))


if __name__ == "__main__":
    main()
