"""Timing of the conformer variant (use_conv_enc / use_conv_dec: relative-position attention, macaron FFNs, GLU + depthwise
convolution + GroupNorm modules) at the conf_small shape, greedy decoding, B utterances x 1000 frames.  Not the headline
bench; prints one JSON line.
    python tools/time_conformer.py [--batch 32] [--frames 1000] [--precision bf16]"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cassnat_asr_public_amd import synth  # noqa: E402
from cassnat_asr_public_amd.models.cassnat import make_model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--transformer-encoder", action="store_true",
                    help="use_conv_enc False: transformer encoder under the conformer decoder (the shipped cassnat_decode.yaml)")
    ap.add_argument("--streams", type=int, default=1, help="also time N decode pipelines (pipeline.DecodePipelines) over 40 batches")
    a = ap.parse_args()
    args = synth.make_args("conf_small", use_conv_enc=not a.transformer_encoder)
    args.hip_precision = a.precision
    args.hip_max_batch, args.hip_max_frames = a.batch, a.frames
    state = synth.make_state(args, seed=0, blank_bias=synth.BENCH_BLANK_BIAS)
    model = make_model(args.input_size, args).cuda()
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(torch.from_numpy(state[k]))
    fh, sh = synth.make_feats(a.batch, a.frames, args.input_size, seed=1234)
    src, sizes = torch.from_numpy(fh).cuda(), torch.from_numpy(sh).cuda()
    times = []
    for r in range(a.reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hyp, hyp_len, score = model.decode_device(src, sizes, args)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    best = min(times[1:])
    piped = None
    if a.streams > 1:
        from cassnat_asr_public_amd.pipeline import DecodePipelines

        pipes = DecodePipelines(model, a.streams, a.batch, a.frames)
        for n in (2 * a.streams, 40):  # warm-up, then the timed run
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in pipes.decode(((src, sizes, k) for k in range(n)), args, sos=1, as_lists=False):
                pass
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        piped = round(40 * a.batch / el, 1)
        pipes.close()
    print(json.dumps({"workload": "conformer CASS-NAT (conf_small: 12L %s encoder, 1+1+6 conformer decoder, d 256, V 1028), greedy" % ("transformer" if a.transformer_encoder else "conformer"),
                      "batch": a.batch, "frames": a.frames, "precision": a.precision, "sec_per_batch": round(best, 5),
                      "utt_per_sec": round(a.batch / best, 1), "rtf": round(best / (a.batch * a.frames * 0.01), 7),
                      "tokens_max": int(hyp_len.max()), "pipelines": a.streams, "pipelined_utt_per_sec": piped,
                      "all_runs_sec": [round(t, 5) for t in times]}))


if __name__ == "__main__":
    main()
