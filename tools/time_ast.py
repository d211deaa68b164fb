"""Timing of the autoregressive (AST) path at BASELINE configs[3] shape: same 12-layer encoder, 6 decoder layers, beam 10,
ctc_beam 15, joint CTC/attention scoring; B utterances x 1000 frames.  Not the headline bench (bench.py measures
configs[1]); prints one JSON line with utterances/s, ms per decode step and where a step's wall time goes.

    python tools/time_ast.py [--batch 32] [--frames 1000] [--precision bf16] [--ctc-weight 0.3] [--ratio 0.3]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cassnat_asr_public_amd import synth  # noqa: E402
from cassnat_asr_public_amd.models.transformer import make_model  # noqa: E402


class Vocab:
    word2index = {"blank": 0, "sos": 1, "eos": 2, "unk": 3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--ctc-weight", type=float, default=0.3)
    ap.add_argument("--ratio", type=float, default=0.3)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--streams", type=int, default=1,
                    help="decode pipelines (engine handle + HIP stream + host thread each) working on independent batches: a decode "
                         "step occupies a handful of CUs, so throughput - not latency - scales with pipelines")
    a = ap.parse_args()
    args = synth.make_args_ast("config4", ctc_weight=a.ctc_weight, max_decode_ratio=a.ratio)
    args.hip_precision = a.precision
    args.hip_max_batch = a.batch
    args.hip_max_frames = a.frames
    state = synth.make_state(args, seed=0, gain=2.0)
    feats, _ = synth.make_feats(a.batch, a.frames, args.input_size, seed=1234)
    model = make_model(args.input_size, args).cuda()
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(torch.from_numpy(state[k]))
    src = torch.from_numpy(feats).cuda()
    mask = (src[:, :, 0] != args.padding_idx).unsqueeze(1)
    times = []
    for r in range(a.reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        beams = model.beam_decode(src, mask, Vocab, args)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    best = min(times[1:])
    steps = max(len(b[0]["hyp"]) for b in beams) - 1
    out = {"workload": "BASELINE configs[3]: AST beam search, 12L enc / 6L dec, beam 10, ctc_beam 15",
           "batch": a.batch, "frames": a.frames, "precision": a.precision, "ctc_weight": a.ctc_weight,
           "decode_steps": steps, "sec_per_batch": round(best, 4), "utt_per_sec": round(a.batch / best, 2),
           "rtf": round(best / (a.batch * a.frames * 0.01), 6), "ms_per_decode_step": round(1e3 * best / max(steps, 1), 3),
           "all_runs_sec": [round(t, 4) for t in times]}
    if a.streams > 1:
        import threading

        models = [model]
        for _ in range(a.streams - 1):
            m2 = make_model(args.input_size, args).cuda()
            with torch.no_grad():
                for k, p in m2.named_parameters():
                    p.copy_(torch.from_numpy(state[k]))
            models.append(m2)

        def worker(i, n):
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for _ in range(n):
                    models[i].beam_decode(src, mask, Vocab, args)
                st.synchronize()

        for n in (1, a.reps):  # warm-up round (engine builds), then the timed one
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            th = [threading.Thread(target=worker, args=(i, n)) for i in range(a.streams)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
        out.update({"pipelines": a.streams, "pipelined_utt_per_sec": round(a.streams * a.reps * a.batch / el, 2),
                    "pipelined_sec_per_batch": round(el / (a.streams * a.reps), 4)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
