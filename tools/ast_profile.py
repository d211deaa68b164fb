"""Where a batch of the autoregressive (AST, BASELINE configs[3]) decode spends its time: HIP-event time per kernel tag (the engine's
own profile hooks) against the wall time of the call - the difference is launch overhead and gaps between kernels.
    python tools/ast_profile.py [--precision bf16]      (GPU box)"""
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
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    args = synth.make_args_ast("config4", ctc_weight=0.3, max_decode_ratio=0.3)
    args.hip_precision, args.hip_max_batch, args.hip_max_frames = a.precision, a.batch, 1000
    state = synth.make_state(args, seed=0, gain=2.0)
    feats, _ = synth.make_feats(a.batch, 1000, args.input_size, seed=1234)
    model = make_model(args.input_size, args).cuda()
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(torch.from_numpy(state[k]))
    src = torch.from_numpy(feats).cuda()
    mask = (src[:, :, 0] != args.padding_idx).unsqueeze(1)
    for _ in range(2):
        model.beam_decode(src, mask, Vocab, args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.beam_decode(src, mask, Vocab, args)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    eng = model._engine
    eng.profile_begin()
    t0 = time.perf_counter()
    model.beam_decode(src, mask, Vocab, args)
    torch.cuda.synchronize()
    wall_prof = time.perf_counter() - t0
    prof = eng.profile_end()
    tot = sum(v["ms"] for v in prof.values())
    n = sum(v["count"] for v in prof.values())
    rows = sorted(prof.items(), key=lambda kv: -kv[1]["ms"])
    print(json.dumps({"precision": a.precision, "wall_ms": round(wall * 1e3, 2), "wall_ms_with_events": round(wall_prof * 1e3, 2),
                      "tagged_kernel_ms": round(tot, 2), "tagged_launches": n,
                      "by_tag": {k: {"count": v["count"], "ms": round(v["ms"], 3), "us_each": round(1e3 * v["ms"] / max(v["count"], 1), 2)} for k, v in rows}}))


if __name__ == "__main__":
    main()
