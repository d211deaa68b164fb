"""Which host-side setting makes the sporadic 60-90 ms stalls of repeated ESA decodes go away?  Runs tools/time_esa.py's loop
under a few settings in ONE process (same engine): python gc on/off, torch intra-op threads, a short sleep between repetitions.
    python tools/esa_stall_probe.py"""
import gc
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


args = synth.make_args("config2", sample_num=50, rank_model="lm", threshold=0.9)
args.hip_precision = "bf16"
args.hip_max_batch, args.hip_max_frames = 32, 1000
lm_args = synth.make_args_lm("lm_small", vocab_size=args.vocab_size)
lm_args.hip_precision = "bf16"
state = synth.make_state(args, seed=0, blank_bias=synth.BENCH_BLANK_BIAS)
lm_state = synth.make_state(lm_args, seed=9, gain=2.0)
model, lm = make_model(80, args).cuda(), make_lm(lm_args).cuda()
with torch.no_grad():
    for k, p in model.named_parameters():
        p.copy_(torch.from_numpy(state[k]))
    for k, p in lm.named_parameters():
        p.copy_(torch.from_numpy(lm_state[k]))
fh, sh = synth.make_feats(32, 1000, 80, seed=1234)
src, sizes = torch.from_numpy(fh).cuda(), torch.from_numpy(sh).cuda()
mask = (src[:, :, 0] != 0).unsqueeze(1)
args.esa_select = torch.randint(0, 2, (32 * 50, 250, 1))  # fixed draws: no torch.randint inside the loop


def run(n, pause=0.0):
    out = []
    for _ in range(n):
        if pause:
            time.sleep(pause)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.beam_decode(src, mask, sizes, Vocab, args, lm)
        torch.cuda.synchronize()
        out.append(round((time.perf_counter() - t0) * 1e3, 1))
    return out


run(2)
print("default            ", run(12), flush=True)
gc.disable()
print("gc off             ", run(12), flush=True)
gc.enable()
torch.set_num_threads(1)
print("1 torch thread     ", run(12), flush=True)
print("sleep 50 ms between", run(12, 0.05), flush=True)
del args.esa_select
print("with torch.randint ", run(12), flush=True)
