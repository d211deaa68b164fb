"""Timing of the waveform -> log-mel front-end (csrc/fbank.hip): B utterances x S seconds of 16 kHz audio resident in HBM.
    python tools/time_fbank.py [--batch 32] [--seconds 10]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cassnat_asr_public_amd import hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--seconds", type=float, default=10.015)
a = ap.parse_args()
L = hip.lib()
o = hip.CnFbankOpts()
L.cn_fbank_default_opts(o)
ns = int(a.seconds * 16000)
T = L.cn_fbank_num_frames(o, ns)
wave = (3000 * torch.randn(a.batch, ns, device="cuda")).contiguous()
nsd = torch.full((a.batch,), ns, dtype=torch.int32, device="cuda")
out = torch.empty(a.batch, T, 80, device="cuda")
run = lambda: hip.check(L.cn_fbank(o, hip._ptr(wave), hip._ptr(nsd), a.batch, ns, None, None, hip._ptr(out), T, 0.0, hip.current_stream()))
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
bytes_alg = wave.numel() * 4 + out.numel() * 4
print(json.dumps({"kernel": "fbank_kernel", "batch": a.batch, "seconds_each": a.seconds, "frames": T, "ms": round(ms, 4),
                  "audio_seconds_per_second": round(a.batch * a.seconds / (ms * 1e-3)), "GBps_algorithmic": round(bytes_alg / ms / 1e6, 1)}))
