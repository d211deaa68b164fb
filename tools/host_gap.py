"""Where does a decode pipeline's host thread spend its time?  Replays bench.py's N-pipeline loop with host timers:
per batch the time inside the C call (launches + the token-count sync), in the Python glue around it (tensor allocation,
record packing) and in the main thread's record unpacking; prints means and the batches/s.
    python tools/host_gap.py [--streams 4] [--steps 60] [--switch 0.005]"""
import argparse
import os
import queue
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=4)
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--switch", type=float, default=0.005)
a = ap.parse_args()
sys.setswitchinterval(a.switch)

import torch

from cassnat_asr_public_amd import dist as cdist
from cassnat_asr_public_amd import hip, synth
from cassnat_asr_public_amd.models.cassnat import make_model

args = synth.make_args("config2")
args.hip_precision = "bf16"
B, T, F = 32, 1000, args.input_size
args.hip_max_batch, args.hip_max_frames = B, T
state = synth.make_state(args, seed=0, blank_bias=synth.BENCH_BLANK_BIAS)
models = []
for i in range(a.streams):
    m = make_model(F, args).cuda(0)
    with torch.no_grad():
        for k, p in m.named_parameters():
            p.copy_(torch.from_numpy(state[k]))
    m.build_engine(B, T, with_weights=True)
    models.append(m)
fh, sh = synth.make_feats(B, T, F, seed=1234)
feats, sizes = torch.from_numpy(fh).cuda(), torch.from_numpy(sh).cuda()
NS = a.streams
tim = {"c_call": 0.0, "glue": 0.0, "unpack": 0.0, "wait_q": 0.0}
lock = threading.Lock()


def run(n):
    done = [queue.Queue() for _ in range(NS)]

    def worker(i):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for k in range(i, n, NS):
                t0 = time.perf_counter()
                eng = models[i]._engine
                real = eng.decode
                tc = [0.0]

                def timed(*x):
                    t = time.perf_counter()
                    r = real(*x)
                    tc[0] += time.perf_counter() - t
                    return r

                eng.decode = timed
                hyp, hl, sc = models[i].decode_device(feats, sizes, args)
                eng.decode = real
                rec = cdist.pack_records(hyp, hl, sc)
                ev = torch.cuda.Event()
                ev.record(st)
                done[i].put((rec, ev))
                t1 = time.perf_counter()
                with lock:
                    tim["c_call"] += tc[0]
                    tim["glue"] += t1 - t0 - tc[0]
            st.synchronize()

    th = [threading.Thread(target=worker, args=(i,)) for i in range(NS)]
    for t in th:
        t.start()
    for k in range(n):
        t0 = time.perf_counter()
        rec, ev = done[k % NS].get()
        t1 = time.perf_counter()
        ev.wait(torch.cuda.current_stream())
        cdist.unpack_records(rec)
        tim["unpack"] += time.perf_counter() - t1
        tim["wait_q"] += t1 - t0
    for t in th:
        t.join()


run(2 * NS)
for k in tim:
    tim[k] = 0.0
torch.cuda.synchronize()
t0 = time.perf_counter()
run(a.steps)
torch.cuda.synchronize()
el = time.perf_counter() - t0
print(f"streams {NS} switch {a.switch}: {a.steps / el * B:.0f} utt/s, {el / a.steps * 1e3:.3f} ms/step; per batch (ms): "
      + ", ".join(f"{k} {v / a.steps * 1e3:.3f}" for k, v in tim.items()))
