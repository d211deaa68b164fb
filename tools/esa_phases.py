"""Per-phase wall times (ms, synchronised between phases) of ESA decoding at sample_num 50 on the bench shape: draws to the
device, encoder + CTC generator, the four decoder-side group passes, LM input, LM scoring, scores to the host.  Eight
repetitions with the same draws: shows where a repetition's time goes and that the occasional 50-70 ms outlier is not tied
to a phase (it lands wherever the GPU was idle before).
    python tools/esa_phases.py"""
import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
from cassnat_asr_public_amd import synth, hip
from cassnat_asr_public_amd.models.cassnat import make_model
from cassnat_asr_public_amd.models.lm import make_model as make_lm
args = synth.make_args("config2", sample_num=50, rank_model="lm", threshold=0.9)
args.hip_precision = "bf16"; args.hip_max_batch, args.hip_max_frames = 32, 1000
lm_args = synth.make_args_lm("lm_small", vocab_size=args.vocab_size); lm_args.hip_precision = "bf16"
state = synth.make_state(args, seed=0, blank_bias=synth.BENCH_BLANK_BIAS); lm_state = synth.make_state(lm_args, seed=9, gain=2.0)
model = make_model(80, args).cuda(); lm = make_lm(lm_args).cuda()
with torch.no_grad():
    for k, p in model.named_parameters(): p.copy_(torch.from_numpy(state[k]))
    for k, p in lm.named_parameters(): p.copy_(torch.from_numpy(lm_state[k]))
fh, sh = synth.make_feats(32, 1000, 80, seed=1234)
src, sizes = torch.from_numpy(fh).cuda(), torch.from_numpy(sh).cuda()
B, T, S, Tp = 32, 1000, 50, 250
dev = src.device
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for rep in range(8):
    torch.manual_seed(0)
    t = [sync()]
    eng = model.engine(B, T, esa_group=16)
    opts = hip.Engine.make_opts(args); opts.sos = 1
    select = torch.randint(0, 2, (B * S, Tp, 1))
    select = select.reshape(B, S, Tp).to(torch.uint8).transpose(0, 1).contiguous(); select[0] = 0
    select = select.to(dev); t.append(sync())
    eng.esa_begin(src, opts); t.append(sync())
    stride = Tp + 2
    tok = torch.zeros(S, B, stride, dtype=torch.int32, device=dev); val = torch.zeros(S, B, stride, dtype=torch.float32, device=dev)
    ylen = torch.zeros(S, B, dtype=torch.int32, device=dev)
    U = 0
    for g0 in range(0, S, 16):
        g1 = min(S, g0 + 16)
        U = max(U, eng.esa_sample(select[g0:g1], args.threshold, sizes, opts, tok[g0:g1], val[g0:g1], ylen[g0:g1]))
        t.append(sync())
    tokf, ylf = tok.reshape(S * B, stride), ylen.reshape(S * B)
    lm_in = torch.cat([torch.full((S * B, 1), 1, dtype=torch.int32, device=dev), tokf[:, : stride - 1]], 1).contiguous(); t.append(sync())
    sc = lm.score_tokens(lm_in, tokf.contiguous(), ylf.contiguous(), U, max_frames=1000); t.append(sync())
    x = sc.reshape(S, B, stride)[:, :, :U].transpose(0, 1).cpu(); t.append(sync())
    print(rep, U, " ".join(f"{(b - a) * 1e3:.1f}" for a, b in zip(t, t[1:])), flush=True)
