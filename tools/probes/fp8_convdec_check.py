import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cassnat_asr_public_amd import synth
from cassnat_asr_public_amd.models.cassnat import make_model
class Vocab: word2index = {"blank": 0, "sos": 1, "eos": 2, "unk": 3}
# transformer encoder + conformer decoder side (the shipped decode YAML's use_conv_dec) under the fp8 engine
args = synth.make_args("conf_small", use_conv_enc=False)
state = synth.make_state(args, seed=3, blank_bias=0.35)
feats, sizes = synth.make_feats(4, 400, 80, lengths=[400, 333, 250, 180], seed=5)
outs = {}
for prec in ("bf16", "fp8"):
    args.hip_precision = prec
    m = make_model(80, args).cuda()
    with torch.no_grad():
        for k, p in m.named_parameters(): p.copy_(torch.from_numpy(state[k]))
        src = torch.from_numpy(feats)
        out, _ = m.beam_decode(src.cuda(), (src[:, :, 0] != 0).unsqueeze(1).cuda(), torch.from_numpy(sizes).cuda(), Vocab, args)
    outs[prec] = out
    print(prec, [len(s[0]["hyp"]) for s in out], [round(float(s[0]["score"]), 2) for s in out])
