import os, sys, time, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cassnat_asr_public_amd.data import kaldi_io
from cassnat_asr_public_amd.data import speech_loader as SL
torch.set_num_threads(1)
rng = np.random.default_rng(5)
N = 1500
lengths = [int(x) for x in rng.integers(300, 1501, size=N)]
tmp = tempfile.mkdtemp()
def mats():
    for b, n in enumerate(lengths):
        yield f"utt{b:05d}", rng.standard_normal((n, 80)).astype(np.float32)
scp = os.path.join(tmp, "feats.scp")
kaldi_io.write_ark_scp(os.path.join(tmp, "feats.ark"), scp, mats())
stats = np.zeros((2, 81)); stats[0, :80] = 0.1 * 1000; stats[1, :80] = 1.2 * 1000; stats[0, 80] = 1000
kaldi_io.write_ark_scp(os.path.join(tmp, "cmvn.ark"), os.path.join(tmp, "cmvn.scp"), [("global", stats.astype(np.float64))])
class A: left_ctx = right_ctx = 0; skip_frame = 1; rank = 0
class V: word2index = {"unk": 3, "sos": 1, "eos": 2}
ds = SL.SpeechDataset(V, [{"name": "t", "scp_path": scp}], A)
ds._load_cmvn(open(os.path.join(tmp, "cmvn.scp")).read().split()[1])
for threads in (1,):
    for rep in range(2):
        dl = SL.SpeechDataLoader(ds, 32, padding_idx=0)
        t0 = time.perf_counter(); n = 0
        for b in dl: n += len(b[0])
        el = time.perf_counter() - t0
    print(f"threads {threads}: {n} utts in {el:.3f} s = {n/el:.0f} utt/s")
# pieces
t0=time.perf_counter()
for i in range(N): ds[i]
print("getitem only", time.perf_counter()-t0)
