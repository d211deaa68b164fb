"""Helper of tests/test_host_cpu.py::test_eight_ranks_over_gloo_*: one rank of `decode_asr --task cassnat` on the CPU with the device
engine replaced by a stub (the host side is the product's own: CassNATTask, the snake deal, DecodePipelines, the result merge).

The stub "decodes" an utterance from its OWN frames only - token ids derived from the utterance id it finds in feature 0 and from
its unpadded frame count - so a result line is independent of batch mates, padding and rank, and any mix-up of utterances, lengths
or order shows in the file.    python tests/_stub_rank_worker.py <repo> <decode_asr argv ...>
"""
import sys
import types

import torch

REPO = sys.argv[1]
sys.path.insert(0, REPO)
from cassnat_asr_public_amd import dist as cdist  # noqa: E402
from cassnat_asr_public_amd.tasks import cassnat_task  # noqa: E402


class StubEngine:
    def __init__(self, batch, frames):
        self.cfg = types.SimpleNamespace(max_batch=batch, max_frames=frames)

    def close(self):
        pass


class StubModel(torch.nn.Module):
    _conf_dec = False
    _hyper = {}

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.zeros(4))

    def _check_args(self, args, lm_model):
        pass

    def weights_key(self):
        return tuple(p._version for p in self.parameters())

    def new_engine(self, batch, frames, with_weights=True, share=None):
        return StubEngine(batch, frames)

    def decode_device(self, feats, ratio, args, sos, engine=None, sub_batch=0, sub_rows=None, sub_frames=None, u_hint=0, want_ticket=False):
        B, T, _ = feats.shape
        own_T = [T] * B
        if sub_rows:  # a merged pass: the ratio of an utterance is relative to ITS batch's frame count
            own_T, o = [], 0
            for nb, t in zip(sub_rows, sub_frames):
                own_T += [t] * nb
        hyp = torch.zeros(B, 6, dtype=torch.int32)
        for b in range(B):
            uid = int(round(float(feats[b, 0, 0]))) - 1
            n = int(round(float(ratio[b]) * own_T[b]))
            assert n >= 1 and float(feats[b, n - 1, 0]) != 0 and (n == T or float(feats[b, n, 0]) == 0), "length / padding mix-up"
            hyp[b] = torch.tensor([sos, 4 + uid % 20, 4 + (uid // 20) % 20, 4 + n % 20, 4 + (n // 20) % 20, 2])
        out = (hyp, torch.full((B,), 6, dtype=torch.int32), torch.tensor([-float(b) for b in range(B)], dtype=torch.float64))
        return out + (-1,) if want_ticket else out


def _broadcast_stub(engine, src=0, group=None):  # (the one weight broadcast per rank: a collective every rank must issue)
    import torch.distributed as dist

    t = torch.arange(16, dtype=torch.int32) if dist.get_rank() == src else torch.zeros(16, dtype=torch.int32)
    dist.broadcast(t, src=src, group=group)
    assert t[15].item() == 15
    return 64


cassnat_task.make_cassnat_model = lambda input_size, args: StubModel()
cdist.broadcast_weights = _broadcast_stub
cassnat_task.cdist.broadcast_weights = _broadcast_stub

from cassnat_asr_public_amd.bin import decode_asr  # noqa: E402

sys.exit(decode_asr.main(sys.argv[2:]))
