"""Drop-in for the inference use of the reference's ``models.lm`` (src/models/lm.py): the TransformerLM that ranks the ESA
samples of CASS-NAT decoding (src/models/cassnat.py:499-523, ``rank_model == 'lm'``).

``make_model(args) -> TransformerLM`` holds the reference's parameter names (``text_embed.0.lut.weight``, ``encoder.*``,
``out_generator.proj.*``; checkpoint keys of src/tasks/cassnat_task.py:85-125).  No arithmetic lives here:
``score_tokens`` hands the tokens to libcassnat_hip.so (``cn_lm_score``: embedding, encoder stack under the causal + length
mask, generator, log-softmax, gather of the target's log-probability)."""
from types import SimpleNamespace

import torch
import torch.nn as nn

from .. import hip
from .cassnat import _Generator, _Params, _Stack, create_pe


class _Lut(_Params):
    def __init__(self, vocab, d):
        super().__init__()
        self.lut = nn.Embedding(vocab, d)


class TransformerLM(nn.Module):
    def __init__(self, args):
        super().__init__()
        d = args.d_model
        self.text_embed = nn.ModuleList([_Lut(args.vocab_size, d)])  # "text_embed.0.lut.weight" (index 1 = PositionalEncoding, no parameters)
        self.encoder = _Stack(d, args.d_ff, args.N, True, False, True)
        self.out_generator = _Generator(d, args.vocab_size)
        self.pe = create_pe(d)
        self._hyper = dict(input_size=80, d_model=d, n_head=args.n_head, d_encff=args.d_ff, d_decff=args.d_ff, N_enc=args.N,
                           N_extra=0, N_self_dec=0, N_mix_dec=0, vocab_size=args.vocab_size, ast=2)
        self.hip_precision = getattr(args, "hip_precision", "bf16")
        if self.hip_precision == "fp8":  # the e4m3 products are the NAT recogniser's encoder (BASELINE config 5): this model runs bf16 beside it
            self.hip_precision = "bf16"
        self._engine = None
        self._engine_key = None

    def cuda(self, device=None):
        self._device = 0 if device is None else (device if isinstance(device, int) else torch.device(device).index or 0)
        return self

    def forward(self, *a, **k):
        raise NotImplementedError("use score_tokens (the log-probability tensor of the reference's forward is never materialised)")

    def engine(self, rows_batch, max_frames):
        key = (tuple(p._version for p in self.parameters()), self.hip_precision)
        if (self._engine is None or self._engine_key != key or rows_batch > self._engine.cfg.max_batch
                or max_frames > self._engine.cfg.max_frames):
            if self._engine is not None:
                self._engine.close()
            eng = hip.Engine(SimpleNamespace(**self._hyper), precision=self.hip_precision, max_batch=rows_batch,
                             max_frames=max_frames, device=getattr(self, "_device", torch.cuda.current_device()))
            eng.load_state({k: v.detach() for k, v in self.named_parameters()}, self.pe)
            self._engine, self._engine_key = eng, key
        return self._engine

    def score_tokens(self, lm_input, target, length, U, max_frames=2048):
        """lm_input / target int32 cuda (N, ld), length int32 cuda (N,) -> float32 cuda (N, ld): log p(target[n][u] | lm_input[n][..u])
        for u < U under the mask (j <= u and j < length[n]) - what cassnat.py:507-520 gathers from lm_model(lm_input, mask)."""
        eng = self.engine(lm_input.shape[0], max_frames)
        score = torch.zeros(lm_input.shape, dtype=torch.float32, device=lm_input.device)
        eng.lm_score(lm_input, target, length, U, score)
        return score


def make_model(args):
    """Same role as src/models/lm.py:16-31."""
    model = TransformerLM(args)
    for p in model.parameters():
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)
    return model
