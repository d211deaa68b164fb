"""Drop-in for the inference half of the reference's ``models.transformer`` (src/models/transformer.py): the
autoregressive (AST) model with joint CTC/attention beam search - BASELINE config 4, SURVEY 8a row a18.

Same surface: ``make_model(input_size, args) -> Transformer`` with the reference's parameter names, and
``Transformer.beam_decode(src, src_mask, vocab, args, lm_model=None) -> batch_top_seqs``.  By default the WHOLE search
runs on the device (``cn_decode_ast``): token embedding, the decoder layers on the NEW position only (keys/values of
the prefix come from a KV cache addressed through per-hypothesis ancestor tables; the reference re-runs the decoder
on the whole prefix), generator + log-softmax + top-k, the CTC prefix scorer and the beam bookkeeping
(transformer.py:157-240) - no host round trip per step.  ``args.hip_host_beam = True`` keeps the bookkeeping in
Python over ``cn_ast_begin / cn_ast_step / cn_ast_ctc_score`` (same results; used to cross-check the device beam).
"""
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

from .. import hip
from .cassnat import _ConvEmbedding, _Generator, _Params, _Stack, create_pe


class _Lut(_Params):
    def __init__(self, vocab, d):
        super().__init__()
        self.lut = nn.Embedding(vocab, d)


class Transformer(nn.Module):
    """Attribute names are the checkpoint key prefixes of the reference (src/models/transformer.py:55-60)."""

    def __init__(self, input_size, args):
        super().__init__()
        d = args.d_model
        self.src_embed = _ConvEmbedding(input_size, d)
        self.tgt_embed = nn.ModuleList([_Lut(args.vocab_size, d)])  # "tgt_embed.0.lut.weight"
        self.encoder = _Stack(d, args.d_ff, args.N_enc, True, False, True)
        self.decoder = _Stack(d, args.d_ff, args.N_dec, True, True, True)
        self.ctc_generator = _Generator(d, args.vocab_size)
        self.att_generator = _Generator(d, args.vocab_size)
        self.pe = create_pe(d)
        self._hyper = dict(input_size=input_size, d_model=d, n_head=args.n_head, d_encff=args.d_ff, d_decff=args.d_ff,
                           N_enc=args.N_enc, N_extra=0, N_self_dec=0, N_mix_dec=args.N_dec, vocab_size=args.vocab_size, ast=1)
        self.hip_precision = getattr(args, "hip_precision", "bf16")
        if self.hip_precision == "fp8":  # the e4m3 products are the NAT recogniser's encoder (BASELINE config 5): this model runs bf16 beside it
            self.hip_precision = "bf16"
        self.hip_max_batch = getattr(args, "hip_max_batch", 32)
        self.hip_max_frames = getattr(args, "hip_max_frames", 2048)
        self._engine = None
        self._engine_key = None

    def cuda(self, device=None):
        self._device = 0 if device is None else (device if isinstance(device, int) else torch.device(device).index or 0)
        return self

    def forward(self, *a, **k):
        raise NotImplementedError("training forward is out of scope; use beam_decode")

    def _new_handle(self, batch, frames, share_with=None, esa_group=1):
        return hip.Engine(SimpleNamespace(**self._hyper), precision=self.hip_precision,
                          max_batch=max(batch, self.hip_max_batch), max_frames=max(frames, self.hip_max_frames),
                          device=getattr(self, "_device", torch.cuda.current_device()), share_with=share_with, esa_group=esa_group)

    def teacher_score(self, feats, lm_input, target, length, n_per_utt, U, args):
        """ESA ranking with ``rank_model == 'at_baseline'`` (src/models/cassnat.py:514-520): this model's encoder on the B
        utterances, then its decoder teacher-forced on ``lm_input`` (N = B * n_per_utt rows, row e of utterance e % B) ->
        log-probabilities of ``target`` (float32 (N, ld) cuda; the reference takes softmax probabilities: ``.exp()``)."""
        B, T, _ = feats.shape
        eng = self._engine
        key = (tuple(p._version for p in self.parameters()), self.hip_precision)
        if (eng is None or self._engine_key != key or B > eng.cfg.max_batch or T > eng.cfg.max_frames
                or n_per_utt > max(1, eng.cfg.esa_group)):
            if eng is not None and self._engine_key == key:
                new = self._new_handle(B, T, share_with=eng, esa_group=n_per_utt)
                eng.close()
            else:
                if eng is not None:
                    eng.close()
                new = self._new_handle(B, T, esa_group=n_per_utt)
                new.load_state({k: v.detach() for k, v in self.named_parameters()}, self.pe)
            self._engine, self._engine_key = new, key
            eng = new
        opts = hip.CnDecodeOpts(padding_idx=int(args.padding_idx), sos=1, beam_width=1)
        score = torch.zeros(lm_input.shape, dtype=torch.float32, device=lm_input.device)
        eng.ast_teacher_score(feats, opts, lm_input, target, length, n_per_utt, U, score)
        return score

    def build_engine(self, batch, frames, with_weights=True):
        """The model's engine.  Current weights are never re-packed: a rebuild for a larger workspace shares the old handle's
        device blob.  ``with_weights=False``: layout only, the blob arrives by RCCL broadcast (dist.broadcast_weights)."""
        key = (tuple(p._version for p in self.parameters()), self.hip_precision)
        old = self._engine
        if old is not None and self._engine_key == key and old.finalized:
            eng = self._new_handle(batch, frames, share_with=old)
            old.close()
        else:
            if old is not None:
                old.close()
            if with_weights and getattr(self, "_remote_version", None) == key[0]:
                raise hip.HipError("this rank's parameters were never loaded (its weights arrived by broadcast)")
            eng = self._new_handle(batch, frames)
            if with_weights:
                eng.load_state({k: v.detach() for k, v in self.named_parameters()}, self.pe)
            else:
                eng.finalize()
                self._remote_version = key[0]
        self._engine, self._engine_key = eng, key
        return eng

    def new_engine(self, batch, frames, share):
        """A further handle (own workspace and KV cache) on the device blob of ``share``: what a decode pipeline owns."""
        return self._new_handle(batch, frames, share_with=share)

    def engine(self, batch, frames):
        key = (tuple(p._version for p in self.parameters()), self.hip_precision)
        if (self._engine is None or self._engine_key != key or batch > self._engine.cfg.max_batch
                or frames > self._engine.cfg.max_frames):
            self.build_engine(batch, frames)
        return self._engine

    def beam_decode(self, src, src_mask, vocab, args, lm_model=None, engine=None):
        """Same contract as the reference's Transformer.beam_decode (src/models/transformer.py:122-241), lm_weight == 0.
        ``engine``: run on this handle (a decode pipeline's, see ``new_engine``) instead of the model's own."""
        if getattr(args, "lm_weight", 0) > 0:
            raise NotImplementedError("LM fusion is outside the accelerated path")
        sos, eos = vocab.word2index["sos"], vocab.word2index["eos"]
        assert vocab.word2index["blank"] == args.padding_idx
        dev = torch.device("cuda", getattr(self, "_device", torch.cuda.current_device()))
        feats = src.to(dev, torch.float32).contiguous()
        B, T, _ = feats.shape
        Tp = ((T - 1) // 2 + 1 - 1) // 2 + 1
        eng = engine if engine is not None else self.engine(B, T)
        use_ctc = args.ctc_weight > 0
        bw = int(args.beam_width)
        K = int(args.ctc_beam) if use_ctc else bw
        max_step = int(args.max_decode_ratio * Tp) if args.max_decode_ratio > 0 else Tp
        max_len = max_step + 1
        opts = hip.CnDecodeOpts(padding_idx=int(args.padding_idx), sos=sos, beam_width=1)
        lp = args.length_penalty
        if not getattr(args, "hip_host_beam", False):
            ao = hip.CnAstOpts(ctc_weight=float(args.ctc_weight) if use_ctc else 0.0, temperature=float(args.T), ctc_beam=K,
                               beam_width=bw, max_step=max_step, eos=eos, use_length_penalty=int(lp is not None),
                               one_minus_ctc_weight=float(1 - args.ctc_weight),
                               length_penalty=float(lp) if lp is not None else 0.0)
            hyp = torch.empty(B, bw, max_len, dtype=torch.int32, device=dev)
            hlen = torch.empty(B, bw, dtype=torch.int32, device=dev)
            score = torch.empty(B, bw, dtype=torch.float64, device=dev)
            eng.ast_decode(feats, opts, ao, hyp, hlen, score)
            hyp, hlen, score = hyp.cpu().numpy(), hlen.cpu().numpy(), score.cpu().numpy()
            eng.check_range("Transformer.beam_decode")  # (fp16 engines: the features were inside the half-precision range)
            out = []
            for b in range(B):
                row = []
                for j in range(bw):
                    h = hyp[b, j, : hlen[b, j]].tolist()
                    row.append({"ys": torch.tensor([h], dtype=torch.long), "score": float(score[b, j]), "hyp": h})
                out.append(row)
            return out
        eng.ast_begin(feats, opts, use_ctc, max_len, B * bw, K if use_ctc else 0)
        w32 = np.float32(args.ctc_weight)
        u32 = np.float32(1 - args.ctc_weight)

        beams = [[{"score": 0.0, "hyp": [sos], "anc": [], "ctc_ref": -1 - b, "ctc_prev": np.float32(0.0)}] for b in range(B)]
        idx_d = torch.empty(B * bw, K, dtype=torch.int32, device=dev)
        val_d = torch.empty(B * bw, K, dtype=torch.float32, device=dev)
        ctc_d = torch.empty(B * bw, K, dtype=torch.float32, device=dev)
        for i in range(max_step):
            live = [(b, s) for b in range(B) for s in beams[b] if s["hyp"][-1] != eos]
            if not live:
                break
            n = len(live)
            tok = np.array([s["hyp"][-1] for _, s in live], np.int32)
            utt = np.array([b for b, _ in live], np.int32)
            anc = np.zeros((n, max_len), np.int32)
            keyok = np.zeros((n, max_len), np.uint8)
            for k, (_, s) in enumerate(live):
                anc[k, :i] = s["anc"]
                anc[k, i] = k
                keyok[k, : i + 1] = [t != args.padding_idx for t in s["hyp"]]
            tok_d, utt_d = torch.from_numpy(tok).to(dev), torch.from_numpy(utt).to(dev)
            anc_d, keyok_d = torch.from_numpy(anc).to(dev), torch.from_numpy(keyok).to(dev)
            eng.ast_step(i, tok_d, utt_d, anc_d, keyok_d, args.T, K, idx_d[:n], val_d[:n])
            if use_ctc:
                ref_d = torch.from_numpy(np.array([s["ctc_ref"] for _, s in live], np.int32)).to(dev)
                eng.ast_ctc_score(i, utt_d, tok_d, idx_d[:n], ref_d, i & 1, eos, ctc_d[:n])
                ctc = ctc_d[:n].cpu().numpy()
            indices, att = idx_d[:n].cpu().numpy(), val_d[:n].cpu().numpy()
            if use_ctc:
                prev = np.array([s["ctc_prev"] for _, s in live], np.float32)[:, None]
                local = w32 * (ctc - prev) + u32 * att  # float32, same op order as transformer.py:205-206
                local_idx = np.argsort(-local, axis=1, kind="stable")[:, :bw]
                local_scores = np.take_along_axis(local, local_idx, 1)
                tokens = np.take_along_axis(indices, local_idx, 1)
            else:
                local_scores, tokens = att[:, :bw], indices[:, :bw]
            cand = [[s for s in beams[b] if s["hyp"][-1] == eos] for b in range(B)]
            for k, (b, s) in enumerate(live):
                for j in range(bw):
                    new = {"score": s["score"] + float(local_scores[k, j]), "hyp": s["hyp"] + [int(tokens[k, j])],
                           "anc": s["anc"] + [k]}
                    if use_ctc:
                        ti = int(local_idx[k, j])
                        new["ctc_ref"] = k * K + ti
                        new["ctc_prev"] = ctc[k, ti]
                    cand[b].append(new)
            for b in range(B):
                if lp is not None:
                    cand[b].sort(key=lambda x: x["score"] + (len(x["hyp"]) - 1) * lp, reverse=True)
                else:
                    cand[b].sort(key=lambda x: x["score"], reverse=True)
                beams[b] = cand[b][:bw]
        eng.check_range("Transformer.beam_decode")
        return [[{"ys": torch.tensor([s["hyp"]], dtype=torch.long), "score": s["score"], "hyp": s["hyp"]} for s in beams[b]]
                for b in range(B)]

    def fast_decode_with_ctc(self, src, src_mask, vocab, args, lm_model=None, engine=None):
        """Same contract as the reference's Transformer.fast_decode_with_ctc (src/models/transformer.py:243-342; ArtTask decode_type
        'ctc_correct'), lm_weight == 0: the CTC greedy hypothesis is the decoder's teacher-forced input, the decoder a correction
        model.  Encoder, CTC collapse, decoder and the per-row top-k run on the device (``cn_ast_ctc_correct``); the finish loop -
        O(rows * beam^2) numbers - is the reference's, here on the host: row i is consumed while i <= length[b], an eos is scored
        but not appended."""
        if getattr(args, "lm_weight", 0) > 0 or lm_model is not None:
            raise NotImplementedError("LM fusion is outside the accelerated path")
        sos, eos = vocab.word2index["sos"], vocab.word2index["eos"]
        assert vocab.word2index["blank"] == args.padding_idx
        dev = torch.device("cuda", getattr(self, "_device", torch.cuda.current_device()))
        feats = src.to(dev, torch.float32).contiguous()
        B, T, _ = feats.shape
        eng = engine if engine is not None else self.engine(B, T)
        bw = int(args.beam_width)
        opts = hip.CnDecodeOpts(padding_idx=int(args.padding_idx), sos=sos, beam_width=1)
        length, tok, val = eng.ast_ctc_correct(feats, opts, bw)
        length, tok, val = length.cpu().numpy(), tok.cpu().numpy(), val.cpu().numpy()
        eng.check_range("Transformer.fast_decode_with_ctc")
        lp = args.length_penalty
        ys = torch.ones(1, 1).fill_(sos).long()
        out = []
        for b in range(B):
            beams = [{"ys": ys, "score": 0.0, "hyp": [sos]}]
            for i in range(int(length[b]) + 1):
                cand = [{"ys": ys, "score": s["score"] + float(val[b, i, j]),
                         "hyp": s["hyp"] + [int(tok[b, i, j])] if int(tok[b, i, j]) != eos else s["hyp"]}
                        for s in beams for j in range(bw)]
                if lp is not None:
                    cand.sort(key=lambda s: s["score"] + (len(s["hyp"]) - 1) * lp, reverse=True)
                else:
                    cand.sort(key=lambda s: s["score"], reverse=True)
                beams = cand[:bw]
            out.append(beams)
        return out


def make_model(input_size, args):
    """Same role as src/models/transformer.py:19-37."""
    model = Transformer(input_size, args)
    for p in model.parameters():
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)
    return model
