"""Drop-in for the inference half of the reference's ``models.cassnat`` (src/models/cassnat.py).

Same surface - ``make_model(input_size, args) -> CassNAT`` and
``CassNAT.beam_decode(src, x_mask, src_size, vocab, args, lm_model=None, ...) -> (batch_top_seqs, args)`` -
and the same parameter names (they are the checkpoint keys loaded by
``BaseTask.load_test_model``, src/tasks/base_task.py:45-54), but no arithmetic lives here: the module
tree only *holds* parameters; ``beam_decode`` hands them once to libcassnat_hip.so and every stage
(conv subsampling, encoder, CTC alignment, extractor, NAT decoder, greedy finish) runs as hand-written
gfx950 kernels behind the C ABI (include/cassnat_hip.h).  There is no CPU fallback.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .. import hip


class _OneHostThread:
    """The host-side tensor arithmetic of a decode call (masks, means and picks over a few hundred thousand numbers) runs
    single-threaded.  Measured: left to torch's intra-op pool, those few CPU ops start an OpenMP team whose idle workers keep
    spinning; on a GPU box whose process has a CPU quota (16 cores for one GPU) the spinning exhausts the cgroup's CFS quota
    and the whole process - the thread that launches kernels included - is throttled for the rest of the 100 ms period:
    sporadic 60-90 ms stalls on every other ESA decode (18 ms each otherwise; tools/esa_stall_probe.py)."""

    def __enter__(self):
        self.n = torch.get_num_threads()
        if self.n != 1:
            torch.set_num_threads(1)
        return self

    def __exit__(self, *exc):
        if self.n != 1:
            torch.set_num_threads(self.n)
        return False


def create_pe(d_model, max_len=5000):
    """Sinusoid table, same closed form as src/models/cassnat.py:91-99 (a buffer, not a parameter)."""
    position = torch.arange(0.0, max_len).unsqueeze(1)
    div_term = torch.exp(torch.arange(0.0, d_model, 2) * -(math.log(10000.0) / d_model))
    pe = torch.zeros(max_len, d_model)
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe


# ---- parameter holders: names mirror the reference module tree, there is deliberately no forward() ----
class _Params(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: computation runs in libcassnat_hip.so via CassNAT.beam_decode")


class _Linear(_Params):
    def __init__(self, n_in, n_out):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n_out, n_in))
        self.bias = nn.Parameter(torch.empty(n_out))
        bound = 1.0 / math.sqrt(n_in)
        nn.init.uniform_(self.weight, -bound, bound)
        nn.init.uniform_(self.bias, -bound, bound)


class _Conv(_Params):
    def __init__(self, c_in, c_out):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(c_out, c_in, 3, 3))
        self.bias = nn.Parameter(torch.empty(c_out))
        bound = 1.0 / math.sqrt(c_in * 9)
        nn.init.uniform_(self.weight, -bound, bound)
        nn.init.uniform_(self.bias, -bound, bound)


class _Norm(_Params):
    def __init__(self, d):
        super().__init__()
        self.a_2 = nn.Parameter(torch.ones(d))
        self.b_2 = nn.Parameter(torch.zeros(d))


class _Sublayer(_Params):
    def __init__(self, d):
        super().__init__()
        self.norm = _Norm(d)


class _Attention(_Params):
    def __init__(self, d):
        super().__init__()
        self.linears = nn.ModuleList([_Linear(d, d) for _ in range(4)])  # 0=Q 1=K 2=V 3=O


class _FeedForward(_Params):
    def __init__(self, d, d_ff):
        super().__init__()
        self.w_1 = _Linear(d, d_ff)
        self.w_2 = _Linear(d_ff, d)


class _Block(_Params):
    def __init__(self, d, d_ff, self_attn, src_attn):
        super().__init__()
        if self_attn:
            self.self_attn = _Attention(d)
        if src_attn:
            self.src_attn = _Attention(d)
        self.feed_forward = _FeedForward(d, d_ff)
        self.sublayer = nn.ModuleList([_Sublayer(d) for _ in range(1 + int(self_attn) + int(src_attn))])


class _Stack(_Params):
    def __init__(self, d, d_ff, n, self_attn, src_attn, final_norm):
        super().__init__()
        self.layers = nn.ModuleList([_Block(d, d_ff, self_attn, src_attn) for _ in range(n)])
        if final_norm:
            self.norm = _Norm(d)


# ---- conformer variants (src/models/cassnat.py:29-57; modules/attention.py:68-147, conformer_related.py:15-44) ----
class _RelPos(_Params):
    """RelativePositionalEncoding: the frozen sinusoid rows are a (non-trainable) parameter of the checkpoint."""

    def __init__(self, d, max_rel):
        super().__init__()
        self.embedding = nn.Embedding.from_pretrained(create_pe(d, 2 * max_rel + 1), freeze=True)


class _RelAttention(_Params):
    def __init__(self, d, h):
        super().__init__()
        self.pos_bias_u = nn.Parameter(torch.empty(h, d // h))
        self.pos_bias_v = nn.Parameter(torch.empty(h, d // h))
        nn.init.xavier_uniform_(self.pos_bias_u)
        nn.init.xavier_uniform_(self.pos_bias_v)
        self.linears = nn.ModuleList([_Linear(d, d) for _ in range(4)])
        self.linear_pos = nn.Linear(d, d, bias=False)


class _Conv1d(_Params):
    def __init__(self, c_out, c_in_per_group, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(c_out, c_in_per_group, k))
        self.bias = nn.Parameter(torch.empty(c_out))
        bound = 1.0 / math.sqrt(c_in_per_group * k)
        nn.init.uniform_(self.weight, -bound, bound)
        nn.init.uniform_(self.bias, -bound, bound)


class _GroupNorm(_Params):
    def __init__(self, d):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(d))
        self.bias = nn.Parameter(torch.zeros(d))


class _ConvModule(_Params):
    def __init__(self, d, k):
        super().__init__()
        self.pointwise_conv1 = _Conv1d(2 * d, d, 1)
        self.depthwise_conv = _Conv1d(d, 1, k)
        self.norm = _GroupNorm(d)
        self.pointwise_conv2 = _Conv1d(d, d, 1)


class _ConformerBlock(_Params):
    """SelfAttLayer / MixAttLayer of fanat_conformer_blocks.py: registration order src_attn?, self_attn, feed_forward1,
    conv_module, feed_forward2, sublayer."""

    def __init__(self, d, h, d_ff, k, src_attn):
        super().__init__()
        if src_attn:
            self.src_attn = _Attention(d)
        self.self_attn = _RelAttention(d, h)
        self.feed_forward1 = _FeedForward(d, d_ff)
        self.conv_module = _ConvModule(d, k)
        self.feed_forward2 = _FeedForward(d, d_ff)
        self.sublayer = nn.ModuleList([_Sublayer(d) for _ in range(5 if src_attn else 4)])


class _ConformerStack(_Params):
    def __init__(self, d, h, d_ff, k, n, src_attn, final_norm):
        super().__init__()
        self.layers = nn.ModuleList([_ConformerBlock(d, h, d_ff, k, src_attn) for _ in range(n)])
        if final_norm:
            self.norm = _Norm(d)


class _ConformerExtractorLayer(_Params):
    """SrcAttLayer (fanat_conformer_blocks.py:41-60): src_attn, feed_forward, ONE sublayer connection, pos_enc."""

    def __init__(self, d, d_ff, max_rel):
        super().__init__()
        self.src_attn = _Attention(d)
        self.feed_forward = _FeedForward(d, d_ff)
        self.sublayer = _Sublayer(d)
        self.pos_enc = _RelPos(d, max_rel)


class _ConformerExtractor(_Params):
    def __init__(self, d, d_ff, max_rel, n):
        super().__init__()
        assert n == 1, "the reference's conformer extractor has exactly one layer (fanat_conformer_blocks.py:178)"
        self.layers = nn.ModuleList([_ConformerExtractorLayer(d, d_ff, max_rel)])


class _ConvEmbedding(_Params):
    def __init__(self, input_size, d, rel_max=None):
        super().__init__()
        # indices 0 and 2 as in nn.Sequential(conv1, ReLU, conv2, ReLU)  (src/models/modules/embedding.py:102-105)
        self.conv = nn.ModuleDict({"0": _Conv(1, d), "2": _Conv(d, d)})
        self.linear_out = _Linear(d * (((input_size - 1) // 2) // 2 + 1), d)
        if rel_max is not None:
            self.pos_enc = _RelPos(d, rel_max)


class _Generator(_Params):
    def __init__(self, d, vocab):
        super().__init__()
        self.proj = _Linear(d, vocab)


class CassNAT(nn.Module):
    """Attribute names are the checkpoint key prefixes of the reference (src/models/cassnat.py:118-125)."""

    #: engine settings, overridable through ``args`` (hip_precision, hip_max_batch, hip_max_frames)
    def __init__(self, input_size, args):
        super().__init__()
        d, h = args.d_model, args.n_head
        self.input_size = input_size
        conf_enc, conf_dec = bool(getattr(args, "use_conv_enc", False)), bool(getattr(args, "use_conv_dec", False))
        if conf_enc or conf_dec:
            assert getattr(args, "pos_type", "absolute") == "relative", "conformer must use relative positional encoding"
        if conf_enc:
            self.src_embed = _ConvEmbedding(input_size, d, args.enc_max_relative_len)
            self.encoder = _ConformerStack(d, h, args.d_encff, args.enc_kernel_size, args.N_enc, False, True)
        else:
            self.src_embed = _ConvEmbedding(input_size, d)
            self.encoder = _Stack(d, args.d_encff, args.N_enc, True, False, True)
        if conf_dec:
            self.acembed_extractor = _ConformerExtractor(d, args.d_ff, args.dec_max_relative_len, args.N_extra)
            self.embed_mapper = _ConformerStack(d, h, args.d_decff, args.dec_kernel_size, args.N_self_dec, False, False)
            self.decoder = _ConformerStack(d, h, args.d_decff, args.dec_kernel_size, args.N_mix_dec, True, True)
        else:
            self.acembed_extractor = _Stack(d, args.d_decff, args.N_extra, False, True, False)
            self.embed_mapper = _Stack(d, args.d_decff, args.N_self_dec, True, False, False)
            self.decoder = _Stack(d, args.d_decff, args.N_mix_dec, True, True, True)
        self.ctc_generator = _Generator(d, args.vocab_size)
        self.att_generator = _Generator(d, args.vocab_size)
        self.pe = create_pe(d)
        self._hyper = dict(input_size=input_size, d_model=d, n_head=args.n_head, d_encff=args.d_encff,
                           d_decff=args.d_decff, N_enc=args.N_enc, N_extra=args.N_extra, N_self_dec=args.N_self_dec,
                           N_mix_dec=args.N_mix_dec, vocab_size=args.vocab_size, conf_enc=int(conf_enc), conf_dec=int(conf_dec),
                           enc_max_rel=getattr(args, "enc_max_relative_len", 0) if conf_enc else 0,
                           dec_max_rel=getattr(args, "dec_max_relative_len", 0) if conf_dec else 0,
                           enc_kernel=getattr(args, "enc_kernel_size", 0) if conf_enc else 0,
                           dec_kernel=getattr(args, "dec_kernel_size", 0) if conf_dec else 0,
                           d_ff=getattr(args, "d_ff", 0) if conf_dec else 0)
        self._conf_dec = conf_dec
        self.hip_precision = getattr(args, "hip_precision", "bf16")
        self._hyper["fp8_scope"], self._hyper["fp8_ffn_first_layer"] = hip.parse_fp8_scope(getattr(args, "hip_fp8_scope", "all"))
        self.hip_max_batch = getattr(args, "hip_max_batch", 32)
        self.hip_max_frames = getattr(args, "hip_max_frames", 2048)
        self._engine = None
        self._engine_key = None
        self._invalidations = 0  # bumped by invalidate_engine / load_state_dict: holders of further engine handles key on it

    # the reference moves the module with .cuda(); parameters stay where they are - the engine owns HBM copies
    def cuda(self, device=None):
        self._device = 0 if device is None else (device if isinstance(device, int) else torch.device(device).index or 0)
        return self

    def forward(self, *a, **k):
        raise NotImplementedError("training forward is out of scope; use beam_decode")

    # ------------------------------------------------------------------------------------------ engine
    def _weights_version(self):
        return tuple(p._version for p in self.parameters())

    def _new_handle(self, batch, frames, esa_group=1, share_with=None):
        from types import SimpleNamespace

        return hip.Engine(SimpleNamespace(**self._hyper), precision=self.hip_precision,
                          max_batch=max(batch, self.hip_max_batch), max_frames=max(frames, self.hip_max_frames),
                          device=getattr(self, "_device", torch.cuda.current_device()), esa_group=esa_group, share_with=share_with)

    def build_engine(self, batch, frames, with_weights=True, esa_group=1):
        """Create the model's HIP engine.  ``with_weights=False`` allocates the (layout-identical) weight blob only: the
        contents then arrive by RCCL broadcast from the rank that read the checkpoint (cassnat_asr_public_amd.dist).

        An engine whose weights are still current (same parameter versions, same precision) is never re-packed: the new
        handle shares its device blob (``cn_model_create_shared``) and only gets the larger workspace.  That also makes a
        rebuild rank-safe - on a rank that received its weights by broadcast the local ``nn.Parameter``s were never loaded,
        so packing from them would silently decode with random weights."""
        key = (self._weights_version(), self.hip_precision)
        old = self._engine
        if old is not None and self._engine_key == key and old.finalized:
            eng = self._new_handle(batch, frames, esa_group, share_with=old)
            old.close()  # (the blob stays: reference counted in the library)
        else:
            if old is not None:
                old.close()
            if with_weights and self._params_unloaded():
                raise hip.HipError("this rank's parameters were never loaded (its weights arrived by broadcast): the engine cannot "
                                   "be re-packed from them - load the checkpoint here, or broadcast again into a new engine")
            eng = self._new_handle(batch, frames, esa_group)
            if with_weights:
                eng.load_state({k: v.detach() for k, v in self.named_parameters()}, self.pe)
            else:
                eng.finalize()
                self._remote_version = key[0]  # the parameters as they are now are NOT what the engine will hold
        self._engine, self._engine_key = eng, key
        return eng

    def new_engine(self, batch, frames, with_weights=True, share=None):
        """An additional engine handle for the same parameters: what a decode pipeline owns (cassnat_asr_public_amd.pipeline).
        ``share``: an engine whose device copy of the packed weights the new handle uses (own workspace only) - the pipelines of
        one GPU share ONE blob.  Without it the handle packs (``with_weights``) or allocates (broadcast receiver) its own.
        The model's own engine (``engine()``) is not touched."""
        if share is not None:
            return self._new_handle(batch, frames, share_with=share)
        eng = self._new_handle(batch, frames)
        if with_weights:
            if self._params_unloaded():
                raise hip.HipError("this rank's parameters were never loaded: share an engine that received the broadcast instead")
            eng.load_state({k: v.detach() for k, v in self.named_parameters()}, self.pe)
        else:
            eng.finalize()
        return eng

    def invalidate_engine(self):
        """Call after changing parameters through ``.data`` (which does not bump tensor versions)."""
        self._engine_key = None
        self._invalidations += 1

    def weights_key(self):
        """What an engine handle's packed weights depend on: parameter versions, explicit invalidations, precision and the fp8
        scope.  Holders of handles of their own (a task's cached decode pipelines) rebuild when it changes."""
        return (self._weights_version(), self._invalidations, self.hip_precision, self._hyper["fp8_scope"], self._hyper["fp8_ffn_first_layer"])

    def _params_unloaded(self):
        """True while the local nn.Parameters are still the untouched initial values of a rank whose engine weights came by
        broadcast (``build_engine(with_weights=False)``); any later in-place load changes the tensor versions."""
        return getattr(self, "_remote_version", None) is not None and self._remote_version == self._weights_version()

    def load_state_dict(self, *a, **k):
        self.invalidate_engine()
        self._remote_version = None
        return super().load_state_dict(*a, **k)

    def engine(self, batch, frames, esa_group=1):
        """(Re)build the HIP engine when weights changed or the workspace is too small."""
        key = (self._weights_version(), self.hip_precision)
        if (self._engine is None or self._engine_key != key or batch > self._engine.cfg.max_batch
                or frames > self._engine.cfg.max_frames or esa_group > max(1, self._engine.cfg.esa_group)):
            self.build_engine(batch, frames, esa_group=esa_group)
        return self._engine

    # ------------------------------------------------------------------------------------------ decode
    def _check_args(self, args, lm_model):
        dtype = getattr(args, "decode_type", "att_only")
        if not getattr(args, "use_trigger", True) and (dtype != "att_only" or getattr(args, "sample_num", 0) > 1):
            # (the reference reads decode_type / sample_num only inside `if args.use_trigger:`, src/models/cassnat.py:435-468)
            raise NotImplementedError("use_trigger=False goes with decode_type att_only and sample_num <= 1")
        if dtype not in ("att_only", "ctc_att"):
            raise NotImplementedError("decode_type '%s' is outside the accelerated path (att_only, ctc_att; ctc_only goes through "
                                      "utils.beam_decode.ctc_beam_decode)" % dtype)
        if dtype == "ctc_att" and getattr(args, "sample_num", 0) > 1:
            raise NotImplementedError("ctc_att with sample_num > 1 (the n best CTC hypotheses ranked by a language model) needs the "
                                      "in-loop LM fusion of ctc_beam_decode, which is outside the accelerated path")
        if getattr(args, "lm_weight", 0) > 0 and lm_model is not None:
            raise NotImplementedError("LM shallow fusion in the finish loop is outside the accelerated path")
        if getattr(args, "sample_num", 0) > 1:
            rank = getattr(args, "rank_model", "lm")
            ok = (lm_model is not None and ((rank == "lm" and hasattr(lm_model, "score_tokens")) or
                                            (rank == "at_baseline" and hasattr(lm_model, "teacher_score")) or
                                            (rank == "n-gram" and hasattr(lm_model, "score"))))
            if not ok:
                raise NotImplementedError("ESA ranking needs rank_model 'lm' (models.lm.TransformerLM), 'at_baseline' "
                                          "(models.transformer.Transformer) or 'n-gram' (an object with kenlm's score(str))")
            if not 1 <= int(args.beam_width) <= 16:
                raise NotImplementedError("ESA: beam_width must be in [1, 16]")
        if getattr(args, "test_hitrate", False):
            raise NotImplementedError("test_hitrate needs the training-time viterbi aligner")
        if self._conf_dec and getattr(args, "use_unimask", False):
            raise NotImplementedError("use_unimask with the conformer decoder: the reference itself cannot run it "
                                      "(cassnat.py:486-488 indexes the (x, pos_embed) tuple)")

    def _range_ok(self):
        """fp16 engine: the features of the call just finished were inside the range its half-precision operands hold."""
        if self.hip_precision in ("fp16", "bf16x3") and self._engine is not None:
            self._engine.check_range("beam_decode")

    def decode_device(self, src, src_size, args, sos=1, engine=None, sub_batch=0, sub_rows=None, sub_frames=None, u_hint=0,
                      want_ticket=False):
        """The device half of beam_decode: returns cuda tensors (hyp (B,S) int32, hyp_len (B,) int32, score (B,) f64).
        ``engine``: run on this handle (a decode pipeline's) instead of the model's own.
        ``sub_rows`` / ``sub_frames``: the call is a merged pass over that many reference batches of those frame counts
        (``hip.Engine.decode_merged``); ``u_hint`` > 0: predicted row count, no mid-pass host sync; ``want_ticket``: also return
        the ticket (the caller checks ``engine.ticket(t)`` once the stream has drained)."""
        dev = torch.device("cuda", getattr(self, "_device", torch.cuda.current_device()))
        feats = src.to(dev, torch.float32).contiguous()
        ratio = src_size.to(dev, torch.float32).contiguous()
        B, T, _ = feats.shape
        eng = engine if engine is not None else self.engine(B, T)
        opts = hip.Engine.make_opts(args, capture=getattr(args, "hip_capture", False))
        opts.sos = sos
        opts.sub_batch = int(sub_batch)  # coalesced batches of that size each (pipeline.DecodePipelines): per-batch hypotheses
        stride = ((T - 1) // 2 + 1 - 1) // 2 + 1 + 2
        hyp = torch.empty(B, stride, dtype=torch.int32, device=dev)
        hyp_len = torch.empty(B, dtype=torch.int32, device=dev)
        score = torch.empty(B, dtype=torch.float64, device=dev)
        if sub_rows or u_hint or want_ticket:
            t = eng.decode_merged(feats, ratio, opts, sub_rows, sub_frames, hyp, hyp_len, score, u_hint=u_hint)
            return (hyp, hyp_len, score, t) if want_ticket else (hyp, hyp_len, score)
        eng.decode(feats, ratio, opts, hyp, hyp_len, score)
        return hyp, hyp_len, score

    def beam_decode(self, src, x_mask, src_size, vocab, args, lm_model=None, ctc_top_seqs=None, labels=None,
                    label_sizes=None):
        """Same contract as the reference's CassNAT.beam_decode (src/models/cassnat.py:420-637) for
        ``use_trigger=True, sample_num<=1, decode_type='att_only', lm_weight==0`` (anything else raises).

        ``x_mask`` is accepted for signature compatibility; like the reference's caller
        (src/tasks/cassnat_task.py:328) the padding mask is ``src[:,:,0] != padding_idx`` and is re-derived on
        the device from ``src`` itself.
        """
        self._check_args(args, lm_model)
        sos = vocab.word2index["sos"]
        assert vocab.word2index["blank"] == args.padding_idx, "CTC blank id and padding_idx must agree"
        if getattr(args, "sample_num", 0) > 1:
            with _OneHostThread():
                out = self._esa_decode(src, src_size, args, lm_model, sos, vocab)
            self._range_ok()
            return out, args
        if getattr(args, "decode_type", "att_only") == "ctc_att":
            hyp, hyp_len, score = self._decode_forced(src, src_size, args, sos, ctc_top_seqs)
        else:
            hyp, hyp_len, score = self.decode_device(src, src_size, args, sos)
        if args.beam_width > 1:
            out = self._host_beam(self._engine, args, sos)
            self._range_ok()
            return out, args
        hyp_h, len_h, score_h = hyp.cpu().numpy(), hyp_len.cpu().numpy(), score.cpu().numpy()
        self._range_ok()
        if self.hip_precision == "fp16":
            hip.check_fp16_range(score_h, "beam_decode")
        ys = torch.ones(1, 1).fill_(sos).long()
        out = []
        for b in range(hyp_h.shape[0]):
            out.append([{"ys": ys, "score": float(score_h[b]), "hyp": hyp_h[b, : len_h[b]].tolist()}])
        return out, args

    def _decode_forced(self, src, src_size, args, sos, ctc_top_seqs):
        """decode_type 'ctc_att' (src/models/cassnat.py:446-448): the decoder side is triggered by the forced (Viterbi)
        alignment of the best CTC beam hypothesis of every utterance - ``ctc_top_seqs[b][0]['hyp']``, what
        ``utils.beam_decode.ctc_beam_decode`` returned for this batch - instead of the greedy CTC path."""
        if ctc_top_seqs is None:
            raise ValueError("decode_type 'ctc_att' needs ctc_top_seqs (utils.beam_decode.ctc_beam_decode of the same batch)")
        dev = torch.device("cuda", getattr(self, "_device", torch.cuda.current_device()))
        feats = src.to(dev, torch.float32).contiguous()
        ratio = src_size.to(dev, torch.float32).contiguous()
        B, T, _ = feats.shape
        Tp = ((T - 1) // 2 + 1 - 1) // 2 + 1
        lab = [list(ctc_top_seqs[b][0]["hyp"]) for b in range(B)]
        ymax = max(len(x) for x in lab)
        labels = np.zeros((B, max(ymax, 1)), np.int32)
        for b, x in enumerate(lab):
            labels[b, : len(x)] = x
        eng = self.engine(B, T)
        opts = hip.Engine.make_opts(args, capture=getattr(args, "hip_capture", False))
        opts.sos = sos
        hyp = torch.empty(B, Tp + 2, dtype=torch.int32, device=dev)
        hyp_len = torch.empty(B, dtype=torch.int32, device=dev)
        score = torch.empty(B, dtype=torch.float64, device=dev)
        eng.decode_forced(feats, ratio, opts, torch.from_numpy(labels).to(dev), torch.tensor([len(x) for x in lab], dtype=torch.int32, device=dev),
                          ymax, hyp, hyp_len, score)
        return hyp, hyp_len, score

    def _esa_decode(self, src, src_size, args, lm_model, sos, vocab=None):
        """Error-based sampling of alignments + LM ranking (src/models/cassnat.py:370-376, 441-445, 499-561; sample_num > 1,
        rank_model 'lm', lm_weight 0, beam_width 1).  The encoder and the CTC generator run once; the samples (one alignment
        per utterance each, sample 0 = the best path) go through the decoder side ``args.hip_esa_group`` (default 16) at a time -
        one device pass of B * group query sets over the B utterances' encoder outputs (alignment -> extractor -> decoder ->
        generator argmax); the TransformerLM scores all samples' tokens in one pass, and the ranking - a mean over at
        most T' numbers per sample - is done here exactly as the reference writes it.  The 0/1 draws come from
        ``torch.randint(0, 2, (B * sample_num, T', 1))`` like the reference's (same seed, same stream);
        ``args.esa_select`` overrides them (tests)."""
        dev = torch.device("cuda", getattr(self, "_device", torch.cuda.current_device()))
        feats = src.to(dev, torch.float32).contiguous()
        ratio = src_size.to(dev, torch.float32).contiguous()
        B, T, _ = feats.shape
        S = int(args.sample_num)
        Tp = ((T - 1) // 2 + 1 - 1) // 2 + 1
        # samples per decoder pass: the decoder side runs B * group query sets over the B utterances' encoder outputs
        group = max(1, min(S, int(getattr(args, "hip_esa_group", 16))))
        eng = self.engine(B, T, esa_group=group)
        opts = hip.Engine.make_opts(args)
        opts.sos = sos
        bw = int(args.beam_width)
        opts.beam_width = 1  # the sample passes keep the best label per row (what the ranking reads); beam_width > 1: see below
        select = getattr(args, "esa_select", None)
        if select is None:
            select = torch.randint(0, 2, (B * S, Tp, 1))
        select = torch.as_tensor(select).reshape(B, S, Tp).to(torch.uint8).transpose(0, 1).contiguous()  # (S, B, T')
        select[0] = 0  # include_best: sample 0 is the best path (cassnat.py:441-445)
        select = select.to(dev)
        eng.esa_begin(feats, opts)
        stride = Tp + 2
        tok = torch.zeros(S, B, stride, dtype=torch.int32, device=dev)
        val = torch.zeros(S, B, stride, dtype=torch.float32, device=dev)
        ylen = torch.zeros(S, B, dtype=torch.int32, device=dev)
        U, force = 0, 0
        if (self._conf_dec or self._hyper.get("conf_enc")) and S > group:
            # conformer blocks: GroupNorm covers an utterance's padded rows too, so all groups decode on the row count of ALL
            # samples (what the reference's single batch has); a first, alignment-only sweep finds it
            force = max(eng.esa_sample(select[g0:min(S, g0 + group)], args.threshold, ratio, opts, None, None, None, force_U=-1)
                        for g0 in range(0, S, group))
        for g0 in range(0, S, group):
            g1 = min(S, g0 + group)
            U = max(U, eng.esa_sample(select[g0:g1], args.threshold, ratio, opts, tok[g0:g1], val[g0:g1], ylen[g0:g1], force_U=force))
        # LM input = [sos] + predictions shifted right; score of every predicted token under the causal + length mask
        tokf, ylf = tok.reshape(S * B, stride), ylen.reshape(S * B)
        ylen_h = ylen.transpose(0, 1).cpu().long()                                                # (B, S)
        rank = getattr(args, "rank_model", "lm")
        if rank == "n-gram":
            # cassnat.py:523-535: the n-gram model (kenlm) scores the predicted word pieces as text, per sample: score / tgt_len
            tok_c = tok.cpu().numpy()
            prob_sum = torch.zeros(B, S)
            for b in range(B):
                for s_i in range(S):
                    n = int(ylen_h[b, s_i])
                    pieces = [vocab.index2word[int(t)] for t in tok_c[s_i, b, :n] if int(t) != 2]
                    prob_sum[b, s_i] = lm_model.score("".join(pieces).replace("\u2581", " ").strip()) / n
        else:
            lm_in = torch.cat([torch.full((S * B, 1), sos, dtype=torch.int32, device=dev), tokf[:, : stride - 1]], 1).contiguous()
            if rank == "at_baseline":  # the autoregressive model scores the predictions teacher-forced; PROBABILITIES, not logs
                lm_score = lm_model.teacher_score(feats, lm_in, tokf.contiguous(), ylf.contiguous(), S, U, args).exp()
            else:
                lm_score = lm_model.score_tokens(lm_in, tokf.contiguous(), ylf.contiguous(), U, max_frames=max(T, 64))
            lm_score = lm_score.reshape(S, B, stride)[:, :, :U].transpose(0, 1).cpu()              # (B, S, U)
            tmask = torch.arange(U).view(1, 1, U) < ylen_h.unsqueeze(-1)
            lm_score = lm_score.masked_fill(tmask == 0, 0)
            prob_sum = lm_score.sum(-1) / (lm_score != 0).sum(-1).float()                         # cassnat.py:521-522
        pick = prob_sum.max(-1, keepdim=True)[1]
        tok_h, val_h = tok.cpu().numpy(), val.cpu().numpy()
        ylen_sel = ylen_h.gather(1, pick).squeeze(1).numpy()
        ymax = int(ylen_sel.max())
        ys = torch.ones(1, 1).fill_(sos).long()
        if bw > 1:
            return self._esa_beam_finish(eng, select, pick, args, ratio, opts, bw, force, ylen_sel, ymax, sos)
        out = []
        for b in range(B):
            s_b, n = int(pick[b, 0]), int(ylen_sel[b])
            hyp, score = [sos], 0.0
            for i in range(min(n + 1, ymax)):  # the greedy finish consumes position i while i <= ylen[b] (cassnat.py:580-637)
                if i < n:
                    hyp.append(int(tok_h[s_b, b, i]))
                    score += float(val_h[s_b, b, i])
                else:  # one row past the sample's mask: the reference reads an all-zero row there (+ 0.0, arbitrary tie token)
                    hyp.append(0)
            out.append([{"ys": ys, "score": score, "hyp": hyp}])
        return out

    def _esa_beam_finish(self, eng, select, pick, args, ratio, opts, bw, force, ylen_sel, ymax, sos):
        """ESA with beam_width > 1 (src/models/cassnat.py:556-561, 574-637): the finish loop runs on the SELECTED sample of every
        utterance, on its att_out with the rows at or past the sample's own count zeroed (:556).  One more decoder pass over the
        B selected alignments keeps the bw best labels per row (``cn_esa_sample`` with opts.beam_width = bw); a zeroed row gives bw
        candidates of + 0.0 whose labels - torch.topk of equal values - are implementation-defined in the reference: 0 here."""
        B = pick.shape[0]
        idx = pick.reshape(1, B, 1).expand(1, B, select.shape[2]).to(select.device)
        sel = select.gather(0, idx).contiguous()                                                  # (1, B, T'): utterance b's winner
        opts.beam_width = bw
        eng.esa_sample(sel, args.threshold, ratio, opts, None, None, None, force_U=force)
        opts.beam_width = 1
        tk, tv = eng.fetch("topk_idx"), eng.fetch("topk_val")                                      # (B, U, bw)
        lp = args.length_penalty
        ys = torch.ones(1, 1).fill_(sos).long()
        out = []
        for b in range(B):
            n = int(ylen_sel[b])
            beams = [{"ys": ys, "score": 0.0, "hyp": [sos]}]
            for i in range(min(n + 1, ymax)):
                if i < n:
                    cand = [{"ys": ys, "score": s["score"] + float(tv[b, i, j]), "hyp": s["hyp"] + [int(tk[b, i, j])]} for s in beams for j in range(bw)]
                else:
                    cand = [{"ys": ys, "score": s["score"] + 0.0, "hyp": s["hyp"] + [0]} for s in beams for j in range(bw)]
                if lp is not None:
                    cand.sort(key=lambda s: s["score"] + (len(s["hyp"]) - 1) * lp, reverse=True)
                else:
                    cand.sort(key=lambda s: s["score"], reverse=True)
                beams = cand[:bw]
            out.append(beams)
        return out

    @staticmethod
    def _host_beam(eng, args, sos):
        """beam_width > 1 (src/models/cassnat.py:580-636 with lm_weight == 0): the per-position top-k comes from
        the device, the O(U * beam^2) bookkeeping is plain Python on B*U*k numbers."""
        idx, val, ylen = eng.fetch("topk_idx"), eng.fetch("topk_val"), eng.fetch("ylen")
        B, U, k = idx.shape
        ys = torch.ones(1, 1).fill_(sos).long()
        lp = args.length_penalty
        out = []
        for b in range(B):
            beams = [{"ys": ys, "score": 0.0, "hyp": [sos]}]
            for i in range(min(int(ylen[b]) + 1, U)):
                cand = [{"ys": ys, "score": s["score"] + float(val[b, i, j]), "hyp": s["hyp"] + [int(idx[b, i, j])]}
                        for s in beams for j in range(k)]
                if lp is not None:
                    cand.sort(key=lambda s: s["score"] + (len(s["hyp"]) - 1) * lp, reverse=True)
                else:
                    cand.sort(key=lambda s: s["score"], reverse=True)
                beams = cand[:k]
            out.append(beams)
        return out


def make_model(input_size, args):
    """Same role as src/models/cassnat.py:21-89: transformer blocks, or conformer blocks with use_conv_enc / use_conv_dec."""
    if not getattr(args, "use_conv_enc", False):
        assert args.model_type == "transformer"
    model = CassNAT(input_size, args)
    for name, p in model.named_parameters():  # src/models/cassnat.py:86-88 (the frozen position tables stay as they are)
        if p.dim() > 1 and p.requires_grad:
            nn.init.xavier_uniform_(p)
    return model
