"""Seeded synthetic weights, features and decode-argument presets.

There is no trained checkpoint or audio available offline, so every parity
test and the benchmark run on weights drawn here (numpy ``default_rng`` is
bit-reproducible across machines) and on N(0,1) "post-CMVN fbank" features.

The parameter names/shapes are the checkpoint keys of the reference model
(``src/models/cassnat.py:21-89`` builds them, ``src/tasks/base_task.py:45-54``
loads them by name), so a state dict produced here loads into the reference
and into :class:`cassnat_asr_public_amd.models.cassnat.CassNAT` alike.
"""
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np

# Model presets. "config2" is BASELINE.json configs[1]; "config1" is configs[0];
# "tiny" exists so that every stage tensor fits in a small golden fixture.
PRESETS = {
    "tiny": dict(input_size=80, d_model=128, n_head=2, d_encff=256, d_decff=256, d_ff=256,
                 N_enc=2, N_extra=1, N_self_dec=1, N_mix_dec=1, vocab_size=40),
    "config1": dict(input_size=80, d_model=256, n_head=4, d_encff=2048, d_decff=2048, d_ff=2048,
                    N_enc=2, N_extra=1, N_self_dec=1, N_mix_dec=1, vocab_size=1028),
    "config2": dict(input_size=80, d_model=256, n_head=4, d_encff=2048, d_decff=2048, d_ff=2048,
                    N_enc=12, N_extra=1, N_self_dec=3, N_mix_dec=2, vocab_size=5000),
}

# Conformer variants: use_conv_enc / use_conv_dec with relative positions, what the shipped YAMLs
# configure (egs/librispeech/conf/cassnat_decode.yaml:17-25: relative, enc 20 / kernel 31, dec 8 / kernel 3).
CONFORMER = dict(use_conv_enc=True, use_conv_dec=True, pos_type="relative", share_ff=False)
PRESETS_CONF = {
    "tiny_conf": dict(PRESETS["tiny"], enc_max_relative_len=5, enc_kernel_size=7, dec_max_relative_len=3, dec_kernel_size=3,
                      **CONFORMER),
    # the shipped decode YAML's shape at d_model 256 / 4 heads (its commented small-model values), 12-layer encoder
    "conf_small": dict(input_size=80, d_model=256, n_head=4, d_encff=1024, d_decff=1024, d_ff=2048, N_enc=12, N_extra=1,
                       N_self_dec=1, N_mix_dec=6, vocab_size=1028, enc_max_relative_len=20, enc_kernel_size=31,
                       dec_max_relative_len=8, dec_kernel_size=3, **CONFORMER),
}
PRESETS.update(PRESETS_CONF)

# TransformerLM that ranks the ESA samples (src/models/lm.py; egs/librispeech/conf/lm.yaml shape family)
PRESETS_LM = {
    "tiny_lm": dict(d_model=128, n_head=2, d_ff=256, N=2, vocab_size=40, dropout=0.0),
    "lm_small": dict(d_model=256, n_head=4, d_ff=1024, N=4, vocab_size=5000, dropout=0.0),
}


def make_args_lm(preset="tiny_lm", **overrides):
    d = dict(PRESETS_LM[preset] if isinstance(preset, str) else preset)
    d.update(overrides)
    return SimpleNamespace(**d)


def param_shapes_lm(args):
    """Named parameters of the reference's TransformerLM (src/models/lm.py:16-31): text_embed.0.lut, encoder, out_generator."""
    d = args.d_model
    shapes = OrderedDict()
    shapes["text_embed.0.lut.weight"] = (args.vocab_size, d)
    for n in range(args.N):
        p = f"encoder.layers.{n}"
        for i in range(4):
            shapes[f"{p}.self_attn.linears.{i}.weight"] = (d, d)
            shapes[f"{p}.self_attn.linears.{i}.bias"] = (d,)
        shapes[p + ".feed_forward.w_1.weight"] = (args.d_ff, d)
        shapes[p + ".feed_forward.w_1.bias"] = (args.d_ff,)
        shapes[p + ".feed_forward.w_2.weight"] = (d, args.d_ff)
        shapes[p + ".feed_forward.w_2.bias"] = (d,)
        for i in range(2):
            shapes[f"{p}.sublayer.{i}.norm.a_2"] = (d,)
            shapes[f"{p}.sublayer.{i}.norm.b_2"] = (d,)
    shapes["encoder.norm.a_2"] = (d,)
    shapes["encoder.norm.b_2"] = (d,)
    shapes["out_generator.proj.weight"] = (args.vocab_size, d)
    shapes["out_generator.proj.bias"] = (args.vocab_size,)
    return shapes


PRESETS_AST = {
    "tiny_ast": dict(input_size=80, d_model=128, n_head=2, d_ff=256, d_encff=256, N_enc=2, N_dec=2, vocab_size=40),
    # BASELINE configs[3]: AST beam=10 on the config-2 encoder
    "config4": dict(input_size=80, d_model=256, n_head=4, d_ff=2048, d_encff=2048, N_enc=12, N_dec=6, vocab_size=5000),
}
AST_DECODE_DEFAULTS = dict(ctc_weight=0.3, max_decode_ratio=0.3, T=1.0, ctc_beam=15, beam_width=10, lm_weight=0,
                           length_penalty=0, ctc_alpha=1, interctc_alpha=0, interctc_layer=0, decode_type="ctc_att")

# Added to the CTC blank logit of the seeded config-2 model: with random weights the argmax is never blank
# (U ~ 195 of 250 frames); +0.9 gives 39-58 tokens per 10 s utterance, the range real LibriSpeech shows.
BENCH_BLANK_BIAS = 0.9

# Every attribute the NAST greedy path reads from the flat ``args`` bag
# (reference: src/models/cassnat.py:30-66 and :435-636); the reference has no
# defaults for the YAML-only ones, we give each an explicit one.
DECODE_DEFAULTS = dict(
    model_type="transformer", use_conv_enc=False, use_conv_dec=False, dropout=0.0,
    padding_idx=0, label_smooth=0, interctc_alpha=0, interce_alpha=0, interctc_layer=0,
    interce_layer=0, pos_type="absolute", share_ff=False,
    use_trigger=True, src_trigger=False, use_unimask=False, left_trigger=0, right_trigger=0,
    sample_num=0, threshold=0.9, test_hitrate=False, decode_type="att_only",
    save_embedding=False, use_gpu=False, lm_weight=0, beam_width=1, length_penalty=0,
    rank_model="lm", ctc_lm_weight=0, print_utt2diff=False, use_cmvn=False,
    dataset_type="SpeechDataset", left_ctx=0, right_ctx=0, skip_frame=1, n_features=80,
    filter_max=100000, filter_min=0, rank=0,
)


def make_args(preset="tiny", **overrides):
    """Flat attribute bag in the style of the reference's argparse+YAML namespace."""
    d = dict(DECODE_DEFAULTS)
    d.update(PRESETS[preset] if isinstance(preset, str) else preset)
    d.update(overrides)
    return SimpleNamespace(**d)


def make_args_ast(preset="tiny_ast", **overrides):
    d = dict(DECODE_DEFAULTS)
    d.update(AST_DECODE_DEFAULTS)
    d.update(PRESETS_AST[preset] if isinstance(preset, str) else preset)
    d.update(overrides)
    return SimpleNamespace(**d)


def param_shapes_ast(args):
    """Named parameters of the reference's AST model (src/models/transformer.py:19-37, registration order :55-60)."""
    d, V, F = args.d_model, args.vocab_size, args.input_size
    f2 = ((F - 1) // 2) // 2 + 1
    shapes = OrderedDict()

    def lin(prefix, n_out, n_in):
        shapes[prefix + ".weight"] = (n_out, n_in)
        shapes[prefix + ".bias"] = (n_out,)

    def mha(prefix):
        for i in range(4):
            lin(f"{prefix}.linears.{i}", d, d)

    def norm(prefix):
        shapes[prefix + ".a_2"] = (d,)
        shapes[prefix + ".b_2"] = (d,)

    shapes["src_embed.conv.0.weight"] = (d, 1, 3, 3)
    shapes["src_embed.conv.0.bias"] = (d,)
    shapes["src_embed.conv.2.weight"] = (d, d, 3, 3)
    shapes["src_embed.conv.2.bias"] = (d,)
    lin("src_embed.linear_out", d, d * f2)
    shapes["tgt_embed.0.lut.weight"] = (V, d)
    for n in range(args.N_enc):
        p = f"encoder.layers.{n}"
        mha(p + ".self_attn")
        lin(p + ".feed_forward.w_1", args.d_ff, d)
        lin(p + ".feed_forward.w_2", d, args.d_ff)
        norm(p + ".sublayer.0.norm")
        norm(p + ".sublayer.1.norm")
    norm("encoder.norm")
    for n in range(args.N_dec):
        p = f"decoder.layers.{n}"
        mha(p + ".self_attn")
        mha(p + ".src_attn")
        lin(p + ".feed_forward.w_1", args.d_ff, d)
        lin(p + ".feed_forward.w_2", d, args.d_ff)
        for i in range(3):
            norm(p + f".sublayer.{i}.norm")
    norm("decoder.norm")
    lin("ctc_generator.proj", V, d)
    lin("att_generator.proj", V, d)
    return shapes


def param_shapes(args):
    """Ordered ``name -> shape`` of every named parameter of the transformer NAST model.

    Order follows module registration order in the reference (src_embed, encoder,
    acembed_extractor, embed_mapper, decoder, ctc_generator, att_generator;
    src/models/cassnat.py:118-124).
    """
    d, V, F = args.d_model, args.vocab_size, args.input_size
    f2 = ((F - 1) // 2) // 2 + 1  # src/models/modules/embedding.py:108
    shapes = OrderedDict()

    def lin(prefix, n_out, n_in):
        shapes[prefix + ".weight"] = (n_out, n_in)
        shapes[prefix + ".bias"] = (n_out,)

    def mha(prefix):
        for i in range(4):
            lin(f"{prefix}.linears.{i}", d, d)

    def ffn(prefix, dff):
        lin(prefix + ".w_1", dff, d)
        lin(prefix + ".w_2", d, dff)

    def norm(prefix):
        shapes[prefix + ".a_2"] = (d,)
        shapes[prefix + ".b_2"] = (d,)

    shapes["src_embed.conv.0.weight"] = (d, 1, 3, 3)
    shapes["src_embed.conv.0.bias"] = (d,)
    shapes["src_embed.conv.2.weight"] = (d, d, 3, 3)
    shapes["src_embed.conv.2.bias"] = (d,)
    lin("src_embed.linear_out", d, d * f2)
    for n in range(args.N_enc):
        p = f"encoder.layers.{n}"
        mha(p + ".self_attn")
        ffn(p + ".feed_forward", args.d_encff)
        norm(p + ".sublayer.0.norm")
        norm(p + ".sublayer.1.norm")
    norm("encoder.norm")
    for n in range(args.N_extra):
        p = f"acembed_extractor.layers.{n}"
        mha(p + ".src_attn")
        ffn(p + ".feed_forward", args.d_decff)
        norm(p + ".sublayer.0.norm")
        norm(p + ".sublayer.1.norm")
    for n in range(args.N_self_dec):
        p = f"embed_mapper.layers.{n}"
        mha(p + ".self_attn")
        ffn(p + ".feed_forward", args.d_decff)
        norm(p + ".sublayer.0.norm")
        norm(p + ".sublayer.1.norm")
    for n in range(args.N_mix_dec):
        p = f"decoder.layers.{n}"
        mha(p + ".self_attn")
        mha(p + ".src_attn")
        ffn(p + ".feed_forward", args.d_decff)
        norm(p + ".sublayer.0.norm")
        norm(p + ".sublayer.1.norm")
        norm(p + ".sublayer.2.norm")
    norm("decoder.norm")
    lin("ctc_generator.proj", V, d)
    lin("att_generator.proj", V, d)
    return shapes


def sinusoid_rows(d_model, n):
    """Rows 0..n-1 of the reference's sinusoid table (src/models/modules/embedding.py:40-46, cassnat.py:91-99)."""
    pe = np.zeros((n, d_model), np.float32)
    pos = np.arange(n, dtype=np.float32)[:, None]
    div = np.exp(np.arange(0, d_model, 2, dtype=np.float32) * np.float32(-(np.log(10000.0) / d_model)))
    pe[:, 0::2] = np.sin(pos * div)
    pe[:, 1::2] = np.cos(pos * div)
    return pe


def param_shapes_conformer(args):
    """Named parameters of the conformer CASS-NAT variants (src/models/cassnat.py:29-57 with use_conv_enc / use_conv_dec),
    registration order of the reference (checked against the instantiated reference model in oracle/make_goldens.py)."""
    d, V, F = args.d_model, args.vocab_size, args.input_size
    f2 = ((F - 1) // 2) // 2 + 1
    dk = d // args.n_head
    shapes = OrderedDict()

    def lin(prefix, n_out, n_in):
        shapes[prefix + ".weight"] = (n_out, n_in)
        shapes[prefix + ".bias"] = (n_out,)

    def mha(prefix):
        for i in range(4):
            lin(f"{prefix}.linears.{i}", d, d)

    def relmha(prefix):
        shapes[prefix + ".pos_bias_u"] = (args.n_head, dk)
        shapes[prefix + ".pos_bias_v"] = (args.n_head, dk)
        mha(prefix)
        shapes[prefix + ".linear_pos.weight"] = (d, d)

    def ffn(prefix, dff):
        lin(prefix + ".w_1", dff, d)
        lin(prefix + ".w_2", d, dff)

    def convm(prefix, k):
        shapes[prefix + ".pointwise_conv1.weight"] = (2 * d, d, 1)
        shapes[prefix + ".pointwise_conv1.bias"] = (2 * d,)
        shapes[prefix + ".depthwise_conv.weight"] = (d, 1, k)
        shapes[prefix + ".depthwise_conv.bias"] = (d,)
        shapes[prefix + ".norm.weight"] = (d,)
        shapes[prefix + ".norm.bias"] = (d,)
        shapes[prefix + ".pointwise_conv2.weight"] = (d, d, 1)
        shapes[prefix + ".pointwise_conv2.bias"] = (d,)

    def norm(prefix):
        shapes[prefix + ".a_2"] = (d,)
        shapes[prefix + ".b_2"] = (d,)

    shapes["src_embed.conv.0.weight"] = (d, 1, 3, 3)
    shapes["src_embed.conv.0.bias"] = (d,)
    shapes["src_embed.conv.2.weight"] = (d, d, 3, 3)
    shapes["src_embed.conv.2.bias"] = (d,)
    lin("src_embed.linear_out", d, d * f2)
    if args.use_conv_enc:
        shapes["src_embed.pos_enc.embedding.weight"] = (2 * args.enc_max_relative_len + 1, d)
    for n in range(args.N_enc):
        p = f"encoder.layers.{n}"
        if args.use_conv_enc:
            relmha(p + ".self_attn")
            ffn(p + ".feed_forward1", args.d_encff)
            convm(p + ".conv_module", args.enc_kernel_size)
            ffn(p + ".feed_forward2", args.d_encff)
            for i in range(4):
                norm(p + f".sublayer.{i}.norm")
        else:
            mha(p + ".self_attn")
            ffn(p + ".feed_forward", args.d_encff)
            norm(p + ".sublayer.0.norm")
            norm(p + ".sublayer.1.norm")
    norm("encoder.norm")
    assert args.use_conv_dec, "param_shapes_conformer: the transformer decoder side is param_shapes()"
    p = "acembed_extractor.layers.0"
    mha(p + ".src_attn")
    ffn(p + ".feed_forward", args.d_ff)
    norm(p + ".sublayer.norm")
    shapes[p + ".pos_enc.embedding.weight"] = (2 * args.dec_max_relative_len + 1, d)
    for n in range(args.N_self_dec):
        p = f"embed_mapper.layers.{n}"
        relmha(p + ".self_attn")
        ffn(p + ".feed_forward1", args.d_decff)
        convm(p + ".conv_module", args.dec_kernel_size)
        ffn(p + ".feed_forward2", args.d_decff)
        for i in range(4):
            norm(p + f".sublayer.{i}.norm")
    for n in range(args.N_mix_dec):
        p = f"decoder.layers.{n}"
        mha(p + ".src_attn")
        relmha(p + ".self_attn")
        ffn(p + ".feed_forward1", args.d_decff)
        convm(p + ".conv_module", args.dec_kernel_size)
        ffn(p + ".feed_forward2", args.d_decff)
        for i in range(5):
            norm(p + f".sublayer.{i}.norm")
    norm("decoder.norm")
    lin("ctc_generator.proj", V, d)
    lin("att_generator.proj", V, d)
    return shapes


def make_state(args, seed=0, blank_bias=0.0, gain=1.0):
    """Seeded float32 state dict (numpy arrays).

    Matrices/conv kernels: Xavier-uniform, the law the reference applies to every
    parameter with dim>1 (src/models/cassnat.py:86-88).  Vectors are deliberately
    non-trivial (LayerNorm a_2 != 1, b_2 != 0, non-zero biases) so that a kernel
    that drops one is caught.  ``blank_bias`` is added to the CTC blank logit to
    obtain a realistic token count U with random weights.
    """
    rng = np.random.default_rng(seed)
    state = OrderedDict()
    if hasattr(args, "N_dec"):
        shapes = param_shapes_ast(args)
    elif hasattr(args, "N") and not hasattr(args, "N_enc"):
        shapes = param_shapes_lm(args)
    elif getattr(args, "use_conv_dec", False) or getattr(args, "use_conv_enc", False):
        shapes = param_shapes_conformer(args)
    else:
        shapes = param_shapes(args)
    for name, shape in shapes.items():
        if name.endswith("pos_enc.embedding.weight"):  # frozen sinusoid rows (embedding.py:40-47), not a free parameter
            w = sinusoid_rows(shape[1], shape[0])
        elif name.endswith(".norm.weight"):  # GroupNorm gain of the convolution module
            w = 1.0 + 0.1 * rng.uniform(-1, 1, size=shape)
        elif name.endswith(".norm.bias"):
            w = 0.1 * rng.uniform(-1, 1, size=shape)
        elif len(shape) > 1:
            receptive = int(np.prod(shape[2:])) if len(shape) > 2 else 1
            fan_in, fan_out = shape[1] * receptive, shape[0] * receptive
            bound = gain * np.sqrt(6.0 / (fan_in + fan_out))
            w = rng.uniform(-bound, bound, size=shape)
        elif name.endswith(".a_2"):
            w = 1.0 + 0.1 * rng.uniform(-1, 1, size=shape)
        elif name.endswith(".b_2"):
            w = 0.1 * rng.uniform(-1, 1, size=shape)
        else:
            w = 0.05 * rng.uniform(-1, 1, size=shape)
        state[name] = np.ascontiguousarray(w, dtype=np.float32)
    if blank_bias:
        state["ctc_generator.proj.bias"][0] += np.float32(blank_bias)
    return state


def make_feats(batch, frames, feat_dim=80, lengths=None, seed=1234):
    """N(0,1) features shaped like a collated batch (src/data/speech_loader.py:327-356).

    Returns ``(feats (B,T,F) f32, feat_sizes (B,) f32 ratio)``: padded tails are
    exactly 0.0, valid frames have a non-zero first feature (the reference derives
    the padding mask from ``feats[:,:,0] != 0``, src/tasks/cassnat_task.py:328).
    """
    rng = np.random.default_rng(seed)
    feats = rng.standard_normal((batch, frames, feat_dim)).astype(np.float32)
    col0 = feats[:, :, 0]
    col0[col0 == 0.0] = 1e-3
    if lengths is None:
        lengths = [frames] * batch
    lengths = np.asarray(lengths, dtype=np.int64)
    assert lengths.max() == frames, "the longest utterance defines the padded length"
    for b, n in enumerate(lengths):
        feats[b, n:] = 0.0
    sizes = (lengths.astype(np.float64) / float(frames)).astype(np.float32)
    return feats, sizes


def ragged_lengths(batch, frames, lo, seed=7):
    """Lengths uniform in [lo, frames], sorted descending, longest == frames."""
    rng = np.random.default_rng(seed)
    n = np.sort(rng.integers(lo, frames + 1, size=batch))[::-1].copy()
    n[0] = frames
    return n
