"""ORACLE - test infrastructure only.  CPU restatement (torch-CPU fp32 ATen ops) of the conformer variant of CASS-NAT
greedy inference: `use_conv_enc` / `use_conv_dec`, `pos_type == "relative"` - what the shipped YAMLs
configure (egs/librispeech/conf/cassnat_decode.yaml:17-25).  Pinned by tests/test_oracle_golden.py against fixtures
produced by running the reference's CassNAT.beam_decode (oracle/make_goldens.py).  Citations: file:line under /root/reference/.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .cassnat_oracle import (_t, align_to_intervals, best_path_align, beam_finish, generator, greedy_finish, key_mask_from_feats,
                             layer_norm, linear, multi_head_attention, sinusoid_table, src_size_frames, target_mask,
                             to_torch_state)

FLOAT_MIN = float(np.finfo(np.float32).min)


def swish(x):
    """src/models/modules/conformer_related.py:10-12."""
    return x * torch.sigmoid(x)


def rel_pos_embed(d_model, t, max_rel):
    """RelativePositionalEncoding.forward, src/models/modules/embedding.py:33-60: rows for distances -(t-1) .. t-1, clamped
    to +-max_rel, taken from the sinusoid table of 2*max_rel+1 positions.  -> (2t-1, d)"""
    table = sinusoid_table(d_model, 2 * max_rel + 1)
    idx = torch.clamp(torch.arange(-(t - 1), t), -max_rel, max_rel) + max_rel
    return table[idx]


def rel_multi_head_attention(st, prefix, x, mask, pos_embed, n_head):
    """RelMultiHeadedAttention.forward (self attention: query = key = value = x), src/models/modules/attention.py:68-147.

    The zero-pad / view "shift" (:124-128) selects, for query i and key j, the column j - i + t_q - 1 of scores_bd.
    mask (B, 1 or t_q, t_k) bool; masked scores are filled with float32-min and the probabilities re-masked to 0 (:133-134).
    """
    B, T, d = x.shape
    dk = d // n_head
    q = linear(st, prefix + ".linears.0", x).view(B, T, n_head, dk)
    k = linear(st, prefix + ".linears.1", x).view(B, T, n_head, dk).transpose(1, 2)
    v = linear(st, prefix + ".linears.2", x).view(B, T, n_head, dk).transpose(1, 2)
    p = F.linear(pos_embed, st[prefix + ".linear_pos.weight"]).view(-1, n_head, dk).transpose(0, 1)  # (h, 2T-1, dk)
    qu = (q + st[prefix + ".pos_bias_u"]).transpose(1, 2)
    qv = (q + st[prefix + ".pos_bias_v"]).transpose(1, 2)
    ac = torch.matmul(qu, k.transpose(-2, -1))
    bd = torch.matmul(qv, p.transpose(-2, -1).unsqueeze(0))  # (B, h, T, 2T-1)
    zero = torch.zeros(B, n_head, T, 1)
    padded = torch.cat([zero, bd], dim=-1).view(B, n_head, 2 * T, T)
    bd = padded[:, :, 1:].reshape(B, n_head, T, 2 * T - 1)[:, :, :, :T]
    scores = (ac + bd) / math.sqrt(dk)
    m = mask.unsqueeze(1)
    scores = scores.masked_fill(m == 0, FLOAT_MIN)
    pa = F.softmax(scores, dim=-1).masked_fill(m == 0, 0.0)
    ctx = torch.matmul(pa, v).transpose(1, 2).contiguous().view(B, T, d)
    return linear(st, prefix + ".linears.3", ctx)


def conv_module(st, prefix, x):
    """ConvModule.forward, src/models/modules/conformer_related.py:15-44: pointwise conv + GLU, depthwise conv, GroupNorm(1, C)
    over (C, T) of each utterance - padded frames included -, Swish, pointwise conv."""
    y = x.transpose(1, 2)
    y = F.glu(F.conv1d(y, st[prefix + ".pointwise_conv1.weight"], st[prefix + ".pointwise_conv1.bias"]), dim=1)
    w = st[prefix + ".depthwise_conv.weight"]
    y = F.conv1d(y, w, st[prefix + ".depthwise_conv.bias"], padding=(w.size(2) - 1) // 2, groups=w.size(0))
    y = swish(F.group_norm(y, 1, st[prefix + ".norm.weight"], st[prefix + ".norm.bias"]))
    y = F.conv1d(y, st[prefix + ".pointwise_conv2.weight"], st[prefix + ".pointwise_conv2.bias"])
    return y.transpose(1, 2)


def feed_forward_swish(st, prefix, x):
    """PositionwiseFeedForward with activation=Swish (src/models/cassnat.py:35, positionff.py:15-16)."""
    return linear(st, prefix + ".w_2", swish(linear(st, prefix + ".w_1", x)))


def _sub(st, prefix, i, x, fn, scale=1.0):
    """SublayerConnection (src/models/modules/utils.py:23-32): x + scale * fn(LN(x))."""
    y = layer_norm(x, st[f"{prefix}.sublayer.{i}.norm.a_2"], st[f"{prefix}.sublayer.{i}.norm.b_2"])
    return x + scale * fn(y)


def conformer_self_layer(st, p, x, mask, pos_embed, n_head, ff_scale=0.5):
    """SelfAttLayer.forward, relative branch (src/models/blocks/fanat_conformer_blocks.py:26-38)."""
    x = _sub(st, p, 0, x, lambda y: feed_forward_swish(st, p + ".feed_forward1", y), ff_scale)
    x = _sub(st, p, 2, x, lambda y: rel_multi_head_attention(st, p + ".self_attn", y, mask, pos_embed, n_head))
    x = _sub(st, p, 1, x, lambda y: conv_module(st, p + ".conv_module", y))
    x = _sub(st, p, 3, x, lambda y: feed_forward_swish(st, p + ".feed_forward2", y), ff_scale)
    return x


def conformer_mix_layer(st, p, x, memory, src_mask, self_mask, pos_embed, n_head, ff_scale=0.5):
    """MixAttLayer.forward, relative branch (fanat_conformer_blocks.py:85-97)."""
    x = _sub(st, p, 0, x, lambda y: feed_forward_swish(st, p + ".feed_forward1", y), ff_scale)
    x = _sub(st, p, 2, x, lambda y: rel_multi_head_attention(st, p + ".self_attn", y, self_mask, pos_embed, n_head))
    x = _sub(st, p, 1, x, lambda y: conv_module(st, p + ".conv_module", y))
    x = _sub(st, p, 3, x, lambda y: multi_head_attention(st, p + ".src_attn", y, memory, memory, src_mask, n_head))
    x = _sub(st, p, 4, x, lambda y: feed_forward_swish(st, p + ".feed_forward2", y), ff_scale)
    return x


def conv_embed_rel(st, feats, x_mask, max_rel):
    """ConvEmbedding.forward with RelativePositionalEncoding (embedding.py:112-124, 48-58): x * sqrt(d), NO additive PE."""
    d = st["src_embed.conv.0.bias"].numel()
    c1 = F.relu(F.conv2d(feats.unsqueeze(1), st["src_embed.conv.0.weight"], st["src_embed.conv.0.bias"], stride=2, padding=1))
    c2 = F.relu(F.conv2d(c1, st["src_embed.conv.2.weight"], st["src_embed.conv.2.bias"], stride=2, padding=1))
    b, c, t, f = c2.size()
    y = linear(st, "src_embed.linear_out", c2.transpose(1, 2).contiguous().view(b, t, c * f)) * math.sqrt(d)
    return y, x_mask[:, :, ::2][:, :, ::2], rel_pos_embed(d, t, max_rel)


def decode_nast_conformer(state, feats, size_ratio, args, stages=False):
    """CassNAT.beam_decode (src/models/cassnat.py:420-637) for use_conv_enc / use_conv_dec models, greedy NAST configuration
    (use_trigger, sample_num <= 1, decode_type 'att_only', lm_weight 0, no use_unimask: the reference's tuple plumbing does
    not support it with the conformer decoder, cassnat.py:486-488)."""
    st = state if isinstance(next(iter(state.values())), torch.Tensor) else to_torch_state(state)
    feats = _t(feats).float()
    out = {}
    H = args.n_head
    with torch.no_grad():
        x_mask = key_mask_from_feats(feats, args.padding_idx)
        if args.use_conv_enc:
            x, src_mask, pos = conv_embed_rel(st, feats, x_mask, args.enc_max_relative_len)
            layers = []
            for n in range(args.N_enc):  # Encoder.forward, fanat_conformer_blocks.py:150-170
                x = conformer_self_layer(st, f"encoder.layers.{n}", x, src_mask, pos, H)
                layers.append(x)
            x_embed = layers and None
            enc_h = layer_norm(x, st["encoder.norm.a_2"], st["encoder.norm.b_2"])
        else:
            from .cassnat_oracle import conv_embed, encoder
            x, src_mask, _, _ = conv_embed(st, feats, x_mask)
            layers = []
            enc_h = encoder(st, x, src_mask, args.N_enc, H, collect=layers)
        ctc_out = generator(st, "ctc_generator", enc_h)
        t_sub = ctc_out.size(1)
        src_size = src_size_frames(size_ratio, t_sub)
        km = src_mask.squeeze(1).numpy()
        best_paths = ctc_out.argmax(-1).numpy()
        shift, ylen0, ymax0 = best_path_align(best_paths, km, args.padding_idx)
        trig, ylen, ymax = align_to_intervals(shift, ylen0, ymax0, km, src_size, args.padding_idx, args.left_trigger, args.right_trigger)
        tmask_t, trigger_t = torch.from_numpy(target_mask(ylen, ymax)), torch.from_numpy(trig)
        d = enc_h.size(-1)
        queries = sinusoid_table(d)[:ymax].unsqueeze(0).repeat(feats.size(0), 1, 1)
        cross_mask = trigger_t if args.src_trigger else src_mask
        if args.use_conv_dec:
            # ConAcExtra = SrcAttLayer (fanat_conformer_blocks.py:41-60): attention output WITHOUT residual or pre-norm, scaled by
            # sqrt(d), relative position rows for the token axis, then x + FFN(LN x) with the d_ff-wide Swish FFN
            assert args.N_extra == 1
            p = "acembed_extractor.layers.0"
            ac = multi_head_attention(st, p + ".src_attn", queries, enc_h, enc_h, trigger_t, H) * math.sqrt(d)
            dpos = rel_pos_embed(d, ymax, args.dec_max_relative_len)
            ac = ac + feed_forward_swish(st, p + ".feed_forward",
                                         layer_norm(ac, st[p + ".sublayer.norm.a_2"], st[p + ".sublayer.norm.b_2"]))
            pred = ac
            for n in range(args.N_self_dec):  # SelfAttDecoder, no final norm (:188-208)
                pred = conformer_self_layer(st, f"embed_mapper.layers.{n}", pred, tmask_t, dpos, H)
            x = pred
            for n in range(args.N_mix_dec):   # MixAttDecoder + final norm (:219-243)
                x = conformer_mix_layer(st, f"decoder.layers.{n}", x, enc_h, cross_mask, tmask_t, dpos, H)
            dec_h = layer_norm(x, st["decoder.norm.a_2"], st["decoder.norm.b_2"])
        else:
            from .cassnat_oracle import acoustic_extractor, mix_att_decoder, self_att_decoder
            ac = acoustic_extractor(st, args.N_extra, H, queries, enc_h, trigger_t)
            pred = self_att_decoder(st, args.N_self_dec, H, ac, tmask_t)
            dec_h = mix_att_decoder(st, args.N_mix_dec, H, pred, enc_h, cross_mask, tmask_t)
        att_out = generator(st, "att_generator", dec_h)
        if args.beam_width == 1:
            hyps, scores = greedy_finish(att_out, ylen, ymax)
        else:
            beams = beam_finish(att_out, ylen, ymax, args.beam_width, args.length_penalty)
            hyps, scores = [b[0]["hyp"] for b in beams], [b[0]["score"] for b in beams]
    out.update(hyps=hyps, scores=scores, ylen=ylen, ymax=ymax, src_size=src_size, aligned_seq_shift=shift, best_paths=best_paths)
    if stages:
        out.update(enc_layers=layers, enc_h=enc_h, ctc_out=ctc_out, trigger=trig, ac_embed=ac, pred_embed=pred, dec_h=dec_h,
                   att_out=att_out, src_mask=src_mask)
    return out


def decode_nast_esa_conformer(state, lm_state, feats, size_ratio, args, lm_args, select, sos=1):
    """ESA decoding (cassnat_oracle.decode_nast_esa: src/models/cassnat.py:370-376, 441-445, 499-561) with the conformer blocks
    of this file in place of the transformer ones - the shipped cassnat_decode.yaml's combination (use_conv_dec + sample_num)."""
    from .cassnat_oracle import decode_nast_esa

    H = args.n_head

    def encode(st, feats_t, x_mask):
        if args.use_conv_enc:
            x, src_mask, pos = conv_embed_rel(st, feats_t, x_mask, args.enc_max_relative_len)
            for n in range(args.N_enc):
                x = conformer_self_layer(st, f"encoder.layers.{n}", x, src_mask, pos, H)
            return layer_norm(x, st["encoder.norm.a_2"], st["encoder.norm.b_2"]), src_mask
        from .cassnat_oracle import conv_embed, encoder
        x, src_mask, _, _ = conv_embed(st, feats_t, x_mask)
        return encoder(st, x, src_mask, args.N_enc, H), src_mask

    def decode(st, queries, memory, src_mask, trigger_t, tmask_t):
        assert args.use_conv_dec and args.N_extra == 1
        d, ymax = memory.size(-1), queries.size(1)
        cross_mask = trigger_t if args.src_trigger else src_mask
        p = "acembed_extractor.layers.0"
        ac = multi_head_attention(st, p + ".src_attn", queries, memory, memory, trigger_t, H) * math.sqrt(d)
        dpos = rel_pos_embed(d, ymax, args.dec_max_relative_len)
        ac = ac + feed_forward_swish(st, p + ".feed_forward", layer_norm(ac, st[p + ".sublayer.norm.a_2"], st[p + ".sublayer.norm.b_2"]))
        x = ac
        for n in range(args.N_self_dec):
            x = conformer_self_layer(st, f"embed_mapper.layers.{n}", x, tmask_t, dpos, H)
        for n in range(args.N_mix_dec):
            x = conformer_mix_layer(st, f"decoder.layers.{n}", x, memory, cross_mask, tmask_t, dpos, H)
        return layer_norm(x, st["decoder.norm.a_2"], st["decoder.norm.b_2"])

    return decode_nast_esa(state, lm_state, feats, size_ratio, args, lm_args, select, sos=sos, encode_fn=encode, decode_fn=decode)
