"""Drop-in for the reference's ``utils.beam_decode.ctc_beam_decode`` (src/utils/beam_decode.py:8-93): CTC prefix beam search
over the encoder's log-posteriors - what ``decode_type: ctc_only`` returns and what ``decode_type: ctc_att`` aligns the NAT
decoder to (src/tasks/cassnat_task.py:335-341).

Same call and the same result - per utterance a best-first list of ``{'ys', 'p_blk', 'p_nblk', 'score_ctc', 'score_lm',
'hyp'}`` (``hyp`` without sos) - but the encoder, the CTC generator, the per-frame pruning and the whole frame loop run on
the device (``cn_ctc_beam``; csrc/ctc_beam.hip).  The in-loop language-model fusion of the reference (which its own comment
calls "not applicable temporarily") is outside the accelerated path: ``lm_model`` must be None.
"""
import torch

from .. import hip

logzero, logone = -1e10, 0  # src/utils/ctc_prefix.py:11-12


def ctc_beam_decode(model, src, src_mask, src_size, vocab, args, lm_model=None, engine=None):
    """``model``: a CassNAT (src/tasks/cassnat_task.py:335-341) or the autoregressive Transformer with its CTC head
    (src/tasks/art_task.py:252-253).  ``engine``: run on this handle (a decode pipeline's) instead of the model's own."""
    if lm_model is not None:
        raise NotImplementedError("CTC beam search with in-loop LM fusion is outside the accelerated path (ctc_lm_weight must be 0)")
    if args.ctc_lp is None:
        raise TypeError("ctc_lp must be a number (with None the reference's sort key is a lambda and sorted() fails)")
    sos = vocab.word2index["sos"]
    assert vocab.word2index["blank"] == args.padding_idx, "CTC blank id and padding_idx must agree"
    dev = torch.device("cuda", getattr(model, "_device", torch.cuda.current_device()))
    feats = src.to(dev, torch.float32).contiguous()
    ratio = src_size.to(dev, torch.float32).contiguous()
    B, T, _ = feats.shape
    eng = engine if engine is not None else model.engine(B, T)
    # (args.hip_capture as in CassNAT.beam_decode: a capture run keeps every stage on the kernels that leave full tensors behind,
    # so that `ctc_out` fetched after a later call of the same engine is the tensor this search ran on)
    opts = hip.Engine.make_opts(args, capture=getattr(args, "hip_capture", False))
    opts.sos = sos
    hyp, hlen, sc, pb, pnb, nb = eng.ctc_beam(feats, ratio, opts, int(args.ctc_beam), int(args.ctc_pruning), float(args.ctc_lp))
    hyp, hlen, sc, pb, pnb, nb = (t.cpu().numpy() for t in (hyp, hlen, sc, pb, pnb, nb))
    out = []
    for b in range(B):
        seqs = []
        for j in range(int(nb[b])):
            h = hyp[b, j, : hlen[b, j]].tolist()
            seqs.append({"ys": torch.tensor([[sos] + h], dtype=torch.long), "p_blk": float(pb[b, j]), "p_nblk": float(pnb[b, j]),
                         "score_ctc": float(sc[b, j]), "score_lm": 0.0, "hyp": h})
        out.append(seqs)
    return out
