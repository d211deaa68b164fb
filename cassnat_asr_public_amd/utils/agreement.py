"""Margin-conditioned agreement of a reduced-precision engine with the fp32 reference.

The CTC alignment (src/models/cassnat.py:378-389) is an arg-max per frame: a frame can only change its label when the error on
the two best log-posteriors exceeds the reference's top-2 margin on that frame, i.e. ``margin < 2 max|d log-posterior|``.  On a
random-weight model the posteriors are nearly flat (median margin 0.03..0.08), so an absolute flip RATE says little about a trained
model, whose posteriors are peaked; what transfers is WHERE the flips sit: every one of them on a frame whose margin is within a
small multiple of the measured logit error, none on a clear-margin frame.  Used by the GPU tests, tools/fp8_accuracy.py and
bench.py (reporting only: no arithmetic of the product path lives here).
"""
import numpy as np

MARGIN_EDGES = (1e-4, 1e-3, 1e-2, 0.05, 0.2)


def flips_by_margin(best, ref_best, margin, own=None, edges=MARGIN_EDGES):
    """best / ref_best: integer labels per frame; margin: the reference's top-2 log-posterior margin per frame; own: optional
    boolean mask of the frames that count (an utterance's own frames).  Returns a JSON-friendly dict."""
    best, ref_best = np.asarray(best), np.asarray(ref_best)
    margin = np.asarray(margin, np.float32)
    own = np.ones(best.shape, bool) if own is None else np.asarray(own, bool)
    flips = (best != ref_best) & own
    n = int(own.sum())
    out = {"frames": n, "flips": int(flips.sum()), "flip_rate": float(flips.sum() / max(n, 1)),
           "max_flip_margin": float(margin[flips].max()) if flips.any() else 0.0}
    for thr in (0.05, 0.2):
        sel = own & (margin >= thr)
        out[f"frames_margin_ge_{thr}"] = int(sel.sum())
        out[f"flips_margin_ge_{thr}"] = int((flips & sel).sum())
        out[f"flip_rate_margin_ge_{thr}"] = float((flips & sel).sum() / max(int(sel.sum()), 1))
    lo, hist = 0.0, {}
    for hi in tuple(edges) + (np.inf,):
        sel = own & (margin >= lo) & (margin < hi)
        hist[f"[{lo:g},{hi:g})"] = [int((flips & sel).sum()), int(sel.sum())]
        lo = hi
    out["flips_frames_by_margin"] = hist
    return out


def assert_flips_explained(rep, logit_err, k, what=""):
    """No flip on a frame whose margin exceeds k x the measured logit error (a flip needs the two best logits to move by at
    least the margin between them: k = 2 is the exact bound when logit_err is the maximum over ALL logits; the fixtures hold a
    strided sample of the logits, so the tests use a larger k and say so)."""
    assert rep["max_flip_margin"] <= k * logit_err, (
        f"{what}: an arg-max flip sits on a frame of margin {rep['max_flip_margin']:.4g} > {k} x logit error {logit_err:.4g} - "
        f"a clear-margin frame flipped: {rep}")
