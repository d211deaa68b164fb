"""Frame splicing and subsampling of a (T, dim) feature matrix (reference: src/data/feat_op.py:4-31),
vectorised: one gather instead of a Python loop per context frame."""
import numpy as np


def context_feat(feat_mat, left_context, right_context):
    """Concatenate each frame with `left_context` past and `right_context` future frames; the first/last frame is
    replicated past the edges.  Column blocks are ordered oldest -> newest."""
    if left_context == 0 and right_context == 0:
        return feat_mat
    n = feat_mat.shape[0]
    offsets = np.arange(-left_context, right_context + 1)
    idx = np.clip(np.arange(n)[:, None] + offsets[None, :], 0, n - 1)
    return feat_mat[idx].reshape(n, -1)


def skip_feat(feat_mat, skip):
    """Keep every `skip`-th frame starting with frame 0."""
    if skip in (0, 1):
        return feat_mat
    return feat_mat[::skip]
