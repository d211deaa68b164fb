"""Utterance sharding across the GPUs of one node: one process per GPU, torch.distributed (RCCL over xGMI).

The reference shards decoding by splitting feats.scp into nj files and starting one process per GPU, each
loading the checkpoint from disk and writing its own result file (egs/librispeech/run_hubert.sh:94-116):
utterances are independent, there is no collective.  Here rank 0 alone reads/packs the checkpoint and the
packed weight blob is broadcast once (57 MB bf16 for the config-2 model); every batch ends with one
all-gather of fixed-size hypothesis records (B x (stride+3) int32 per rank - latency bound).  No other
collective exists on the path.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_indices(lengths, world, rank):
    """Length-sorted snake deal (0,1,..,W-1,W-1,..,1,0,0,1,..): every rank gets a similar mix of lengths and a
    similar total number of frames.  Returns this rank's indices into the global list, longest first."""
    order = np.argsort(-np.asarray(lengths), kind="stable")
    pos = np.arange(len(order))
    lap, k = pos // world, pos % world
    owner = np.where(lap % 2 == 0, k, world - 1 - k)
    return order[owner == rank]


class _CudaBlob:
    """Zero-copy view of raw device memory for torch (``__cuda_array_interface__`` v2)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def broadcast_weights(engine, src=0, group=None):
    """RCCL broadcast of the packed weight blob from `src` into every other rank's (layout-identical) blob."""
    ptr, nbytes = engine.weight_blob()
    t = torch.as_tensor(_CudaBlob(ptr, nbytes), device="cuda")
    # every rank's blob must have the sender's layout: one 16-byte all-reduce (max of n and of -n), no pickled objects
    chk = torch.tensor([nbytes, -nbytes], dtype=torch.int64, device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(chk, op=dist.ReduceOp.MAX, group=group)
    if int(chk[0]) != nbytes or int(chk[1]) != -nbytes:
        raise RuntimeError(f"weight blob layout differs across ranks: {nbytes} bytes here, {-int(chk[1])}..{int(chk[0])} over the ranks")
    dist.broadcast(t, src=src, group=group)
    torch.cuda.synchronize()
    return nbytes


def pack_records(hyp, hyp_len, score):
    """(B,S) int32, (B,) int32, (B,) float64 -> (B, S+3) int32 records [len, score_lo, score_hi, tokens...]."""
    sc = score.contiguous().view(torch.int32).view(-1, 2)
    return torch.cat([hyp_len.view(-1, 1), sc, hyp], dim=1).contiguous()


def unpack_records(rec, as_lists=True):
    """Records -> host.  ``as_lists``: ([tokens of utterance b ...], scores) as ``beam_decode`` reports them; otherwise the
    arrays themselves ``((tokens (N, S) int32, lengths (N,) int32), scores (N,) float64)`` - one device-to-host copy and no
    per-utterance Python work (what a throughput loop over many ranks' records wants)."""
    rec = rec.cpu()
    lens = rec[:, 0].numpy()
    score = rec[:, 1:3].clone(memory_format=torch.contiguous_format).view(torch.float64).view(-1).numpy()  # (clone: offset 0)
    toks = rec[:, 3:].numpy()
    if not as_lists:
        return (toks, lens), score
    return [toks[b, : lens[b]].tolist() for b in range(rec.shape[0])], score


def all_gather_records(rec, group=None):
    """One all-gather per batch: (B, S+3) -> (world*B, S+3), rank-major."""
    world = dist.get_world_size(group)
    if dist.get_backend(group) == "gloo":  # CPU rehearsal / tests: gloo has no all_gather_into_tensor
        parts = [torch.empty_like(rec) for _ in range(world)]
        dist.all_gather(parts, rec, group=group)
        return torch.cat(parts, 0)
    out = torch.empty((world * rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
    dist.all_gather_into_tensor(out, rec, group=group)
    return out
