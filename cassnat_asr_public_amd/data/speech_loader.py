"""Test-time feature pipeline of the reference (src/data/speech_loader.py): Kaldi ark -> global CMVN -> splice /
skip -> zero-padded batch + length ratios.  Only what CassNATTask("test") uses is mirrored (SpeechDataset,
SpeechDataLoader); the training-only DynamicDataset / SSL loaders are out of scope.
"""
import functools
import os
import threading

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from . import kaldi_io
from .feat_op import context_feat, skip_feat


class SingleSet(object):
    """One {name, scp_path[, text_label]} stream -> list of (utt, ark specifier, token ids)."""

    def __init__(self, vocab, data_path, rank=0):
        self.name = data_path["name"]
        entries = kaldi_io.read_scp(data_path["scp_path"])
        if rank == 0:
            print("Reading %d lines from %s" % (len(entries), data_path["scp_path"]))
        labels = None
        if "text_label" in data_path:
            labels = {}
            unk, sos, eos = vocab.word2index["unk"], vocab.word2index["sos"], vocab.word2index["eos"]
            with open(data_path["text_label"], "r") as f:
                for line in f:
                    utt, _, text = line.strip().partition(" ")
                    labels[utt] = [sos] + [vocab.word2index.get(w, unk) for w in text.split(" ")] + [eos]
        self.items = [(utt, spec, labels[utt] if labels is not None else [1]) for utt, spec in entries]

    def get_len(self):
        return len(self.items)


class SpeechDataset(Dataset):
    def __init__(self, vocab, data_paths, args):
        self.left_context, self.right_context = args.left_ctx, args.right_ctx
        self.skip_frame = args.skip_frame
        self.use_cmvn = False
        # set (by the pipelined decoder, for the length of a decode) when the consumer applies the global CMVN itself, on the
        # device (pipeline.DecodePipelines(cmvn=...)): the fast path below then hands the features over as they are in the archive
        self.device_cmvn = False
        self.data_streams = [SingleSet(vocab, p, getattr(args, "rank", 0)) for p in data_paths]
        self._items = [it for s in self.data_streams for it in s.items]

    def _load_cmvn(self, cmvn_file):
        """Kaldi global CMVN stats: row 0 = sums (last column = frame count), row 1 = sums of squares."""
        stats = kaldi_io.load_mat(cmvn_file)
        count = stats[0, -1]
        self.mean = stats[0, :-1] / count
        self.std = np.sqrt(stats[1, :-1] / count - self.mean ** 2)
        self.use_cmvn = True
        return 0

    def __len__(self):
        return len(self._items)

    def __getitem__(self, idx):
        utt, spec, text = self._items[idx]
        if self.left_context == 0 and self.right_context == 0 and self.skip_frame <= 1:
            # The shipped configuration (no splicing, no frame skipping).  Same values as the general path below - the CMVN in
            # float64, rounded to float32 once (where the reference's collate converts: speech_loader.py:340) - without its
            # temporaries: the matrix is read in place from a memory map of the archive, the float64 intermediate lives in a
            # per-thread scratch buffer, and what is handed on is float32 (half the bytes for collate to move).
            feat = kaldi_io.load_mat_view(spec)
            if not self.use_cmvn or self.device_cmvn:
                return utt, feat, text
            assert feat.shape[1] == self.mean.shape[0]
            tmp = _scratch64(feat.shape)
            np.subtract(feat, self.mean, out=tmp)
            np.divide(tmp, self.std, out=tmp)
            return utt, tmp.astype(np.float32), text
        feat = kaldi_io.load_mat(spec)
        if self.use_cmvn:
            assert feat.shape[1] == self.mean.shape[0]
            feat = (feat - self.mean) / self.std
        rem = feat.shape[0] % self.skip_frame if self.skip_frame > 1 else 0
        if rem:
            feat = np.vstack([feat, np.zeros((self.skip_frame - rem, feat.shape[1]))])
        feat = skip_feat(context_feat(feat, self.left_context, self.right_context), self.skip_frame)
        return utt, feat, text

    def can_defer_cmvn(self):
        """The global CMVN commutes with everything this dataset does afterwards (no splicing, no frame skipping) AND the archive
        holds float32 matrices: the device form computes float((double)x - mean) / std) on the float32 rows collate hands over,
        which is the reference's arithmetic bit for bit only when those rows are the archive's own values.  A float64 (`DM`)
        archive is normalised in float64 and rounded once (here, on the host, as the reference does)."""
        if not (self.left_context == 0 and self.right_context == 0 and self.skip_frame <= 1):
            return False
        if getattr(self, "_defer_ok", None) is None:
            # one header per archive file (all matrices of a copy-feats archive share a type)
            firsts = {}
            for _, spec, _ in self._items:
                firsts.setdefault(spec.rpartition(":")[0] or spec, spec)
            self._defer_ok = all(kaldi_io.mat_dtype(spec) == np.float32 for spec in firsts.values())
        return self._defer_ok


_tls = threading.local()


def _scratch64(shape):
    """float64 scratch of at least `shape` for the calling thread (grown by doubling, reused across utterances)"""
    n = int(shape[0]) * int(shape[1])
    buf = getattr(_tls, "buf", None)
    if buf is None or buf.size < n:
        buf = np.empty(max(n, 2 * (buf.size if buf is not None else 0)), dtype=np.float64)
        _tls.buf = buf
    return buf[:n].reshape(shape)


def collate(batch, padding_idx=0):
    """list of (utt, feat (T,F), text) -> (utts, feats (B,Tmax,F) f32, texts (B,L) i64, feat ratios (B,) f32,
    text sizes (B,) i64), padded with `padding_idx` exactly as the reference's SuperviseLoader.collate_fn.
    (Measured on the GPU box, 6000 ragged utterances: copying the rows on a small thread pool, or assembling the batch in page-locked
    memory, both made the recogniser slower than this plain loop - profiles/r04c_*.)"""
    t_max = max(x[1].shape[0] for x in batch)
    l_max = max(len(x[2]) for x in batch)
    feats = torch.empty((len(batch), t_max, batch[0][1].shape[1]))  # (every element is written below: rows, then padding tails)
    fv = feats.numpy()
    texts = torch.full((len(batch), l_max), int(padding_idx), dtype=torch.long)
    ratios = torch.zeros(len(batch))
    sizes = torch.zeros(len(batch), dtype=torch.long)
    utts = []
    for b, (utt, feat, text) in enumerate(batch):
        fv[b, : feat.shape[0]] = feat  # (numpy: a float64 matrix is rounded to float32 here, as torch.Tensor(feat) does; a read-only map is fine)
        if feat.shape[0] < t_max:
            fv[b, feat.shape[0] :] = float(padding_idx)
        texts[b, : len(text)] = torch.as_tensor(text, dtype=torch.long)
        ratios[b] = feat.shape[0] / t_max
        sizes[b] = len(text) - 2
        utts.append(utt)
    return utts, feats, texts, ratios, sizes


def _one_thread_worker(_worker_id):
    """A loader worker reads and pads a few megabytes: one thread.  Left at torch's default, every worker starts an intra-op
    team as wide as the machine whose idle threads spin - on a box with a CPU quota the whole job (the process that launches
    the kernels included) is then throttled for most of every scheduling period (measured: 11 s instead of 0.4 s for 6000
    utterances with four workers)."""
    torch.set_num_threads(1)


class SpeechDataLoader(DataLoader):
    def __init__(self, dataset, batch_size, padding_idx=-1, distributed=False, shuffle=False, num_workers=0, indices=None):
        if distributed or shuffle:
            raise NotImplementedError("training-time sampling is out of scope")
        self.padding_idx = padding_idx
        order = list(range(len(dataset))) if indices is None else list(indices)
        batches = [order[i : i + batch_size] for i in range(0, len(order), batch_size)]
        # worker processes hand their batches over in shared memory, from which a host -> device copy is pathologically slow
        # (83 ms per 9-MB batch measured): the loader's pinning thread moves them into page-locked memory first
        super().__init__(dataset, batch_sampler=batches, num_workers=num_workers,
                         collate_fn=functools.partial(collate, padding_idx=padding_idx),
                         pin_memory=bool(num_workers > 0 and torch.cuda.is_available()),
                         worker_init_fn=_one_thread_worker if num_workers > 0 else None,
                         persistent_workers=bool(num_workers > 0))
