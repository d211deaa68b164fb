"""Decode half of the reference's CassNATTask (src/tasks/cassnat_task.py:24-50, 85-131, 307-377).

Same constructor / load_lm_model / decode surface and the same result-file format ("<utt> tok tok ...");
the model is the HIP-backed CassNAT.  When torch.distributed is initialised (one process per GPU) the utterance
list is dealt over the ranks by length, rank 0 broadcasts the packed weights over RCCL, and every batch's
hypothesis records are all-gathered so that rank 0 writes one complete, input-ordered result file - replacing the
reference's split_scp.pl + per-GPU result files + `cat | sort` (egs/librispeech/run_hubert.sh:94-116).
"""
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from .. import dist as cdist
from ..data.vocab import Vocab
from ..models import make_cassnat_model
from ..utils import util
from ..utils.beam_decode import ctc_beam_decode
from .base_task import BaseTask

# YAML-only keys the reference reads without defaults: explicit defaults here
_DEFAULTS = dict(use_gpu=True, decode_type="att_only", use_cmvn=False, dataset_type="SpeechDataset",
                 print_utt2diff=False, save_embedding=False, use_conv_enc=False, use_conv_dec=False,
                 use_trigger=True, src_trigger=False, use_unimask=False, left_trigger=0, right_trigger=0,
                 sample_num=0, threshold=0.9, test_hitrate=False, beam_width=1, length_penalty=0, lm_weight=0,
                 ctc_lm_weight=0, ctc_beam=1, ctc_pruning=0, ctc_lp=0, rank_model="lm", left_ctx=0, right_ctx=0, skip_frame=1, padding_idx=0,
                 model_type="transformer", dropout=0.0, rank=0)


def hyp_to_words(hyp, vocab, padding_idx):
    """Drop sos / padding ids, stop at the first eos (src/tasks/cassnat_task.py:346-353)."""
    sos, eos = vocab.word2index["sos"], vocab.word2index["eos"]
    words = []
    for idx in hyp:
        if idx == sos or idx == padding_idx:
            continue
        if idx == eos:
            break
        words.append(vocab.index2word[idx])
    return words


def hyps_to_words_batch(toks, lens, vocab, padding_idx, table=None):
    """`hyp_to_words` for a whole batch of hypothesis records (tokens (N, S) int32, lengths (N,)): one pass of array
    arithmetic instead of a Python loop over every token (at 40k utterances per second the per-token loop was most of the
    host's time).  Returns a list of word lists."""
    sos, eos = vocab.word2index["sos"], vocab.word2index["eos"]
    if table is None:
        table = np.array([vocab.index2word[i] for i in range(vocab.n_words)], dtype=object)
    toks, lens = np.asarray(toks), np.asarray(lens)
    pos = np.arange(toks.shape[1])[None, :]
    valid = pos < lens[:, None]
    is_eos = (toks == eos) & valid
    first_eos = np.where(is_eos.any(1), is_eos.argmax(1), toks.shape[1])
    keep = valid & (pos < first_eos[:, None]) & (toks != sos) & (toks != padding_idx)
    flat = table[toks[keep]]
    ends = np.cumsum(keep.sum(1))
    out, o = [], 0
    for e in ends.tolist():
        out.append(flat[o:e].tolist())
        o = e
    return out


class CassNATTask(BaseTask):
    def __init__(self, mode, args):
        for k, v in _DEFAULTS.items():
            if not hasattr(args, k):
                setattr(args, k, v)
        super(CassNATTask, self).__init__(args)
        if mode != "test":
            raise NotImplementedError("training is out of scope of the accelerated path")
        self.vocab = Vocab(args.vocab_file, args.rank)
        args.vocab_size = self.vocab.n_words
        for k in ("interctc_alpha", "interctc_layer", "interce_alpha", "interce_layer", "label_smooth"):
            setattr(args, k, 0)
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.set_model(args)
        self.set_test_dataloader(args, indices=self._shard(args))
        if self.rank == 0:
            self.load_test_model(args.resume_model)
        local = int(os.environ.get("LOCAL_RANK", "0")) if self.world > 1 else 0
        self.model_stats(local, False, False)
        self.lm_model = None

    def set_model(self, args):
        assert args.input_size == (args.left_ctx + args.right_ctx + 1) // args.skip_frame * args.n_features
        self.model = make_cassnat_model(args.input_size, args)

    def _lengths(self, args, need):
        """Frame counts of the utterance list: `utt2num_frames` (test_paths entry, or the file beside the scp) when there is
        one, else - when `need`ed - the matrix headers of the ark entries; zeros otherwise."""
        from ..data import kaldi_io

        scp = args.test_paths[0]["scp_path"]
        entries = kaldi_io.read_scp(scp)
        u2n = args.test_paths[0].get("utt2num_frames") or os.path.join(os.path.dirname(scp), "utt2num_frames")
        if os.path.exists(u2n):
            table = dict((ln.split()[0], int(ln.split()[1])) for ln in open(u2n) if ln.strip())
            if all(utt in table for utt, _ in entries):
                return np.array([table[utt] for utt, _ in entries])
        if need:
            return np.array([kaldi_io.mat_rows(spec) for _, spec in entries])
        return np.zeros(len(entries))

    def _shard(self, args):
        """Indices of this rank's utterances, in decoding order (None = all, file order).  Ranks get a length-sorted snake deal;
        `--hip_bucket 1` sorts a single process's list by length too (longest first: batches of similar frame counts, which pad
        little and share engine passes well - the reference's own batching is file order)."""
        self._order = None
        bucket = bool(int(getattr(args, "hip_bucket", 0)))
        if self.world == 1 and not bucket:
            return None
        lengths = self._lengths(args, need=bucket)
        if self.world == 1:
            return np.argsort(-lengths, kind="stable")
        return cdist.shard_indices(lengths, self.world, self.rank)

    def load_lm_model(self, args):
        """src/tasks/cassnat_task.py:85-125: the model that ranks ESA samples - a TransformerLM (rank_model 'lm'), the
        autoregressive baseline (rank_model 'at_baseline': models.transformer, scoring teacher-forced) or a kenlm n-gram model
        (rank_model 'n-gram', scored on the host as the reference does).  LM shallow fusion (lm_weight > 0) is outside the
        accelerated path."""
        self.lm_model = None
        if args.lm_weight > 0:
            raise NotImplementedError("LM shallow fusion (lm_weight > 0) is outside the accelerated path")
        if getattr(args, "ctc_lm_weight", 0) > 0:
            rank = getattr(args, "rank_model", "lm")
            if rank == "n-gram":
                import kenlm  # (not a dependency of this package: needed for this ranker only, as in the reference)

                self.lm_model = kenlm.Model(args.rnnlm)
                return
            if rank not in ("lm", "at_baseline"):
                raise NotImplementedError("rank_model '%s' is not on the accelerated path" % rank)
            import yaml
            from types import SimpleNamespace

            with open(args.lm_config) as f:
                lm_args = SimpleNamespace(**yaml.safe_load(f))
            lm_args.vocab_size = self.vocab.n_words
            lm_args.hip_precision = getattr(args, "hip_precision", "bf16")
            if rank == "lm":
                from ..models.lm import make_model as make_lm_model

                lm_model = make_lm_model(lm_args)
            else:
                if getattr(lm_args, "model_type", "transformer") != "transformer":
                    raise NotImplementedError("the conformer AST baseline is outside the accelerated path")
                from ..models.transformer import make_model as make_ast_model

                lm_args.interctc_alpha = 0
                lm_model = make_ast_model(args.input_size, lm_args)
            state = torch.load(args.rnnlm, map_location="cpu")["model_state"]
            with torch.no_grad():
                for name, param in lm_model.named_parameters():
                    param.copy_(state[name if name in state else "module." + name])
            self.lm_model = lm_model.cuda(self.local_rank) if hasattr(self, "local_rank") else lm_model

    def _decode_plain(self, args, results, batch_time, progress):
        """The reference's loop (src/tasks/cassnat_task.py:317-356): one beam_decode call per batch."""
        frames, i, end = 0, -1, time.time()
        with torch.no_grad():
            self.model.eval()
            for i, (utt_list, feats, labels, feat_sizes, label_sizes) in enumerate(self.test_loader):
                frames += int(feats.shape[0] * feats.shape[1])
                src_mask = (feats[:, :, 0] != args.padding_idx).unsqueeze(1)
                if args.decode_type == "ctc_only":  # src/tasks/cassnat_task.py:335-336
                    recog = ctc_beam_decode(self.model, feats, src_mask, feat_sizes, self.vocab, args, self.lm_model)
                elif args.decode_type == "ctc_att":  # :338-340
                    top = ctc_beam_decode(self.model, feats, src_mask, feat_sizes, self.vocab, args, self.lm_model)
                    recog, args = self.model.beam_decode(feats, src_mask, feat_sizes, self.vocab, args, self.lm_model, top,
                                                         labels=labels, label_sizes=label_sizes)
                else:
                    recog, args = self.model.beam_decode(feats, src_mask, feat_sizes, self.vocab, args, self.lm_model,
                                                         labels=labels, label_sizes=label_sizes)
                for utt, seqs, lab in zip(utt_list, recog, labels):
                    # utt2diff as the reference computes it (cassnat_task.py:358-360): against the PADDED label row's width
                    results[utt] = (hyp_to_words(seqs[0]["hyp"], self.vocab, args.padding_idx), len(seqs[0]["hyp"]) - len(lab))
                batch_time.update(time.time() - end)
                end = time.time()
                if i % args.print_freq == 0 and self.rank == 0:
                    progress.print(i)
        return frames, i

    def _decode_pipelined(self, args, n_pipes, results, batch_time, progress):
        """Greedy NAST decoding of a test set through N decode pipelines (pipeline.DecodePipelines: engine handle, HIP stream
        and host thread each; the workers pull - load and collate - the batches): same hypotheses in the same order, the
        GPU no longer idles across the two host syncs of a batch."""
        from ..pipeline import DecodePipelines

        self.model._check_args(args, self.lm_model)
        sos = self.vocab.word2index["sos"]
        # global CMVN on the device, behind the host-to-device copy (the reference's float64 arithmetic bit for bit: hip.cmvn_), when
        # nothing else sits between the archive and the batch: the loader then moves raw float32 rows and nothing more
        ds = getattr(self.test_loader, "dataset", None)
        # The packed reader (pipeline.PackedBatch): nothing but a copy sits between the archive and the device - the utterances'
        # rows go from the memory map of the .ark into page-locked memory as they are, a pass at a time, and padding + CMVN happen
        # on the device.  It applies to the shipped configuration (float32 archives, no splicing, no frame skipping) and replaces
        # the DataLoader altogether - its worker PROCESSES had to pickle every batch through shared memory (8.5k utt/s and nine
        # seconds to the first batch with the recipes' `--load_data_workers 4`); that flag now sets the number of copy THREADS.
        defer_ok = bool(ds is not None and hasattr(ds, "can_defer_cmvn") and ds.can_defer_cmvn() and int(getattr(args, "hip_device_cmvn", 1)))
        packed = bool(defer_ok and int(getattr(args, "hip_packed_reader", 1)) and hasattr(self.test_loader, "batch_sampler"))
        dev_cmvn = bool(defer_ok and getattr(ds, "use_cmvn", False) and (packed or getattr(self.test_loader, "num_workers", 0) == 0))
        if dev_cmvn:
            ds.device_cmvn = True
        try:
            return self._decode_pipelined_run(args, n_pipes, results, batch_time, progress, sos, (ds.mean, ds.std) if dev_cmvn else None, packed)
        finally:
            if dev_cmvn:
                ds.device_cmvn = False

    def _decode_pipelined_run(self, args, n_pipes, results, batch_time, progress, sos, cmvn, packed=False):
        from ..data import kaldi_io
        from ..pipeline import DecodePipelines, PackedBatch

        ds = getattr(self.test_loader, "dataset", None)
        if packed:
            first_len = max(kaldi_io.mat_rows(ds._items[i][1]) for i in list(self.test_loader.batch_sampler)[0])
        else:
            first_len = next(iter(self.test_loader))[1].shape[1]
        max_frames = max(getattr(args, "hip_max_frames", 4096), first_len)
        # consecutive batches share an engine pass while they fit the workspace area (hip_coalesce batches of batch_size x 1024
        # frames) and their frame counts are within hip_ragged of each other; passes are filled by area, not by a batch count
        # ... and the pipelines' engines hold their own packed weights and CMVN statistics: a changed parameter (load_state_dict,
        # an in-place edit, invalidate_engine), precision or statistics vector rebuilds them, as the plain path's engine is rebuilt
        import hashlib

        cmvn_id = None if cmvn is None else hashlib.sha1(np.ascontiguousarray(cmvn[0]).tobytes() + np.ascontiguousarray(cmvn[1]).tobytes()).hexdigest()
        key = (n_pipes, args.batch_size, max_frames, int(getattr(args, "hip_coalesce", 10)), float(getattr(args, "hip_ragged", 0.75)),
               cmvn_id, self.model.weights_key(), packed, int(getattr(args, "load_data_workers", 0)) if packed else 0)
        pipes = getattr(self, "_pipes", None)
        if pipes is None or self._pipes_key != key:  # (kept for further decode() calls on this task: engines, threads, streams)
            if pipes is not None:
                pipes.close()
            pipes = DecodePipelines(self.model, n_pipes, args.batch_size, max_frames, with_weights=(self.rank == 0),
                                    after_engine=(lambda e: cdist.broadcast_weights(e, src=0)) if self.world > 1 else None,
                                    coalesce=-max(1, key[3]), ragged=key[4], cmvn=cmvn, copy_threads=key[-1])
            self._pipes, self._pipes_key = pipes, key
        stats0 = dict(pipes.stats)
        meta, frames, i, end = {}, 0, -1, time.time()

        def batches():
            if packed:  # the loader's batches (same utterances, same order) as views into the archives' memory maps
                for j, idx in enumerate(self.test_loader.batch_sampler):
                    items = [ds._items[i] for i in idx]
                    pb = PackedBatch([kaldi_io.load_mat_view(spec) for _, spec, _ in items])
                    # (utt2diff reads the width of the PADDED label row, src/tasks/cassnat_task.py:358-360)
                    meta[j] = ([u for u, _, _ in items], [None] * len(items), pb.shape[0] * pb.shape[1], max(len(t) for _, _, t in items))
                    yield pb, pb.ratios(), j
                return
            for j, (utt_list, feats, labels, feat_sizes, label_sizes) in enumerate(self.test_loader):
                meta[j] = (utt_list, labels, int(feats.shape[0] * feats.shape[1]), int(labels.shape[1]))
                yield feats, feat_sizes, j

        table = np.array([self.vocab.index2word[k] for k in range(self.vocab.n_words)], dtype=object)
        for i, (toks, lens), _scores in pipes.decode(batches(), args, sos=sos, as_lists=False):
            utt_list, labels, nfr, lab_width = meta.pop(i)
            frames += nfr
            words = hyps_to_words_batch(toks, lens, self.vocab, args.padding_idx, table)
            for utt, w, n in zip(utt_list, words, lens.tolist()):
                results[utt] = (w, n - lab_width)
            batch_time.update(time.time() - end)
            end = time.time()
            if i % args.print_freq == 0 and self.rank == 0:
                progress.print(i)
        self.pipeline_stats = {k: pipes.stats[k] - stats0[k] for k in stats0}
        return frames, i

    def close(self, join_timeout=None):
        """Release the decode pipelines (engine workspaces, worker threads) a pipelined decode() left in place."""
        pipes = getattr(self, "_pipes", None)
        if pipes is not None:
            self._pipes = None
            pipes.close(join_timeout)

    def __del__(self):
        # at interpreter shutdown the daemon workers may already be gone or frozen: joining them there can hang the exit.
        # decode_asr closes its task explicitly; a task dropped without close() gives its threads two seconds
        import sys

        try:
            if not sys.is_finalizing():
                self.close(join_timeout=2.0)
        except Exception:
            pass

    def decode(self, args):
        batch_time = util.AverageMeter("Time", ":6.3f")
        progress = util.ProgressMeter(len(self.test_loader), batch_time)
        results = {}
        # args.hip_pipelines (default 2; 1 = the plain loop): beam search, ESA and capture runs keep the plain loop
        n_pipes = int(getattr(args, "hip_pipelines", 2))
        plain_greedy = (args.beam_width == 1 and getattr(args, "sample_num", 0) <= 1 and not getattr(args, "hip_capture", False)
                        and args.decode_type == "att_only")
        # Both branches issue the same collectives (one weight broadcast; the result gather below), and the choice is made from
        # rank-invariant data: the snake deal can leave ranks with batch counts that differ by one.
        n_batches = len(self.test_loader)
        if self.world > 1:
            counts = [None] * self.world
            dist.all_gather_object(counts, n_batches)
            n_batches = min(counts)
        if n_pipes > 1 and plain_greedy and n_batches > 1:
            frames, i = self._decode_pipelined(args, n_pipes, results, batch_time, progress)
        else:
            if self.world > 1:  # weights travel once over RCCL instead of N checkpoint reads
                lens = [b[1].shape[1] for b in [next(iter(self.test_loader))]] if len(self.test_loader) else [16]
                group = 1
                if getattr(args, "sample_num", 0) > 1:  # ESA: size the decoder-side workspace now (a later rebuild would only
                    group = max(1, min(int(args.sample_num), int(getattr(args, "hip_esa_group", 16))))  # share this blob anyway)
                eng = self.model.build_engine(args.batch_size, max(getattr(args, "hip_max_frames", 4096), max(lens)),
                                              with_weights=(self.rank == 0), esa_group=group)
                cdist.broadcast_weights(eng, src=0)
            frames, i = self._decode_plain(args, results, batch_time, progress)
        if self.rank == 0 and i >= 0:
            progress.print(i)
        if self.world > 1:
            gathered = [None] * self.world
            dist.all_gather_object(gathered, results)
            results = {k: v for part in gathered for k, v in part.items()}
        if self.rank == 0:
            order = [line.split()[0] for line in open(args.test_paths[0]["scp_path"]) if line.strip()]
            with open(args.result_file, "w") as out:
                for utt in order:
                    print(utt + " " + " ".join(results[utt][0]), flush=True, file=out)
            if args.print_utt2diff:
                with open(os.path.join(os.path.dirname(args.result_file), "utt2diff"), "w") as f:
                    for utt in order:
                        print(utt + " " + str(max(-3, min(3, results[utt][1]))), file=f)
        self.decoded_frames = frames
        return 0
