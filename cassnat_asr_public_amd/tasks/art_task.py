"""Decode half of the reference's ArtTask (src/tasks/art_task.py:23-66, 233-277): the autoregressive transformer
(BASELINE config 4) behind ``decode_asr.py --task art``.

Same constructor / decode surface and result-file lines ("<utt> tok tok ...").  All three decode types of the reference
(art_task.py:252-259): ``ctc_att`` (joint CTC/attention beam search, src/models/transformer.py:122-241), ``ctc_only`` (CTC prefix
beam search on the encoder, utils.beam_decode.ctc_beam_decode) and ``ctc_correct`` (the decoder as a correction model over the CTC
greedy hypothesis, transformer.py:243-342); LM fusion and the conformer AST raise.  A decode step keeps a handful of CUs busy, so a test set goes through
``args.hip_pipelines`` (default 4) engine handles (own workspace + KV cache, ONE shared device copy of the weights) on their
own HIP streams and host threads - batches pulled from the loader by the workers, result lines written in input order.
With torch.distributed initialised (one process per GPU) the utterances are dealt over the ranks by length exactly as
``CassNATTask`` does, rank 0 reads the checkpoint and broadcasts the packed weights over RCCL, and rank 0 writes the one
input-ordered result file - the reference fans out with split_scp.pl and one process + checkpoint read per GPU
(egs/librispeech/run_art.sh:115-135).
"""
import os
import threading
import time

import numpy as np
import torch
import torch.distributed as dist

from .. import dist as cdist
from ..data.vocab import Vocab
from ..models import make_transformer
from ..utils import util
from ..utils.beam_decode import ctc_beam_decode
from .base_task import BaseTask
from .cassnat_task import hyp_to_words

_DEFAULTS = dict(use_gpu=True, decode_type="ctc_att", use_cmvn=False, dataset_type="SpeechDataset", beam_width=10, ctc_beam=15,
                 ctc_pruning=0, ctc_lp=0, ctc_lm_weight=0, ctc_weight=0.3, max_decode_ratio=0, T=1.0, length_penalty=None, lm_weight=0, left_ctx=0, right_ctx=0,
                 skip_frame=1, padding_idx=0, model_type="transformer", dropout=0.0, rank=0)


class ArtTask(BaseTask):
    def __init__(self, mode, args):
        for k, v in _DEFAULTS.items():
            if not hasattr(args, k):
                setattr(args, k, v)
        super(ArtTask, self).__init__(args)
        if mode != "test":
            raise NotImplementedError("training is out of scope of the accelerated path")
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank() if self.world > 1 else 0
        self.vocab = Vocab(args.vocab_file, args.rank)
        args.vocab_size = self.vocab.n_words
        args.rank = 0
        for k in ("ctc_alpha", "interctc_alpha", "interctc_layer", "label_smooth"):
            setattr(args, k, 0)
        self.set_model(args)
        self.set_test_dataloader(args, indices=self._shard(args))
        if self.rank == 0:
            self.load_test_model(args.resume_model)
        local = int(os.environ.get("LOCAL_RANK", "0")) if self.world > 1 else 0
        self.model_stats(local, False, False)
        self.lm_model = None

    def _shard(self, args):
        """Indices of this rank's utterances (None = all): the length-sorted snake deal of CassNATTask."""
        if self.world == 1:
            return None
        n = sum(1 for _ in open(args.test_paths[0]["scp_path"]))
        lengths = np.zeros(n)
        u2n = args.test_paths[0].get("utt2num_frames")
        if u2n:
            lengths = np.array([int(line.split()[1]) for line in open(u2n)])
        return cdist.shard_indices(lengths, self.world, self.rank)

    def set_model(self, args):
        assert args.input_size == (args.left_ctx + args.right_ctx + 1) // args.skip_frame * args.n_features
        if args.model_type != "transformer":
            raise NotImplementedError("only model_type 'transformer' is on the accelerated AST path")
        self.model = make_transformer(args.input_size, args)

    def load_lm_model(self, args):
        if getattr(args, "lm_weight", 0) > 0:
            raise NotImplementedError("LM shallow fusion (lm_weight > 0) is outside the accelerated path")
        self.lm_model = None

    def _engines(self, n, args):
        """n engine handles on ONE device copy of the weights; with several ranks that copy is rank 0's, broadcast once."""
        first = next(iter(self.test_loader))[1] if len(self.test_loader) else None
        frames = max(getattr(args, "hip_max_frames", 4096), first.shape[1] if first is not None else 16)
        eng0 = self.model.build_engine(args.batch_size, frames, with_weights=(self.rank == 0))
        if self.world > 1:
            cdist.broadcast_weights(eng0, src=0)
        return [eng0] + [self.model.new_engine(args.batch_size, frames, share=eng0) for _ in range(n - 1)]

    def decode(self, args):
        # src/tasks/art_task.py:252-259: ctc_only = CTC prefix beam search on the encoder, ctc_correct = the decoder as a
        # correction model over the CTC greedy hypothesis, ctc_att = joint CTC / attention beam search
        if args.decode_type not in ("ctc_att", "ctc_only", "ctc_correct"):
            raise NotImplementedError("decode_type '%s' (ArtTask knows ctc_only, ctc_correct, ctc_att)" % args.decode_type)
        self._args = args
        batch_time = util.AverageMeter("Time", ":6.3f")
        progress = util.ProgressMeter(len(self.test_loader), batch_time)
        n = max(1, min(int(getattr(args, "hip_pipelines", 4)), max(1, len(self.test_loader))))
        engines = self._engines(n, args)
        it = iter(enumerate(self.test_loader))
        lock = threading.Lock()
        done = {}
        cv = threading.Condition()
        err = []

        def worker(k):
            try:
                torch.cuda.set_device(getattr(self.model, "_device", 0))
                st = torch.cuda.Stream()
                with torch.cuda.stream(st), torch.no_grad():
                    while not err:
                        with lock:
                            try:
                                i, (utt_list, feats, _, feat_sizes, _) = next(it)
                            except StopIteration:
                                break
                        src_mask = (feats[:, :, 0] != args.padding_idx).unsqueeze(1)
                        if args.decode_type == "ctc_only":
                            recog = ctc_beam_decode(self.model, feats, src_mask, feat_sizes, self.vocab, args, self.lm_model, engine=engines[k])
                        elif args.decode_type == "ctc_correct":
                            recog = self.model.fast_decode_with_ctc(feats, src_mask, self.vocab, args, self.lm_model, engine=engines[k])
                        else:
                            recog = self.model.beam_decode(feats, src_mask, self.vocab, args, self.lm_model, engine=engines[k])
                        lines = [utt + " " + " ".join(hyp_to_words(seqs[0]["hyp"], self.vocab, args.padding_idx))
                                 for utt, seqs in zip(utt_list, recog)]
                        with cv:
                            done[i] = lines
                            cv.notify_all()
            except BaseException as e:
                err.append(e)
                with cv:
                    cv.notify_all()

        threads = [threading.Thread(target=worker, args=(k,), daemon=True) for k in range(n)]
        end = time.time()
        for t in threads:
            t.start()
        i = -1
        results = {}
        for i in range(len(self.test_loader)):
            with cv:
                while i not in done and not err:
                    cv.wait(0.05)
                if err:
                    raise err[0]
                lines = done.pop(i)
            for line in lines:
                results[line.split(" ", 1)[0]] = line
            batch_time.update(time.time() - end)
            end = time.time()
            if i % args.print_freq == 0 and self.rank == 0:
                progress.print(i)
        for t in threads:
            t.join()
        for e in engines[1:]:
            e.close()
        if i >= 0 and self.rank == 0:
            progress.print(i)
        if self.world > 1:
            gathered = [None] * self.world
            dist.all_gather_object(gathered, results)
            results = {k: v for part in gathered for k, v in part.items()}
        if self.rank == 0:
            order = [line.split()[0] for line in open(args.test_paths[0]["scp_path"]) if line.strip()]
            with open(args.result_file, "w") as out_file:
                for utt in order:
                    print(results[utt], flush=True, file=out_file)
        return 0
