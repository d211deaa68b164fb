"""Decode pipelines: keep one GPU busy with greedy CASS-NAT decoding of many batches.

The reference decodes batch after batch, each collated to its own longest utterance (src/data/speech_loader.py:327-356,
loop src/tasks/cassnat_task.py:317-356), and reads the data-dependent token count back in the middle of every batch
(src/models/cassnat.py:387 `.item()`).  A 32-utterance batch fills a fraction of an MI355X, so the throughput form of the path
is:

  * **merged engine passes**: a worker takes consecutive batches - of DIFFERENT frame counts - through ONE engine pass
    (``cn_decode_nast_merged``): every utterance's hypothesis and score are exactly those of a pass of its own batch (its frames
    past the batch's length are the convolutions' zero padding, `src_size`, the alignment's shift and the forced EOS frame use
    the batch's own T', keys past it do not exist for the softmax, the greedy finish is limited by the batch's own row count).
    A pass is bounded by the engine's workspace AREA (utterances x frames), not by a batch count: short utterances come in
    larger numbers;
  * **no mid-pass host sync**: the decoder side is launched on a PREDICTED row count (from the passes seen so far, with a
    margin); results do not depend on it as long as it covers the true count, which the worker checks when the pass has
    drained - on a miss (rare) the pass is decoded again exactly.  A worker keeps two passes in flight;
  * **N pipelines per GPU**: engine handle (a workspace; the packed weights are ONE device copy shared by all), HIP stream and
    persistent host thread each (ctypes releases the GIL inside the C call); the workers pull batches from a shared iterator -
    so feature loading and collation run in the workers too - and hand back hypothesis records in submission order.

``bench.py`` measures exactly this object; ``tasks.cassnat_task.CassNATTask.decode`` uses it for test sets.
"""
import math
import os
import queue
import threading
import time

import numpy as np
import torch

from . import dist as cdist
from . import hip as _hip


def subsampled(T):
    """T' of T frames after the two stride-2 convolutions (embedding.py:102-108)."""
    return ((T - 1) // 2 + 1 - 1) // 2 + 1


class _NoStream:
    """Stand-in for a HIP stream in the host-logic tests (no GPU)."""

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def synchronize(self):
        pass


class PackedBatch:
    """A batch the reader has NOT collated: the utterances' archive rows as they lie in the .ark - read-only float32 views (n_b, F)
    into the memory map of the archive (``data.kaldi_io.load_mat_view``) - in the batch's order.  It stands where the padded
    (B, T, F) tensor of ``SuperviseLoader.collate_fn`` (src/data/speech_loader.py:327-356) would: ``shape`` is that tensor's shape,
    ``ratios()`` its float32 length ratios.  The decode pipelines copy the rows of a whole engine pass back to back into page-locked
    memory (one straight memcpy per utterance, no padding), send them with one DMA and spread them over the padded batch on the
    device (``hip.unpack_rows``: padding and, when asked for, the global CMVN in float64 happen there)."""

    __slots__ = ("views", "lens", "shape", "dtype", "is_cuda")

    def __init__(self, views):
        self.views = views
        self.lens = [int(v.shape[0]) for v in views]
        self.shape = (len(views), max(self.lens), int(views[0].shape[1]))
        self.dtype = torch.float32
        self.is_cuda = False

    def ratios(self):
        """collate's ``ratios[b] = feat.shape[0] / t_max``: the Python (double) quotient rounded to float32"""
        t_max = self.shape[1]
        return torch.tensor([n / t_max for n in self.lens], dtype=torch.float32)

    def padded(self, pad=0.0, cmvn=None):
        """The collated tensor itself (host): what the packed path must reproduce; used by the CPU rehearsal and the tests."""
        out = np.full(self.shape, float(pad), np.float32)
        for b, v in enumerate(self.views):
            out[b, : v.shape[0]] = v if cmvn is None else ((v.astype(np.float64) - cmvn[0]) / cmvn[1]).astype(np.float32)
        return torch.from_numpy(out)


class _Job:
    """One ``records()`` call: the shared iterator and the bookkeeping its workers and its consumer meet on."""

    def __init__(self, it, args, sos, n, total=None, coalesce=1, host=False, plan=None, per_pass=None):
        self.it, self.args, self.sos = it, args, sos
        self.host = host  # the consumer wants the records on the host: the producing pipeline sends them itself, one copy per pass
        self.left = total  # batches not yet handed to a worker (None: unknown)
        # a list of known length is cut into n x rounds passes of (nearly) equal size up front - 20 batches on 3 pipelines x 3
        # per pass: 3 3 2 2 2 2 2 2 2, not 3 3 3 3 3 2 1 1 1 - so that the pipelines finish together without one-batch passes
        c = max(1, coalesce)
        self.passes_left = None if total is None else n * max(1, -(-total // (n * c)))
        self.plan = plan  # explicit pass sizes, in the order they are taken (records(plan=...); bench.py --plan)
        self.lock = threading.Lock()
        self.cv = threading.Condition()
        self.slots = {}          # index of a pass's first batch -> (tags, records per batch, event, the pass's record tensor); under cv
        self.state = {"next": 0, "done": False, "err": None, "held": None}
        # batches taken but not yet consumed (bounds the device memory held by records): two passes in flight per pipeline plus
        # as many waiting for the consumer
        self.ahead = threading.Semaphore(4 * n * max(1, per_pass or coalesce) + 2)
        self.finished = threading.Semaphore(0)       # released once by every worker when it has left the job

    def hand_out(self, first, tags, recs, ev, whole):
        """A drained pass goes to the consumer as ONE item (a queue per batch cost the end of a short job 10 us per batch: the
        passes of all pipelines drain together there, and everything after them is serial host time)."""
        with self.cv:
            self.slots[first] = (tags, recs, ev, whole)
            self.cv.notify_all()


class _Pass:
    """One engine pass in flight: what is needed to hand its results out - or to decode it again when the predicted row
    count turns out too small."""

    __slots__ = ("items", "first", "feats", "ratio", "rows", "frames", "ticket", "u_hint", "rec", "recs", "ev", "tp")


class DecodePipelines:
    def __init__(self, model, n_pipelines, batch, frames, with_weights=True, after_engine=None, coalesce=1, share_from=None,
                 ragged=0.75, predict_rows=True, area_frames=None, cmvn=None, copy_threads=0):
        """``model``: a CassNAT holding the parameters; ``batch`` x ``frames``: the largest single batch a pipeline must take.
        The pipelines of one GPU share ONE device copy of the packed weights (``cn_model_create_shared``): the first engine
        packs them - or, with ``with_weights=False`` + ``after_engine(engine)`` (multi-GPU start-up), receives them by RCCL
        broadcast (``dist.broadcast_weights``), once per rank - and the others only get a workspace of their own.
        The worker threads and their HIP streams are created once (at the first call) and live until ``close()``.

        ``coalesce`` = c > 1: a worker takes up to c consecutive batches through ONE engine pass (wider launches: width is what
        fills the chip) with every batch's hypotheses and scores exactly those of a pass of its own.  ``ragged`` = r: batches
        of different frame counts share a pass while the shortest is at least r times the longest (0 < r <= 1; 1 = equal shapes
        only); a pass holds what fits the engine's workspace AREA, ``area_frames`` utterance-frames (default: c batches of
        ``batch`` x min(frames, 1024) - a pass of short batches carries more than c of them when ``coalesce`` is given as a
        negative number -c, "by area only").  ``predict_rows``: launch the decoder side on a predicted row count instead of
        waiting for the true one in the middle of the pass (verified afterwards; a miss is decoded again).
        Transformer blocks only (a conformer's GroupNorm sees the padded rows of a merged pass): conformer models run one batch
        per pass.  ``share_from``: an engine of the same model whose device copy of the weights ALL pipelines of this object use.
        ``cmvn`` = (mean, std) float64 arrays: the batches arrive RAW (a SpeechDataset with ``device_cmvn``) and the global CMVN is
        applied on the device right behind the host-to-device copy (``hip.cmvn_``: the reference's float64 arithmetic bit for bit)
        - the loader's float64 passes over the features were three quarters of its time.
        ``copy_threads`` > 1: the archive rows of a pass of ``PackedBatch``es are copied into page-locked memory by that many host
        threads (inside ``cn_host_gather``) - what ``--load_data_workers`` means on the packed reader path."""
        self.model = model
        self._fp16 = getattr(model, "hip_precision", "") == "fp16"  # (its scores are checked for the half range: hip.check_fp16_range)
        self._guarded = getattr(model, "hip_precision", "") in ("fp16", "bf16x3")  # (engines with a feature-range guard: Engine.check_range)
        self.copy_threads = max(0, int(copy_threads))
        self._packed = [{} for _ in range(max(1, int(n_pipelines)))]  # per pipeline: packed-pass staging buffers by slot
        self.cmvn = None if cmvn is None else (np.ascontiguousarray(cmvn[0], dtype=np.float64), np.ascontiguousarray(cmvn[1], dtype=np.float64))
        self._cmvn_dev = {}
        self.n = max(1, int(n_pipelines))
        conformer = bool(getattr(model, "_conf_dec", False) or getattr(model, "_hyper", {}).get("conf_enc"))
        self.by_area = int(coalesce) < 0
        self.coalesce = 1 if conformer else max(1, abs(int(coalesce)))
        self.ragged = 1.0 if conformer else min(1.0, max(0.05, float(ragged)))
        self.predict = bool(predict_rows) and not conformer
        self.batch, self.frames = int(batch), int(frames)
        area = int(area_frames) if area_frames else self.batch * self.coalesce * min(self.frames, 1024)
        area = max(area, self.batch * self.frames)
        # the engine's workspace is max_batch x max_frames; max_frames = the longest utterance, max_batch = what gives the area
        self.max_batch = max(self.batch, -(-area // self.frames) + (1 if self.coalesce > 1 else 0))
        self.max_utts = 16 * self.max_batch  # (the library sizes its per-utterance buffers for that many)
        if share_from is not None:
            first = model.new_engine(self.max_batch, frames, share=share_from)
        else:
            first = model.new_engine(self.max_batch, frames, with_weights=with_weights)
            if after_engine is not None:
                after_engine(first)
        self.engines = [first]
        for _ in range(self.n - 1):
            self.engines.append(model.new_engine(self.max_batch, frames, share=first))
        cfg = getattr(first, "cfg", None)
        if cfg is not None:  # (what the engine was really created with: the model may round the workspace up)
            self.max_batch, self.frames_cap = int(cfg.max_batch), int(cfg.max_frames)
            self.max_utts = 16 * self.max_batch
        else:
            self.frames_cap = self.frames
        self._threads = []
        self._inbox = []
        self._busy = threading.Lock()  # one records() call at a time
        self._stage = [{} for _ in range(self.n)]  # per pipeline: merged-batch input buffers (two: two passes in flight), by kind
        self._rows = {"ratio": None}  # row-count predictor shared by the pipelines: largest tokens-per-frame ratio seen
        self._rows_lock = threading.Lock()
        self._stats_lock = threading.Lock()
        self.timeline = None  # a list: the workers and the consumer append (label, pipeline, perf_counter) - bench.py --host-timeline
        self.stats = {"passes": 0, "batches": 0, "predicted": 0, "missed": 0, "merged_ragged": 0,
                      # host seconds of the worker threads, by what they were doing (summed over the pipelines)
                      "s_take": 0.0, "s_stage": 0.0, "s_launch": 0.0, "s_retire_wait": 0.0}

    def _mark(self, label, k=-1):
        tl = self.timeline
        if tl is not None:
            tl.append((label, k, time.perf_counter()))

    def _bump(self, key, v):
        """The counters are shared by the worker threads: a read-modify-write on the dict is not atomic across them."""
        with self._stats_lock:
            self.stats[key] += v

    # ------------------------------------------------------------------------------------------ capacity
    def fits(self, rows, T):
        """Does a pass of ``rows`` utterances x ``T`` frames fit the engines' workspace?  (The library's own check, ws_check in
        csrc/model.hip, restated: every buffer scales with one of these products.)"""
        T1 = (T - 1) // 2 + 1
        Tp = (T1 - 1) // 2 + 1
        mT1 = (self.frames_cap - 1) // 2 + 1
        mTp = (mT1 - 1) // 2 + 1
        B = self.max_batch
        return (rows <= self.max_utts and rows * (Tp + 1) <= B * (mTp + 1) and rows * (T1 + 2) <= B * (mT1 + 2)
                and rows * Tp <= B * mTp)

    def _hint(self, T):
        """Predicted row count for a pass whose longest batch has T frames (0: no prediction yet - decode exactly)."""
        if not self.predict:
            return 0
        with self._rows_lock:
            r = self._rows["ratio"]
            hist = self._rows.get("hist", ())
        if r is None:
            return 0
        tp = subsampled(T)
        # margin over the largest tokens-per-frame ratio seen: extra decoder rows cost ~0.1 % of a pass each, a miss a whole pass.
        # It follows the spread of the recent passes' ratios: 8 % when they barely move (a test set of similar speech), up to 40 %
        spread = (max(hist) - min(hist)) / max(hist) if len(hist) >= 4 and max(hist) > 0 else 0.2
        margin = min(1.4, max(1.08, 1.05 + 1.5 * spread))
        return min(tp + 1, int(math.ceil(r * margin * (tp + 1))) + 4)

    def _learn(self, ymax, T):
        with self._rows_lock:
            r = ymax / float(subsampled(T) + 1)
            if self._rows["ratio"] is None or r > self._rows["ratio"]:
                self._rows["ratio"] = r
            self._rows["hist"] = (tuple(self._rows.get("hist", ())) + (r,))[-32:]

    # ------------------------------------------------------------------------------------------ workers
    def _start(self):
        if self._threads:
            return
        self._on_gpu = torch.cuda.is_available()  # (False only in the host-logic tests, which drive this class with a stub model)
        device = getattr(self.model, "_device", None)
        if device is None and self._on_gpu:
            device = torch.cuda.current_device()
        self._device = device
        self._inbox = [queue.Queue() for _ in range(self.n)]
        self._copy_streams = [None] * self.n
        ready = threading.Semaphore(0)

        def loop(k):
            if self._on_gpu:
                torch.cuda.set_device(device)
            st = torch.cuda.Stream() if self._on_gpu else _NoStream()
            self._copy_streams[k] = torch.cuda.Stream() if self._on_gpu else None  # host -> device staging of the NEXT pass
            ready.release()
            with (torch.cuda.stream(st) if self._on_gpu else st), torch.no_grad():
                while True:
                    job = self._inbox[k].get()
                    if job is None:
                        return
                    try:
                        self._work(k, st, job)
                    except BaseException as e:  # surfaces in the consumer
                        with job.lock:
                            job.state["err"] = e
                        with job.cv:
                            job.cv.notify_all()
                        try:  # the failed pass may still have kernels queued against this pipeline's buffers
                            st.synchronize()
                        except BaseException:
                            pass
                    finally:
                        job.finished.release()

        self._threads = [threading.Thread(target=loop, args=(k,), daemon=True) for k in range(self.n)]
        for t in self._threads:
            t.start()
        for _ in self._threads:
            ready.acquire()

    def _take(self, job):
        """Under the job's lock: the next pass = consecutive batches that may share an engine pass.  None: nothing left."""
        state, it, ahead = job.state, job.it, job.ahead
        items = []
        if state["held"] is not None:
            items.append(state["held"])
            state["held"] = None
        else:
            try:
                items.append(next(it))
            except StopIteration:
                state["done"] = True
                return None
        # further batches ride along while the pass fits the workspace and their frame counts are close enough; anything else
        # waits for the next pass.  A list of known length goes in passes of equal size (see _Job)
        want = 64 if self.by_area else self.coalesce
        if job.plan:
            want = min(want, job.plan.pop(0))
        elif job.left is not None and not self.by_area:
            want = min(want, max(1, -(-job.left // max(1, job.passes_left))))
            job.passes_left = max(1, job.passes_left - 1)
        f0 = items[0][0]
        rows, tmax, tmin = int(f0.shape[0]), int(f0.shape[1]), int(f0.shape[1])
        while len(items) < want:
            try:
                nxt = next(it)
            except StopIteration:
                break  # (the next worker to look finds the iterator exhausted)
            f = nxt[0]
            T = int(f.shape[1])
            hi, lo = max(tmax, T), min(tmin, T)
            ok = (type(f) is type(f0) and tuple(f.shape[2:]) == tuple(f0.shape[2:]) and f.dtype == f0.dtype and nxt[1].dtype == items[0][1].dtype
                  and lo >= self.ragged * hi and self.fits(rows + int(f.shape[0]), hi))
            if ok and ahead.acquire(blocking=False):
                items.append(nxt)
                rows, tmax, tmin = rows + int(f.shape[0]), hi, lo
            else:
                state["held"] = nxt
                break
        if job.left is not None:
            job.left = max(0, job.left - len(items))
        i = state["next"]
        state["next"] += len(items)
        return i, items

    def _stage_packed(self, k, slot, items, pad):
        """A pass of ``PackedBatch``es: every utterance's archive rows go back to back into this slot's page-locked buffer (a
        straight copy out of the memory map; with ``copy_threads`` the utterances are dealt over helper threads), ONE DMA takes them
        to the device, and ``hip.unpack_rows`` spreads them over the padded merged batch - frames past an utterance's length get
        the padding value, and the global CMVN (float64, the dataset's arithmetic) is applied on the way when the pipelines have
        the statistics.  No padded batch ever exists on the host."""
        on_gpu, device = self._on_gpu, self._device
        batches = [x[0] for x in items]
        rows = sum(b.shape[0] for b in batches)
        tmax = max(b.shape[1] for b in batches)
        F = batches[0].shape[2]
        if not on_gpu:  # CPU rehearsal of the host logic: the collated tensors themselves
            feats = torch.full((rows, tmax, F), float(pad))
            o = 0
            for b in batches:
                feats[o:o + b.shape[0], : b.shape[1]] = b.padded(pad, self.cmvn)
                o += b.shape[0]
            return feats, torch.cat([x[1] for x in items], 0)
        dev_ = torch.device("cuda", device)
        views = [v for b in batches for v in b.views]
        lens = [n for b in batches for n in b.lens]
        total = sum(lens)
        bufs = self._packed[k].get(slot)
        cap = max(total, self.max_batch * self.frames_cap)
        if bufs is None or bufs["cap"] < total or bufs["F"] != F or bufs["utts"] < rows:
            utts = max(rows, self.max_utts)
            bufs = {"cap": cap, "F": F, "utts": utts,
                    "host": torch.empty(cap * F, dtype=torch.float32, pin_memory=True),
                    "dev": torch.empty(cap * F, dtype=torch.float32, device=dev_),
                    # per utterance: row offset, frames, float32 ratio (as int32 bits) - one small DMA
                    "meta_h": torch.empty(3 * utts, dtype=torch.int32, pin_memory=True),
                    "meta_d": torch.empty(3 * utts, dtype=torch.int32, device=dev_),
                    "out": torch.empty(max(rows * tmax, self.max_batch * self.frames_cap) * F, dtype=torch.float32, device=dev_)}
            self._packed[k][slot] = bufs
        if bufs["out"].numel() < rows * tmax * F:
            bufs["out"] = torch.empty(rows * tmax * F, dtype=torch.float32, device=dev_)
        offs = np.zeros(rows, np.int64)
        np.cumsum(lens[:-1], out=offs[1:])
        from . import hip

        # one GIL-free call copies the pass's rows out of the page cache (numpy's slice assignment holds the GIL: the two pipelines'
        # threads took turns); copy_threads > 1 deals the utterances over that many host threads inside the call
        hip.host_gather(bufs["host"].data_ptr(), views, max(1, self.copy_threads))
        meta = bufs["meta_h"].numpy()
        utts = bufs["utts"]
        meta[:rows] = offs
        meta[utts:utts + rows] = lens
        meta[2 * utts:2 * utts + rows] = torch.cat([x[1] for x in items], 0).numpy().view(np.int32)
        bufs["dev"][: total * F].copy_(bufs["host"][: total * F], non_blocking=True)
        bufs["meta_d"].copy_(bufs["meta_h"], non_blocking=True)
        feats = bufs["out"][: rows * tmax * F].view(rows, tmax, F)
        stats = (None, None)
        if self.cmvn is not None:
            stats = self._cmvn_dev.get(device)
            if stats is None:
                stats = self._cmvn_dev[device] = (torch.from_numpy(self.cmvn[0]).to(dev_), torch.from_numpy(self.cmvn[1]).to(dev_))
        hip.unpack_rows(bufs["dev"], bufs["meta_d"][:utts], bufs["meta_d"][utts:2 * utts], feats, pad, stats[0], stats[1])
        return feats, bufs["meta_d"][2 * utts:2 * utts + rows].view(torch.float32)

    def _stage_inputs(self, k, slot, items, pad):
        """The merged input of a pass: the batches one after the other, padded to the longest with padding frames, in a buffer
        of the pipeline's full capacity (two of them: two passes in flight), allocated at the first merged pass."""
        if isinstance(items[0][0], PackedBatch):
            return self._stage_packed(k, slot, items, pad)
        on_gpu, device = self._on_gpu, self._device
        dev_ = torch.device("cuda", device) if on_gpu else None
        f0, r0 = items[0][0], items[0][1]
        rows = sum(int(x[0].shape[0]) for x in items)
        tmax = max(int(x[0].shape[1]) for x in items)
        feat_dim = int(f0.shape[2])
        key = (slot, feat_dim, f0.dtype, r0.dtype)
        bufs = self._stage[k].get(key)
        need = rows * tmax * feat_dim
        if bufs is None or bufs[0].numel() < need or bufs[1].numel() < rows:
            cap = max(need, self.max_batch * self.frames_cap * feat_dim)
            bufs = (torch.empty(cap, dtype=f0.dtype, device=dev_), torch.empty(max(rows, self.max_utts), dtype=r0.dtype, device=dev_))
            self._stage[k][key] = bufs
        feats = bufs[0][:need].view(rows, tmax, feat_dim)
        ratio = bufs[1][:rows]
        if all(int(x[0].shape[1]) == tmax for x in items):
            torch.cat([x[0].to(dev_, non_blocking=True) if on_gpu else x[0] for x in items], 0, out=feats)  # one launch
        else:
            # a batch first lands on the device as it is (contiguous: a plain asynchronous DMA from pinned host memory), then a
            # device-side copy spreads it over the padded rows (a host tensor copied straight into the strided view is staged
            # synchronously by torch: 2 ms of blocked host per batch)
            srcs = [x[0] if (not on_gpu or x[0].is_cuda) else x[0].to(dev_, non_blocking=True) for x in items]
            feats.fill_(float(pad))
            o = 0
            for x, src in zip(items, srcs):
                nb, t = int(x[0].shape[0]), int(x[0].shape[1])
                feats[o:o + nb, :t].copy_(src, non_blocking=True)
                o += nb
        torch.cat([x[1].to(dev_, non_blocking=True) if on_gpu else x[1] for x in items], 0, out=ratio)
        if self.cmvn is not None:
            # an utterance's frame count from collate's float32 ratio len / t (exact after rounding: t is a few thousand at most)
            lens = torch.cat([(x[1].double().cpu() * int(x[0].shape[1])).round().to(torch.int32) for x in items])
            if on_gpu:
                stats = self._cmvn_dev.get(device)
                if stats is None:
                    stats = self._cmvn_dev[device] = (torch.from_numpy(self.cmvn[0]).to(dev_), torch.from_numpy(self.cmvn[1]).to(dev_))
                from . import hip

                hip.cmvn_(feats, lens.pin_memory().to(dev_, non_blocking=True), stats[0], stats[1])
            else:  # CPU rehearsal of the host logic: the same arithmetic in numpy
                fv = feats.numpy()
                for b, n in enumerate(lens.tolist()):
                    fv[b, :n] = ((fv[b, :n].astype(np.float64) - self.cmvn[0]) / self.cmvn[1]).astype(np.float32)
        return feats, ratio

    def _launch(self, k, st, job, p, exact=False):
        """Enqueue the engine pass of ``p`` (and the trip of its records to the host) on this pipeline's stream."""
        items = p.items
        p.u_hint = 0 if exact else self._hint(max(p.frames))
        single = len(items) == 1
        out = self.model.decode_device(p.feats, p.ratio, job.args, job.sos, engine=self.engines[k],
                                       sub_rows=None if single else p.rows, sub_frames=None if single else p.frames,
                                       u_hint=p.u_hint, want_ticket=True)
        hyp, hyp_len, score, p.ticket = out
        rec = cdist.pack_records(hyp, hyp_len, score)
        if self._on_gpu and job.host:  # pinned buffer from torch's caching host allocator, asynchronous copy on this pipeline's stream
            hbuf = torch.empty(rec.shape, dtype=rec.dtype, device="cpu", pin_memory=True)
            hbuf.copy_(rec, non_blocking=True)
            rec = hbuf
        p.rec = rec
        o, p.recs = 0, []
        for nb in p.rows:
            p.recs.append(rec[o:o + nb])
            o += nb
        p.ev = None
        if self._on_gpu:
            p.ev = torch.cuda.Event()
            p.ev.record(st)

    def _retire(self, k, st, job, p):
        """The pass has drained: check the predicted row count (decode again on a miss), learn from the true one, hand out."""
        if p.ev is not None:
            t_ = time.perf_counter()
            p.ev.synchronize()
            self._bump("s_retire_wait", time.perf_counter() - t_)
            self._mark("drained", k)
        if self._guarded:
            self.engines[k].check_range("DecodePipelines")  # (features beyond the engine's operand range: the pass is not handed out)
        if p.ticket is not None and p.ticket >= 0:
            ymax, used = self.engines[k].ticket(p.ticket)
            self._bump("passes", 1)
            self._bump("batches", len(p.items))
            if p.u_hint:
                self._bump("predicted", 1)
            if used < ymax:  # the prediction fell short: this pass again, exactly (its inputs are still in their staging slot)
                self._bump("missed", 1)
                self._learn(ymax, max(p.frames))
                self._launch(k, st, job, p, exact=True)
                if p.ev is not None:
                    p.ev.synchronize()
            else:
                self._learn(ymax, max(p.frames))
        job.hand_out(p.first, [item[2] for item in p.items], p.recs, p.ev, p.rec)
        self._mark("handed out", k)

    def _work(self, k, st, job):
        state, lock, ahead = job.state, job.lock, job.ahead
        on_gpu, device = self._on_gpu, self._device
        pad = float(getattr(job.args, "padding_idx", 0))
        inflight = []
        n_pass = 0
        try:
            while True:
                while not ahead.acquire(timeout=0.05):  # the consumer is behind: wait, but notice a shutdown
                    if state["done"] or state["err"] is not None:
                        break
                    if inflight:  # ... and do not sit on finished work meanwhile
                        self._retire(k, st, job, inflight.pop(0))
                t_ = time.perf_counter()
                with lock:
                    if state["done"] or state["err"] is not None:
                        break
                    got = self._take(job)
                self._bump("s_take", time.perf_counter() - t_)
                if got is None:
                    break
                self._mark("taken", k)
                p = _Pass()
                p.first, p.items = got
                p.rows = [int(x[0].shape[0]) for x in p.items]
                p.frames = [int(x[0].shape[1]) for x in p.items]
                p.ticket = None
                t_ = time.perf_counter()
                if len(p.items) == 1 and self.cmvn is None and not isinstance(p.items[0][0], PackedBatch):  # (raw / packed batches go through the staging buffer, alone or not)
                    p.feats, p.ratio = p.items[0][0], p.items[0][1]
                else:
                    cs = self._copy_streams[k]
                    if cs is not None and not p.items[0][0].is_cuda:
                        # host batches: the copies (and the padding / scatter kernels behind them) go on a stream of their own,
                        # so that this pass's inputs travel while the previous pass computes; the slot's last reader - the pass
                        # two before this one - has been retired
                        with torch.cuda.stream(cs):
                            p.feats, p.ratio = self._stage_inputs(k, n_pass & 1, p.items, pad)
                            ready_ev = torch.cuda.Event()
                            ready_ev.record(cs)
                        st.wait_event(ready_ev)
                    else:
                        p.feats, p.ratio = self._stage_inputs(k, n_pass & 1, p.items, pad)
                    if len(set(p.frames)) > 1:
                        self._bump("merged_ragged", 1)
                n_pass += 1
                t1_ = time.perf_counter()
                self._bump("s_stage", t1_ - t_)
                self._mark("staged", k)
                self._launch(k, st, job, p)
                self._bump("s_launch", time.perf_counter() - t1_)
                self._mark("launched", k)
                inflight.append(p)
                if len(inflight) >= 2:  # two passes in flight: this one's launches are queued behind the older one's kernels
                    self._retire(k, st, job, inflight.pop(0))
            while inflight and state["err"] is None:
                self._retire(k, st, job, inflight.pop(0))
        finally:
            st.synchronize()

    def close(self, join_timeout=None):
        for q in self._inbox:
            q.put(None)
        for t in self._threads:
            t.join(join_timeout)
        self._threads, self._inbox = [], []
        for e in self.engines:
            e.close()
        self.engines = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def records(self, batches, args, sos=1, host=False, plan=None):
        """``batches``: iterable of ``(feats (B,T,F), size_ratio (B,), tag)`` (host or device tensors).  Yields
        ``(tag, records)`` in the order of the iterable: ``records`` is the device tensor of ``dist.pack_records`` (per
        utterance: length, float64 score, [sos] + tokens), ready for ``dist.all_gather_records`` / ``unpack_records``.  The
        consumer's current stream is made to wait for the producing pipeline's work.  ``host=True``: the records arrive as
        host tensors instead (pinned; sent by the producing pipeline on its own stream, one copy per engine pass, and complete when
        they are yielded) - what a single-GPU consumer wants, which would otherwise pay one blocking copy per batch.
        ``plan``: explicit pass sizes (batches per pass, in the order the passes are taken)."""
        passes = self._passes(batches, args, sos, host, plan)
        try:
            for tags, recs, _ in passes:
                for tag, rec in zip(tags, recs):
                    yield tag, rec
        finally:
            passes.close()  # (an abandoned iterator still waits for the workers to leave the job)

    def _passes(self, batches, args, sos=1, host=False, plan=None):
        """The engine passes of ``records()`` in order, one item each: ``(tags, records per batch, the pass's whole record
        tensor)``."""
        self._start()
        job = _Job(iter(batches), args, sos, self.n, total=len(batches) if hasattr(batches, "__len__") else None,
                   coalesce=self.coalesce, host=host, plan=list(plan) if plan else None, per_pass=64 if self.by_area else None)
        state, lock = job.state, job.lock
        if not self._busy.acquire(timeout=60.0):
            raise RuntimeError("DecodePipelines.records(): the previous records() iterator was never finished or closed")
        try:
            self._mark("job posted")
            for q in self._inbox:
                q.put(job)
            i = 0
            try:
                while True:
                    with lock:
                        finished = state["done"] and i >= state["next"]
                        err = state["err"]
                    if err is not None:
                        raise err
                    if finished:
                        break
                    with job.cv:
                        got = job.slots.pop(i, None)
                        if got is None:
                            job.cv.wait(0.05)  # (a hand-out or a worker's error wakes it)
                            got = job.slots.pop(i, None)
                    if got is None:
                        continue
                    tags, recs, ev, whole = got
                    if ev is not None:
                        if job.host:
                            ev.synchronize()
                        else:
                            ev.wait(torch.cuda.current_stream())
                    job.ahead.release(len(tags))
                    self._mark("yield %d" % i)
                    yield tags, recs, whole
                    i += len(tags)
            finally:
                with lock:
                    state["done"] = True
                stuck = 0
                for _ in range(self.n):  # every worker has left the job (its stream is drained) before the next one starts
                    if not job.finished.acquire(timeout=300.0):
                        stuck += 1
                self._mark("workers left")
                if stuck:
                    raise RuntimeError(f"DecodePipelines: {stuck} pipeline(s) did not leave the job within 300 s (stuck inside a device call?)")
        finally:
            self._busy.release()

    def decode(self, batches, args, sos=1, gather=False, as_lists=True, gather_every=None, plan=None):
        """Hypotheses on the host, in order: yields ``(tag, hyps, scores)`` with ``hyps`` a list of token lists starting
        with ``sos`` (what ``beam_decode`` returns as ``['hyp']``), or with ``as_lists=False`` the arrays ``(tokens (N, S),
        lengths (N,))`` of ``dist.unpack_records``.  ``gather=True``: the multi-GPU path - every step's records of all ranks,
        rank-major.  The all-gather (and the copy home) is issued once per ``gather_every`` consecutive steps of equal shape
        (default: the pipelines' batches per pass; SURVEY 8e: "once per batch (or once per N batches)"), by step index - the
        same sequence of collectives on every rank whatever the pipelines' timing; every rank must decode the same number of
        steps."""
        if not gather:
            passes = self._passes(batches, args, sos, host=True, plan=plan)
            try:
                for tags, recs, whole in passes:  # one unpacking per engine pass, a batch is a slice of it
                    (toks, lens), scores = cdist.unpack_records(whole, as_lists=False)
                    if self._fp16:
                        _hip.check_fp16_range(scores, "DecodePipelines.decode")
                    o = 0
                    for tag, rec in zip(tags, recs):
                        nb = rec.shape[0]
                        if as_lists:
                            yield tag, [toks[b, : lens[b]].tolist() for b in range(o, o + nb)], scores[o:o + nb]
                        else:
                            yield tag, (toks[o:o + nb], lens[o:o + nb]), scores[o:o + nb]
                        o += nb
            finally:
                passes.close()
            return
        group = max(1, int(gather_every or self.coalesce))
        pending = []
        # every rank's records of a step must have ONE shape whatever pass the step rode in (a record's width follows the frame
        # count of its pass, which differs between ranks): they are padded to the width of the engines' longest utterance
        cap = subsampled(self.frames_cap) + 2 + 3

        def widen(r):
            return r if r.shape[1] >= cap else torch.nn.functional.pad(r, (0, cap - r.shape[1]))

        def flush():
            recs = [widen(r) for _, r in pending]
            nb, width = recs[0].shape
            everyone = cdist.all_gather_records(torch.cat(recs, 0) if len(recs) > 1 else recs[0]).cpu()  # one collective, one copy
            world = everyone.shape[0] // (len(recs) * nb)
            steps = everyone.view(world, len(recs), nb, width)
            out = [(tag, cdist.unpack_records(steps[:, j].reshape(world * nb, width), as_lists=as_lists)) for j, (tag, _) in enumerate(pending)]
            pending.clear()
            return out

        for tag, rec in self.records(batches, args, sos, plan=plan):
            if pending and rec.shape[0] != pending[0][1].shape[0]:
                for t_, (h_, s_) in flush():
                    yield t_, h_, s_
            pending.append((tag, rec))
            if len(pending) == group:
                for t_, (h_, s_) in flush():
                    yield t_, h_, s_
        if pending:
            for t_, (h_, s_) in flush():
                yield t_, h_, s_
