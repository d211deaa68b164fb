"""Decode pipelines: keep one GPU busy with greedy CASS-NAT decoding of many batches.

One ``beam_decode`` call needs the host twice - the token count U of the batch is data dependent (the decoder side is
launched after it is read back) and the hypotheses themselves have to reach the host - and most of its kernels occupy a
fraction of the chip (a row-chain launch of a 32-utterance batch sits on 63 of 256 CUs).  So the throughput form of the path
is N independent pipelines per GPU: each owns an engine handle (a workspace; the packed weights are ONE device copy shared
by all of them), a HIP stream and a persistent host thread (ctypes releases the GIL inside the C call), pulls the next batch from a shared iterator - so
feature loading and collation run in the workers too - and hands back device-resident hypothesis records in submission
order.  ``bench.py`` measures exactly this object; ``tasks.cassnat_task.CassNATTask.decode`` uses it for test sets.

The reference has no counterpart (it decodes batch after batch, src/tasks/cassnat_task.py:317-356); results are the same
hypotheses in the same order.
"""
import os
import queue
import threading

import torch

from . import dist as cdist


class _NoStream:
    """Stand-in for a HIP stream in the host-logic tests (no GPU)."""

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def synchronize(self):
        pass


class _Job:
    """One ``records()`` call: the shared iterator and the bookkeeping its workers and its consumer meet on."""

    def __init__(self, it, args, sos, n, total=None, coalesce=1, host=False):
        self.it, self.args, self.sos = it, args, sos
        self.host = host  # the consumer wants the records on the host: the producing pipeline sends them itself, one copy per pass
        self.left = total  # batches not yet handed to a worker (None: unknown)
        # a list of known length is cut into n x rounds passes of (nearly) equal size up front - 20 batches on 3 pipelines x 3
        # per pass: 3 3 2 2 2 2 2 2 2, not 3 3 3 3 3 2 1 1 1 - so that the pipelines finish together without one-batch passes
        c = max(1, coalesce)
        self.passes_left = None if total is None else n * max(1, -(-total // (n * c)))
        # experiment hook (tools/scripts/r02_plan.sh): CASSNAT_PASS_PLAN="10,4,6" = explicit pass sizes, in the order they are taken
        plan = os.environ.get("CASSNAT_PASS_PLAN")
        self.plan = [int(x) for x in plan.split(",")] if plan else None
        self.lock = threading.Lock()
        self.cv = threading.Condition()
        self.slots = {}          # index -> queue of one (tag, records, event)
        self.state = {"next": 0, "done": False, "err": None, "held": None}
        self.ahead = threading.Semaphore(2 * n * max(1, coalesce) + 2)  # batches decoded but not yet consumed (bounds device memory held by records)
        self.finished = threading.Semaphore(0)       # released once by every worker when it has left the job

    def slot(self, i):
        with self.cv:
            if i not in self.slots:
                self.slots[i] = queue.Queue(1)
            return self.slots[i]


class DecodePipelines:
    def __init__(self, model, n_pipelines, batch, frames, with_weights=True, after_engine=None, coalesce=1, share_from=None):
        """``model``: a CassNAT holding the parameters; ``batch`` / ``frames``: workspace size of every pipeline.
        The pipelines of one GPU share ONE device copy of the packed weights (``cn_model_create_shared``): the first engine
        packs them - or, with ``with_weights=False`` + ``after_engine(engine)`` (multi-GPU start-up), receives them by RCCL
        broadcast (``dist.broadcast_weights``), once per rank - and the others only get a workspace of their own.
        The worker threads and their HIP streams are created once (at the first call) and live until ``close()``: a short
        run is not thread start-up.
        ``coalesce`` = c > 1: a worker takes up to c consecutive batches of the same shape through ONE engine pass (wider
        launches: the command processor keeps only about three kernels in flight, so width is what fills the chip) with the
        greedy finish limited per original batch (``cn_decode_opts.sub_batch``) - every batch's hypotheses and scores are
        exactly those of a pass of its own.  When the number of batches is known (``len(batches)``) the last passes are
        cut so that every pipeline gets a similar share of the tail.  Transformer blocks only (a conformer's GroupNorm sees
        the padded rows of the merged batch).
        ``share_from``: an engine of the same model whose device copy of the weights ALL pipelines of this object use."""
        self.model = model
        self.n = max(1, int(n_pipelines))
        self.coalesce = max(1, int(coalesce)) if (not getattr(model, "_conf_dec", False)
                                                  and not getattr(model, "_hyper", {}).get("conf_enc")) else 1
        if share_from is not None:
            first = model.new_engine(batch * self.coalesce, frames, share=share_from)
        else:
            first = model.new_engine(batch * self.coalesce, frames, with_weights=with_weights)
            if after_engine is not None:
                after_engine(first)
        self.engines = [first]
        for _ in range(self.n - 1):
            self.engines.append(model.new_engine(batch * self.coalesce, frames, share=first))
        self._threads = []
        self._inbox = []
        self._busy = threading.Lock()  # one records() call at a time
        self._stage = [{} for _ in range(self.n)]  # per pipeline: merged-batch input buffers at full capacity, by shape

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
        ready = threading.Semaphore(0)

        def loop(k):
            if self._on_gpu:
                torch.cuda.set_device(device)
            st = torch.cuda.Stream() if self._on_gpu else _NoStream()
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
                    finally:
                        job.finished.release()

        self._threads = [threading.Thread(target=loop, args=(k,), daemon=True) for k in range(self.n)]
        for t in self._threads:
            t.start()
        for _ in self._threads:
            ready.acquire()

    def _work(self, k, st, job):
        state, lock, ahead, it = job.state, job.lock, job.ahead, job.it
        on_gpu, device = self._on_gpu, self._device
        while True:
            while not ahead.acquire(timeout=0.05):  # the consumer is behind: wait, but notice a shutdown
                if state["done"] or state["err"] is not None:
                    break
            with lock:
                if state["done"] or state["err"] is not None:
                    break
                items = []
                if state["held"] is not None:
                    items.append(state["held"])
                    state["held"] = None
                else:
                    try:
                        items.append(next(it))
                    except StopIteration:
                        state["done"] = True
                        break
                # further batches of the same shape ride along; another shape waits for the next pass.  A list of known length
                # goes in passes of equal size (see _Job)
                want = self.coalesce
                if job.plan:
                    want = min(want, job.plan.pop(0))
                elif job.left is not None:
                    want = min(want, max(1, -(-job.left // max(1, job.passes_left))))
                    job.passes_left = max(1, job.passes_left - 1)
                while len(items) < want:
                    try:
                        nxt = next(it)
                    except StopIteration:
                        break  # (the next worker to look finds the iterator exhausted)
                    if tuple(nxt[0].shape) == tuple(items[0][0].shape) and ahead.acquire(blocking=False):
                        items.append(nxt)
                    else:
                        state["held"] = nxt
                        break
                if job.left is not None:
                    job.left = max(0, job.left - len(items))
                i = state["next"]
                state["next"] += len(items)
            if len(items) == 1:
                feats, ratio, tag = items[0]
                hyp, hyp_len, score = self.model.decode_device(feats, ratio, job.args, job.sos, engine=self.engines[k])
                recs = [cdist.pack_records(hyp, hyp_len, score)]
            else:
                nb = items[0][0].shape[0]
                dev_ = torch.device("cuda", device) if on_gpu else None
                # the merged input lives in a buffer of the pipeline's full capacity, allocated the first time a shape is merged
                # (any merged warm-up pass, whatever its size, leaves nothing to allocate for the later ones)
                f0, r0 = items[0][0], items[0][1]
                key = (tuple(f0.shape), f0.dtype, tuple(r0.shape), r0.dtype)
                bufs = self._stage[k].get(key)
                if bufs is None:
                    bufs = (torch.empty((self.coalesce * nb,) + tuple(f0.shape[1:]), dtype=f0.dtype, device=dev_),
                            torch.empty((self.coalesce * nb,) + tuple(r0.shape[1:]), dtype=r0.dtype, device=dev_))
                    self._stage[k][key] = bufs
                feats, ratio = bufs[0][: len(items) * nb], bufs[1][: len(items) * nb]
                torch.cat([x[0].to(dev_) if on_gpu else x[0] for x in items], 0, out=feats)  # one launch, nothing allocated
                torch.cat([x[1].to(dev_) if on_gpu else x[1] for x in items], 0, out=ratio)
                hyp, hyp_len, score = self.model.decode_device(feats, ratio, job.args, job.sos, engine=self.engines[k], sub_batch=nb)
                rec = cdist.pack_records(hyp, hyp_len, score)
                recs = [rec[j * nb : (j + 1) * nb] for j in range(len(items))]
            ev = None
            if on_gpu:
                if job.host:  # pinned buffer from torch's caching host allocator, asynchronous copy on this pipeline's stream
                    whole = recs[0] if len(items) == 1 else rec
                    hbuf = torch.empty(whole.shape, dtype=whole.dtype, device="cpu", pin_memory=True)
                    hbuf.copy_(whole, non_blocking=True)
                    nb_ = whole.shape[0] // len(items)
                    recs = [hbuf[j * nb_ : (j + 1) * nb_] for j in range(len(items))]
                ev = torch.cuda.Event()
                ev.record(st)
            for j, item in enumerate(items):
                job.slot(i + j).put((item[2], recs[j], ev))
        st.synchronize()

    def close(self):
        for q in self._inbox:
            q.put(None)
        for t in self._threads:
            t.join()
        self._threads, self._inbox = [], []
        for e in self.engines:
            e.close()
        self.engines = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def records(self, batches, args, sos=1, host=False):
        """``batches``: iterable of ``(feats (B,T,F), size_ratio (B,), tag)`` (host or device tensors).  Yields
        ``(tag, records)`` in the order of the iterable: ``records`` is the device tensor of ``dist.pack_records`` (per
        utterance: length, float64 score, [sos] + tokens), ready for ``dist.all_gather_records`` / ``unpack_records``.  The
        consumer's current stream is made to wait for the producing pipeline's work.  ``host=True``: the records arrive as
        host tensors instead (pinned; sent by the producing pipeline on its own stream, one copy per engine pass, and complete when
        they are yielded) - what a single-GPU consumer wants, which would otherwise pay one blocking copy per batch."""
        self._start()
        job = _Job(iter(batches), args, sos, self.n, total=len(batches) if hasattr(batches, "__len__") else None,
                   coalesce=self.coalesce, host=host)
        state, lock = job.state, job.lock
        with self._busy:
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
                    try:
                        tag, rec, ev = job.slot(i).get(timeout=0.05)
                    except queue.Empty:
                        continue
                    with job.cv:
                        job.slots.pop(i, None)
                    if ev is not None:
                        if job.host:
                            ev.synchronize()
                        else:
                            ev.wait(torch.cuda.current_stream())
                    job.ahead.release()
                    yield tag, rec
                    i += 1
            finally:
                with lock:
                    state["done"] = True
                for _ in range(self.n):  # every worker has left the job (its stream is drained) before the next one starts
                    job.finished.acquire()

    def decode(self, batches, args, sos=1, gather=False, as_lists=True, gather_every=None):
        """Hypotheses on the host, in order: yields ``(tag, hyps, scores)`` with ``hyps`` a list of token lists starting
        with ``sos`` (what ``beam_decode`` returns as ``['hyp']``), or with ``as_lists=False`` the arrays ``(tokens (N, S),
        lengths (N,))`` of ``dist.unpack_records``.  ``gather=True``: the multi-GPU path - every step's records of all ranks,
        rank-major.  The all-gather (and the copy home) is issued once per ``gather_every`` consecutive steps of equal shape
        (default: the pipelines' batches per pass; SURVEY 8e: "once per batch (or once per N batches)"), by step index - the
        same sequence of collectives on every rank whatever the pipelines' timing; every rank must decode the same number of
        steps."""
        if not gather:
            for tag, rec in self.records(batches, args, sos, host=True):
                hyps, scores = cdist.unpack_records(rec, as_lists=as_lists)
                yield tag, hyps, scores
            return
        group = max(1, int(gather_every or self.coalesce))
        pending = []

        def flush():
            recs = [r for _, r in pending]
            nb, width = recs[0].shape
            everyone = cdist.all_gather_records(torch.cat(recs, 0) if len(recs) > 1 else recs[0]).cpu()  # one collective, one copy
            world = everyone.shape[0] // (len(recs) * nb)
            steps = everyone.view(world, len(recs), nb, width)
            out = [(tag, cdist.unpack_records(steps[:, j].reshape(world * nb, width), as_lists=as_lists)) for j, (tag, _) in enumerate(pending)]
            pending.clear()
            return out

        for tag, rec in self.records(batches, args, sos):
            if pending and tuple(rec.shape) != tuple(pending[0][1].shape):
                for t_, (h_, s_) in flush():
                    yield t_, h_, s_
            pending.append((tag, rec))
            if len(pending) == group:
                for t_, (h_, s_) in flush():
                    yield t_, h_, s_
        if pending:
            for t_, (h_, s_) in flush():
                yield t_, h_, s_
