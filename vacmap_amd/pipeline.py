"""The product's batch schedule: length-binned batches inside a bounded window, several batches in flight per GPU.

Counterpart of the reference's worker loop (`get_list_of_readmap_stdout`, mammap_clrnano.py:24086-24154: `t` processes each pull
reads from a queue and push 2 MB of SAM text at a time; input order is not preserved across workers, :24147-24150). Here one process
drives one GPU:

  * reads are taken in arrival order in WINDOWS of `window_batches * batch_reads` reads; inside a window they are sorted by length
    (stable) and cut into batches, so that the one-wavefront-per-read kernels of a batch finish together (`plan_batches`);
  * `inflight` contexts (vm_ctx = HIP stream set + work pools), one host thread each, pull batches from the window, longest reads
    first; their kernels overlap on the device — the latency-bound chain / re-seeding kernels of one batch run under the VALU-bound
    gap fill of another — and every context is told how many share the GPU (`vm_ctx_set_inflight`);
  * results are handed to the caller per batch as they complete (`on_result`), so SAM emission overlaps the next batches.

`bench.py` times `Pipeline.run_resident` (reads already in HBM); `vacmap_amd/driver.py` feeds `Pipeline.run_host` from its FASTX /
BAM reader. The HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); a batch drives one main
and four side streams, so the package asks for 16 before the runtime starts (vacmap_amd/__init__.py).
"""
import os
import threading

import numpy as np

from .lib import Context, ResidentReads, align_batch

class _Roctx:
    """optional roctx ranges around every batch (VMX_ROCTX=1): `rocprofv3 --marker-trace` then shows which batch a kernel belongs to"""

    def __init__(self):
        import ctypes, os
        self.lib = None
        if os.environ.get('VMX_ROCTX') == '1':
            for name in ('libroctx64.so', 'librocprofiler-sdk-roctx.so'):
                try:
                    self.lib = ctypes.CDLL(name); break
                except OSError:
                    continue
            if self.lib is not None:
                self.lib.roctxRangePushA.argtypes = [ctypes.c_char_p]

    def push(self, text):
        if self.lib is not None:
            self.lib.roctxRangePushA(text.encode())

    def pop(self):
        if self.lib is not None:
            self.lib.roctxRangePop()


_roctx = _Roctx()

DEFAULT_INFLIGHT = 5
DEFAULT_BATCH_READS = 4096
DEFAULT_WINDOW_BATCHES = 16
DEFAULT_BATCH_MAX_BASES = 0          # plan_batches: bases a job holds at most (0: no limit)


def _max_bases_default():
    v = os.environ.get('VMX_BATCH_MAX_BASES', '')
    try:
        return int(float(v)) if v else DEFAULT_BATCH_MAX_BASES
    except ValueError:
        return DEFAULT_BATCH_MAX_BASES


def plan_batches(lengths, batch_reads=DEFAULT_BATCH_READS, window_batches=DEFAULT_WINDOW_BATCHES, sort=True, max_bases=None):
    """read lengths in arrival order -> list of int64 index arrays (one per batch), window by window; inside a window the batches
    hold reads of ascending length and are listed LONGEST FIRST (the streams finish on the short ones). sort=False keeps arrival
    order (the unsorted schedule, measured once for comparison).
    max_bases (default: VMX_BATCH_MAX_BASES, else DEFAULT_BATCH_MAX_BASES; 0 = no limit): a batch of `batch_reads` reads that holds more bases than this is
    cut into parts of equal bases, each a job of its own. A context's grow-only pools are sized by the LARGEST job it ever runs; with reads sorted by
    length the batch of a window's longest reads holds three times the bases of the median one (ONT shape: 175 against 61 Mbases) and sized every context
    for itself — 50 GB, five contexts in the HBM next to an hg38-size index. Reads are independent: the records do not change."""
    lengths = np.asarray(lengths, dtype=np.int64)
    n = len(lengths)
    win = max(1, int(batch_reads) * max(1, int(window_batches)))
    if max_bases is None:
        max_bases = _max_bases_default()
    out = []
    for w0 in range(0, n, win):
        idx = np.arange(w0, min(n, w0 + win), dtype=np.int64)
        if sort:
            idx = idx[np.argsort(lengths[idx], kind='stable')]
        if sort:            # cut from the long end: the longest reads fill whole batches, a short remainder holds the shortest
            batches = [idx[max(0, e - batch_reads):e] for e in range(len(idx), 0, -batch_reads)]
        else:
            batches = [idx[i:i + batch_reads] for i in range(0, len(idx), batch_reads)]
        for b in batches:
            out.extend(split_by_bases(b, lengths, max_bases))
    return out


def split_by_bases(batch, lengths, max_bases):
    """one batch (index array) -> its parts of at most ~max_bases bases each (equal shares of the batch's bases, consecutive reads, the part listed
    first holds the batch's last reads — the longest ones of a sorted batch); [batch] when it fits or the limit is off"""
    if not max_bases or max_bases <= 0 or len(batch) <= 1:
        return [batch]
    ln = lengths[batch]
    total = int(ln.sum())
    if total <= max_bases:
        return [batch]
    parts = min(len(batch), -(-total // int(max_bases)))
    cum = np.cumsum(ln)
    cuts = [0] + [int(np.searchsorted(cum, total * j / parts, side='left')) + 1 for j in range(1, parts)] + [len(batch)]
    cuts = sorted(set(min(max(c, 0), len(batch)) for c in cuts))
    return [batch[a:b] for a, b in zip(cuts[:-1], cuts[1:]) if b > a][::-1]


class Pipeline:
    def __init__(self, index, prm, device=0, inflight=DEFAULT_INFLIGHT, first_ctx=None):
        self.index, self.prm = index, prm
        self.inflight = max(1, int(inflight))
        first = first_ctx or index.ctx
        self.device = device
        self.ctxs = [first] + [Context(device, lib=first.lib) for _ in range(self.inflight - 1)]
        for cx in self.ctxs:
            cx.set_inflight(self.inflight)

    def close(self):
        if getattr(self, '_up', None) is not None:
            for sl in self._up[1]:
                sl.close()
            self._up[0].close(); self._up = None
        for cx in self.ctxs[1:]:
            cx.close()
        self.ctxs = self.ctxs[:1]

    def trim_to_memory(self, min_free_gb=10.0, keep=2):
        """after the sizing run of every context (warm): the grow-only pools scale with the longest batch, and four contexts' pools of a long-read
        batch may leave the device without head-room; contexts are given up (their pools freed) until `min_free_gb` is free or `keep` remain.
        Returns the number of contexts dropped."""
        dropped = 0
        while len(self.ctxs) > max(1, keep):
            free, _ = self.ctxs[0].mem_info()
            if free >= min_free_gb * 1e9:
                break
            self.ctxs.pop().close(); dropped += 1
        if dropped:
            self.inflight = len(self.ctxs)
            for cx in self.ctxs:
                cx.set_inflight(self.inflight)
        return dropped

    def grow_to_memory(self, resident=None, run=None, max_inflight=8, min_free_gb=14.0):
        """after warm(): more batches in flight while the HBM has room for another context's pools. What a context's pools take is measured on the first
        one added (free memory before and after its sizing run on the same batch the others were sized with); a context is added while that much plus
        `min_free_gb` is free. Throughput is batches in flight / the latency of a batch under load wherever the step is not bound by the VALUs (HiFi:
        the step is 2.4x its VALU floor; uniform 18 kb reads leave room for seven contexts where the ONT workload's long-read batches fill the HBM with
        five). Returns the number of contexts added."""
        from .lib import VmxError
        added, per_ctx = 0, None
        while len(self.ctxs) < max_inflight:
            free, _ = self.ctxs[0].mem_info()
            need = (per_ctx if per_ctx is not None else 0.0) + min_free_gb * 1e9
            if per_ctx is None and free < 60e9:                   # nothing is known yet: only try with plenty of room
                break
            if per_ctx is not None and free < need:
                break
            cx = Context(self.device, lib=self.ctxs[0].lib)
            try:
                if run is not None:
                    run(cx)
                else:
                    resident.align(self.index, self.prm, want_records=False, ctx=cx)
            except VmxError as e:
                cx.close()
                if e.code != -4:
                    raise
                break
            free2, _ = self.ctxs[0].mem_info()
            per_ctx = max(float(free - free2), 1e9)
            if free2 < min_free_gb * 1e9:                          # it fitted, but left too little: give it back
                cx.close()
                break
            self.ctxs.append(cx); added += 1
        if added:
            self.inflight = len(self.ctxs)
            for cx in self.ctxs:
                cx.set_inflight(self.inflight)
        return added

    def retire_if_low(self, cx, low_water_gb=4.0, keep=2):
        """called by a worker thread between two batches (under the scheduler's lock): the pools are grow-only, and a later window with longer
        reads regrows every context's pools after trim_to_memory has run. When less than `low_water_gb` of HBM is free and more than
        `keep` contexts are at work, this one gives up: its pools are freed and its thread ends — the run goes on with one batch fewer in flight
        instead of failing a hipMalloc (vm_align_batch itself degrades to sub-batches when memory runs out, which is much slower). Returns True
        when the context was given up. The first context (the index's) never retires."""
        if cx is self.ctxs[0] or len(self.ctxs) <= max(1, keep):
            return False
        free, _ = cx.mem_info()
        if free >= low_water_gb * 1e9:
            return False
        self.ctxs.remove(cx); cx.close()
        self.inflight = len(self.ctxs)
        for c in self.ctxs:
            c.set_inflight(self.inflight)
        return True

    def full_ctxs(self):
        """the contexts that may be handed any batch (the small ones, add_small_contexts, only take part in the size-aware schedule of _run)"""
        small = set(id(c) for c in getattr(self, 'small_ctxs', []))
        return [c for c in self.ctxs if id(c) not in small]

    def add_small_contexts(self, resident_small, limit_bases, max_inflight=9, min_free_gb=14.0):
        """contexts for the SHORTER batches only. The work pools are sized by the largest batch a context has seen, and a length-binned window holds a few
        batches of long reads among many of short ones (Gamma lengths: 160 Mbases in the longest batch of a window of 16, ~50 in the median one): contexts that
        will never be handed more than `limit_bases` bases are sized on such a batch (`resident_small`) and take a third of a full context's memory, so that more
        batches are in flight inside the same HBM. The scheduler (_run with job_bases) gives them the jobs at or below the limit. Returns the number added."""
        from .lib import VmxError
        added, per_ctx = 0, None
        self.small_limit = int(limit_bases)
        if not hasattr(self, 'small_ctxs'):
            self.small_ctxs = []
        while len(self.ctxs) < max_inflight:
            free, _ = self.ctxs[0].mem_info()
            if per_ctx is not None and free < per_ctx + min_free_gb * 1e9:
                break
            if per_ctx is None and free < (min_free_gb + 8.0) * 1e9:
                break
            cx = Context(self.device, lib=self.ctxs[0].lib)
            try:
                resident_small.align(self.index, self.prm, want_records=False, ctx=cx)
            except VmxError as e:
                cx.close()
                if e.code != -4:
                    raise
                break
            free2, _ = self.ctxs[0].mem_info()
            per_ctx = max(float(free - free2), 1e9)
            if free2 < min_free_gb * 1e9:
                cx.close()
                break
            self.ctxs.append(cx); self.small_ctxs.append(cx); added += 1
        if added:
            self.inflight = len(self.ctxs)
            for cx in self.ctxs:
                cx.set_inflight(self.inflight)
        return added

    def _run(self, n_jobs, do_job, on_result, job_bases=None, horizon=DEFAULT_WINDOW_BATCHES):
        """`inflight` threads pull job indices in order; do_job(i, ctx) -> result; on_result(i, result) is called under a lock.
        job_bases (with small contexts, add_small_contexts): a full context takes the first job not taken yet; a small one the first job not taken yet that
        holds at most `small_limit` bases, looking no further than `horizon` jobs ahead of the first untaken one (a stream does not hold more than that)"""
        lock = threading.Lock()
        nxt = [0]
        errs = []
        small = set(id(c) for c in getattr(self, 'small_ctxs', [])) if job_bases is not None else set()
        taken = [False] * n_jobs
        lpt = os.environ.get('VMX_SCHED_LPT', '1') != '0'
        import time as _time
        tl = [] if os.environ.get('VMX_DBG_TIMELINE') else None      # tuning aid: (job, context, small?, start s, end s, bases) of every job of this run
        self.timeline = tl; t_run0 = _time.time()

        def take(cx):
            """index of the job this context runs next, -1: none left, -2: none eligible right now"""
            while nxt[0] < n_jobs and taken[nxt[0]]:
                nxt[0] += 1
            if nxt[0] >= n_jobs:
                return -1
            if id(cx) not in small:
                i = nxt[0]
                if lpt and job_bases is not None:
                    # round 6: the LARGEST job among those a stream holds (the first untaken one and `horizon` - 1 behind it), not simply the next one: when a window runs
                    # out of large batches the contexts that free up start on the next window's long-read batches at once instead of after its own small ones — the
                    # drain of a run is tighter (20 batches: 16.0 -> 15.x ms per batch; long runs unchanged)
                    for j in range(nxt[0] + 1, min(n_jobs, nxt[0] + horizon)):
                        if not taken[j] and job_bases[j] > job_bases[i]:
                            i = j
                taken[i] = True
                return i
            for i in range(nxt[0], min(n_jobs, nxt[0] + horizon)):
                if not taken[i] and job_bases[i] <= self.small_limit:
                    taken[i] = True
                    return i
            return -2

        def worker(cx):
            try:
                while not errs:
                    with lock:
                        i = take(cx)
                    if i == -2:
                        _time.sleep(0.0005)
                        continue
                    if i < 0:
                        return
                    _roctx.push('vacmapx batch %d' % i)
                    t_a = _time.time()
                    try:
                        res = do_job(i, cx)               # ctypes releases the GIL for the library call
                    finally:
                        _roctx.pop()
                    if tl is not None:
                        tl.append((i, self.ctxs.index(cx), id(cx) in small, t_a - t_run0, _time.time() - t_run0, int(job_bases[i]) if job_bases is not None else 0))
                    if on_result is not None:
                        with lock:
                            on_result(i, res)
                    with lock:
                        if self.retire_if_low(cx):
                            return
            except BaseException as e:                    # a failed batch must fail the run, not hang it
                errs.append(e)

        if self.inflight == 1 or n_jobs <= 1:
            worker(self.ctxs[0])
        else:
            pool = list(self.ctxs) if small else self.full_ctxs()
            th = [threading.Thread(target=worker, args=(cx,)) for cx in pool[:min(len(pool), n_jobs)]]
            for t in th:
                t.start()
            for t in th:
                t.join()
        if errs:
            raise errs[0]

    def run_stream(self, jobs, do_job, errs=None, ramp=None):
        """`inflight` threads pull jobs from the iterator `jobs` (one at a time, under a lock) and run do_job(job, ctx) until it is
        exhausted: no barrier between groups of jobs, the contexts stay busy across the caller's windows. An exception in any thread is
        appended to `errs` (shared with the caller's other threads), stops the others and is re-raised.
        ramp = dict(run=callable(ctx), target=N or 0, max_inflight=8, min_free_gb=14.0, on_ctx=callable(ctx) or None): the stream STARTS on the contexts
        that exist (the caller has sized the first one) while a background thread creates the others one at a time, sizes each one's work pools
        (run(ctx): the caller's sizing batch) and starts a worker on it — up to `target` contexts, or with target 0 by grow_to_memory's rule (another
        context while what one took, measured on the first one added, plus `min_free_gb` is free; at most `max_inflight`). The sizing runs of five to
        eight contexts take 1-5 s of hipMalloc (~50 GB each) during which the stream used to wait; now the first batches run under them."""
        from .lib import VmxError
        lock = threading.Lock()
        errs = [] if errs is None else errs
        it = iter(jobs)
        done = threading.Event()                       # the job iterator is exhausted: no more contexts are worth sizing
        late = []

        def worker(cx):
            try:
                while not errs:
                    with lock:
                        job = next(it, None)
                    if job is None:
                        done.set()
                        return
                    _roctx.push('vacmapx batch')
                    try:
                        do_job(job, cx)
                    finally:
                        _roctx.pop()
                    with lock:
                        if self.retire_if_low(cx):
                            return
            except BaseException as e:
                errs.append(e)

        def grower():
            try:
                per_ctx = None
                target = int(ramp.get('target') or 0); cap = int(ramp.get('max_inflight', 8)); min_free = float(ramp.get('min_free_gb', 14.0)) * 1e9
                while not errs and not done.is_set() and len(self.ctxs) < (target or cap):
                    free, _ = self.ctxs[0].mem_info()
                    if not target:
                        if per_ctx is None and free < 60e9:
                            break
                        if per_ctx is not None and free < per_ctx + min_free:
                            break
                    cx = Context(self.device, lib=self.ctxs[0].lib)
                    if ramp.get('on_ctx') is not None:
                        ramp['on_ctx'](cx)
                    cx.set_inflight(len(self.ctxs) + 1)
                    try:
                        ramp['run'](cx)
                    except VmxError as e:
                        cx.close()
                        if e.code != -4:                          # VM_ERR_OOM: no room for another context's pools — fewer batches in flight, not a failed run
                            raise
                        self.ramp_oom = getattr(self, 'ramp_oom', 0) + 1
                        break
                    free2, _ = self.ctxs[0].mem_info()
                    per_ctx = max(float(free - free2), 1e9) if per_ctx is None else max(per_ctx, float(free - free2))
                    if free2 < min_free and len(self.ctxs) >= 2:   # it fitted, but left too little: give it back
                        cx.close()
                        break
                    with lock:
                        self.ctxs.append(cx); self.inflight = len(self.ctxs)
                        for c in self.ctxs:
                            c.set_inflight(self.inflight)
                    t = threading.Thread(target=worker, args=(cx,)); late.append(t); t.start()
            except BaseException as e:
                errs.append(e)

        th = [threading.Thread(target=worker, args=(cx,)) for cx in self.full_ctxs()]
        for t in th:
            t.start()
        gt = None
        if ramp is not None:
            gt = threading.Thread(target=grower); gt.start()
        for t in th:
            t.join()
        if gt is not None:
            gt.join()
        for t in late:
            t.join()
        if errs:
            raise errs[0]

    def run_resident(self, resident, want_records=False, on_result=None):
        """resident: list of ResidentReads in schedule order (plan_batches). on_result(i, (status, records or None, stats))"""
        self._run(len(resident), lambda i, cx: resident[i].align(self.index, self.prm, want_records=want_records, ctx=cx), on_result,
                  job_bases=[r.bases for r in resident])

    def run_host(self, batches, on_result=None):
        """batches: list of lists of read sequences (host memory; uploaded by vm_align_batch). on_result(i, (status, records, stats))"""
        self._run(len(batches), lambda i, cx: align_batch(cx, self.index, self.prm, batches[i]), on_result)

    def upload_slots(self, largest_blob, n_slots=None):
        """the uploader's side of run_host_blobs(prefetch=True): a context of its own (its stream carries the copies and the base encoding) and a few
        reusable ResidentReads sized by the largest batch, made before anything is timed (like the contexts' work pools)"""
        if getattr(self, '_up', None) is None:
            first = self.ctxs[0]
            up_ctx = Context(self.device, lib=first.lib)
            n_small = len(getattr(self, 'small_ctxs', []))
            self._up = (up_ctx, [ResidentReads(up_ctx, concat=largest_blob[0], offsets=largest_blob[1]) for _ in range(n_slots or self.inflight + 2 + 2 * n_small)])
        return self._up

    def run_host_blobs(self, blobs, on_result=None, prefetch=False):
        """blobs: list of (uint8 array of the batch's reads back to back, int64 offsets[n + 1]) in HOST memory (the PCIe-inclusive path).
        prefetch=False: vm_align_batch uploads a batch inside the call, in front of its own kernels. prefetch=True: an uploader thread with a context of
        its own streams the batches into HBM ahead of the aligning contexts (vm_reads_reupload into a few reusable slots, at most inflight + 2 batches
        ahead), so the copy of batch i + 1 runs under the kernels of batch i; the aligners then run vm_align_resident. on_result(i, stats dict)"""
        from .lib import align_batch_raw
        if prefetch and blobs:
            import queue
            big = max(blobs, key=lambda b: int(b[1][-1]))
            up_ctx, slots = self.upload_slots(big)
            free_q = queue.Queue()
            for sl in slots:
                free_q.put(sl)
            errs = []
            ready, cv, state = [], threading.Condition(), {'done': False}     # uploaded batches waiting for a context, in upload order
            small = set(id(c) for c in getattr(self, 'small_ctxs', []))
            # (with small contexts the uploader runs further ahead — upload_slots makes two more slots per small context: a small context can only take a batch
            # at or below its limit, and the batches of a window come longest first)

            def producer():
                try:
                    for i, (cat, off) in enumerate(blobs):
                        sl = free_q.get()
                        if sl is None:
                            return
                        sl.reupload(cat, off, ctx=up_ctx)
                        with cv:
                            ready.append((i, sl)); cv.notify_all()
                except BaseException as e:
                    errs.append(e)
                finally:
                    with cv:
                        state['done'] = True; cv.notify_all()
            lock = threading.Lock()

            def consumer(cx):
                is_small = id(cx) in small
                try:
                    while not errs:
                        with cv:
                            item = None
                            while item is None and not errs:
                                for x, (i, sl) in enumerate(ready):
                                    if not is_small or sl.bases <= self.small_limit:
                                        item = ready.pop(x); break
                                if item is None:
                                    if state['done'] and (not ready or is_small):
                                        return
                                    cv.wait(0.05)
                        if item is None:
                            return
                        i, sl = item
                        _st, _r, sd = sl.align(self.index, self.prm, want_records=False, ctx=cx)
                        free_q.put(sl)
                        if on_result is not None:
                            with lock:
                                on_result(i, sd)
                except BaseException as e:
                    errs.append(e); free_q.put(None)
                    with cv:
                        cv.notify_all()
            th = [threading.Thread(target=producer)] + [threading.Thread(target=consumer, args=(cx,)) for cx in list(self.ctxs)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            if errs:
                raise errs[0]
            return

        def job(i, cx):
            raw = align_batch_raw(cx, self.index, self.prm, blobs[i][0], blobs[i][1])
            st = raw.stats
            raw.close()
            return st
        self._run(len(blobs), job, on_result)

    def warm(self, resident=None, run=None, keep=1):
        """run every context once on `resident` (the largest batch), or call run(ctx): sizes the grow-only work pools so that no hipMalloc happens
        later. A context whose sizing run finds no memory left (VM_ERR_OOM: the pools of a batch of very long reads times the contexts before it) is
        given up together with the ones after it, as long as `keep` remain — fewer batches in flight, not a failed run. Returns the contexts dropped."""
        from .lib import VmxError
        dropped = 0
        for i, cx in enumerate(list(self.ctxs)):
            try:
                if run is not None:
                    run(cx)
                else:
                    resident.align(self.index, self.prm, want_records=False, ctx=cx)
            except VmxError as e:
                if e.code != -4 or i < max(1, keep):              # VM_ERR_OOM
                    raise
                for c2 in self.ctxs[i:]:
                    c2.close(); dropped += 1
                self.ctxs = self.ctxs[:i]
                break
        if dropped:
            self.inflight = len(self.ctxs)
            for cx in self.ctxs:
                cx.set_inflight(self.inflight)
        return dropped


def upload_batches(ctx, concat, offsets, plan):
    """gather each planned batch from the concatenated pool (uint8, int64 offsets) and upload it: list of ResidentReads"""
    concat = np.asarray(concat, dtype=np.uint8); offsets = np.asarray(offsets, dtype=np.int64)
    lens = np.diff(offsets)
    out = []
    for idx in plan:
        ln = lens[idx]
        off = np.concatenate([[0], np.cumsum(ln)]).astype(np.int64)
        cat = np.empty(int(off[-1]), np.uint8)
        for j, i in enumerate(idx):
            cat[off[j]:off[j + 1]] = concat[offsets[i]:offsets[i + 1]]
        out.append(ResidentReads(ctx, concat=cat, offsets=off))
    return out
