"""CodecPipeline -- the encoder || decoder schedule as a library object.

The reference's harness codes one item after the other on one model object: compress, then decompress, then the next item
(/root/reference/src/compress/training/step.py:297-340).  A decoder spends part of every slice step waiting for the host entropy
coder, an encoder ends on a hand-over of its last strings: run strictly in turn on one object those gaps are idle GPU time (DESIGN.md
section 6: 41.7 against 48 MP/s on Config 2).  A ``CodecPipeline`` owns what fills them:

* two ``ChannelProgresssiveWACNN`` objects with the same weights -- an encoder and a decoder, each with its own weights copy in HBM,
  workspaces and streams inside libpcodec.so (a codec object serves one call at a time: include/pcodec.h, PC_ERR_STATE);
* two HIP streams (encode, decode) and a decoder host thread, so that the decode of item i runs beside the encode of item i+1;
* the check that the HIP runtime was given enough hardware queues (``GPU_MAX_HW_QUEUES``: the two objects keep ~20 streams busy; with
  the default 4 the overlapped run is bimodal -- profiles/r02_n_hw_queues.log).

Every item is still one compress() and one decompress() of the item (or compress_levels / decompress_levels for a list of levels); the
strings and reconstructions are bit-identical to the sequential calls on one object (tests/test_gpu_codec.py::test_codec_pipeline_*).
Nothing here computes: all arithmetic is in the native library.
"""
import collections
import os
import queue
import threading
import warnings

#: hardware queues the overlapped schedule wants (stable from 12 up; profiles/r02_n_hw_queues.log)
HW_QUEUES = 16


def request_hw_queues(n=HW_QUEUES):
    """Ask the HIP runtime for `n` hardware queues.  The runtime reads GPU_MAX_HW_QUEUES once, when it initialises: call this (or import
    progressivecodec_amd, which does) BEFORE the first GPU call of the process.  Returns True when the setting is (already) in place
    for a runtime that has not started yet, False when HIP was initialised before with fewer queues."""
    cur = os.environ.get("GPU_MAX_HW_QUEUES")
    started = False
    try:
        import sys
        torch = sys.modules.get("torch")
        started = bool(torch is not None and torch.cuda.is_initialized())
    except Exception:                                    # a torch without cuda support: nothing to initialise
        started = False
    if cur is not None:
        try:
            return int(cur) >= 12 or not started
        except ValueError:
            return False
    if started:
        return False
    os.environ["GPU_MAX_HW_QUEUES"] = str(n)
    return True


class PipelineError(RuntimeError):
    pass


class CodecPipeline:
    """``CodecPipeline(state_dict, device="cuda:0")`` builds the encoder and the decoder object from one state dict (the reference's
    1019-key layout) and runs ``update()`` on both; ``CodecPipeline.from_model(net)`` takes an existing, loaded model as the encoder and
    builds the decoder from its state dict.  Extra keyword arguments go to the ``ChannelProgresssiveWACNN`` constructor.

    ``code(jobs)`` is the schedule: a generator over ``jobs`` (dictionaries: ``x`` [B,3,H,W] with H, W multiples of 64, ``quality`` or
    ``qualities`` (a list: the shared-base multi-level path), optional ``mask_pol``) yielding ``(job, enc, dec)`` in job order --
    ``enc`` / ``dec`` are what ``compress`` / ``decompress`` (or ``compress_levels`` / ``decompress_levels``) return.  The encoder runs at
    most ``queue_depth`` items ahead of the decoder.  Results are safe to use on the caller's current stream when they are yielded.

    ``n_pairs`` > 1 (round 4): that many encoder / decoder PAIRS, each with its own objects, streams, encoder thread and decoder thread;
    jobs are handed to whichever pair is free and the results come back in job order.  One pair keeps about 2.4 convolution kernels in
    flight, which leaves the MFMA pipes idle ~30 % of the time; a second pair fills part of that: 47.2 -> 49.7
    MP/s on Config 2, 50.5 with three (tools/multi_pipeline_probe.py, profiles/r04_m_*).  Every pair is another two weights copies
    (1.2 GB) plus its workspaces (3 GB at Config 2, 48 GB at Config 4's shard): the default stays 1; bench.py uses 2.
    """

    def __init__(self, state_dict=None, device="cuda:0", queue_depth=2, _encoder=None, n_pairs=1, **model_kwargs):
        import torch
        from .model import ChannelProgresssiveWACNN
        self.hw_queues_ok = request_hw_queues()
        if not self.hw_queues_ok:
            warnings.warn("CodecPipeline: the HIP runtime was initialised with GPU_MAX_HW_QUEUES=%s (< 12): the encoder / decoder overlap "
                          "shares hardware queues and its rate is bimodal; set GPU_MAX_HW_QUEUES=16 (or import progressivecodec_amd) before the "
                          "first GPU call" % os.environ.get("GPU_MAX_HW_QUEUES", "default (4)"), RuntimeWarning, stacklevel=2)
        if _encoder is not None:
            self.enc = _encoder
            state_dict = _encoder.state_dict()
            device = str(_encoder.device)
            model_kwargs = dict(N=_encoder.cfg.N, M=_encoder.cfg.M, division_dimension=_encoder.cfg.division_dimension, dim_chunk=_encoder.cfg.dim_chunk,
                                multiple_decoder=_encoder.cfg.multiple_decoder, multiple_encoder=_encoder.cfg.multiple_encoder,
                                multiple_hyperprior=_encoder.cfg.multiple_hyperprior, mask_policy=_encoder.mask_policy,
                                joiner_policy=_encoder.cfg.joiner_policy, support_progressive_slices=_encoder.cfg.support_progressive_slices,
                                delta_encode=_encoder.cfg.delta_encode)
        else:
            if state_dict is None:
                raise ValueError("CodecPipeline needs a state dict (or use CodecPipeline.from_model)")
            self.enc = ChannelProgresssiveWACNN(device=device, **model_kwargs)
            self.enc.load_state_dict(state_dict)
            self.enc.update()
        def another():
            net = ChannelProgresssiveWACNN(device=device, **model_kwargs)
            net.load_state_dict(state_dict)           # carries the encoder's CDF tables and scale table when it has them
            net.update()
            return net
        self.dec = another()
        self.device = self.enc.device
        self.queue_depth = max(1, int(queue_depth))
        self.s_enc = torch.cuda.Stream(self.device)
        self.s_dec = torch.cuda.Stream(self.device)
        self.n_pairs = max(1, int(n_pairs))
        #: (encoder, decoder, encode stream, decode stream) per pair; pair 0 is self.enc / self.dec
        self.pairs = [(self.enc, self.dec, self.s_enc, self.s_dec)]
        for _ in range(self.n_pairs - 1):
            self.pairs.append((another(), another(), torch.cuda.Stream(self.device), torch.cuda.Stream(self.device)))
        self._busy = threading.Lock()

    @property
    def objects(self):
        """every codec object of the pipeline (encoders and decoders of all pairs)"""
        return [o for p in self.pairs for o in p[:2]]

    @classmethod
    def from_model(cls, model, queue_depth=2, n_pairs=1):
        """`model`: a loaded and updated ChannelProgresssiveWACNN -- it becomes the encoder; the decoder is a second object with the same
        state dict (another 608 MB of HBM)."""
        if model._gc is None or model._eb is None:
            raise ValueError("Uninitialized CDFs. Run update() first")
        return cls(_encoder=model, queue_depth=queue_depth, n_pairs=n_pairs)

    # ------------------------------------------------------------------ one item on one object
    @staticmethod
    def _encode(net, job):
        mp = job.get("mask_pol", "point-based-std")
        if "qualities" in job:
            return net.compress_levels(job["x"], job["qualities"], mask_pol=mp)
        return net.compress(job["x"], job["quality"], mp, job.get("cust_map"))

    @staticmethod
    def _decode(net, job, enc):
        mp = job.get("mask_pol", "point-based-std")
        if "qualities" in job:
            return net.decompress_levels([d["strings"] for d in enc], enc[0]["shape"], job["qualities"], mask_pol=mp)
        return net.decompress(enc["strings"], enc["shape"], job["quality"], mp, job.get("cust_map"))

    def code_sequential(self, jobs):
        """The same jobs strictly one after the other on the encoder object alone (the reference's loop): the A/B partner of code()."""
        for job in jobs:
            enc = self._encode(self.enc, job)
            yield job, enc, self._decode(self.enc, job, enc)

    # ------------------------------------------------------------------ the overlapped schedule
    def code(self, jobs, on_encoded=None):
        """Generator: (job, enc, dec) per job, in order; the decode of job i runs beside the encode of job i+1.  `on_encoded(job, enc)`
        (optional) is called on the caller's thread right after a job's compress returned, before its decode is queued."""
        import torch
        if self.n_pairs > 1:
            yield from self._code_pairs(jobs, on_encoded)
            return
        if not self._busy.acquire(blocking=False):
            raise PipelineError("this CodecPipeline is already running a code() loop (one schedule per pipeline object)")
        q_in = queue.Queue(maxsize=self.queue_depth)
        done = collections.deque()
        cv = threading.Condition()
        state = {"err": None}
        dev, s_dec, dec_net = self.device, self.s_dec, self.dec

        def decoder():
            try:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(s_dec):
                    while True:
                        item = q_in.get()
                        if item is None:
                            return
                        job, enc = item
                        e0 = None
                        if job.get("time_decode"):               # GPU-stream time of this item's decode (CodecPipeline.decode_ms)
                            e0 = torch.cuda.Event(enable_timing=True)
                            e0.record(s_dec)
                        dec = self._decode(dec_net, job, enc)
                        ev = torch.cuda.Event(enable_timing=e0 is not None)
                        ev.record(s_dec)                         # x_hat is still in flight on s_dec: the consumer's stream waits for this
                        if e0 is not None:
                            job["_dec_events"] = (e0, ev)
                        with cv:
                            done.append((job, enc, dec, ev))
                            cv.notify_all()
            except BaseException as e:                           # surfaced on the caller's thread
                with cv:
                    state["err"] = e
                    cv.notify_all()
                while q_in.get() is not None:                    # keep draining so that the producer can never block on a dead consumer
                    pass

        th = threading.Thread(target=decoder, name="pcodec-decoder", daemon=True)   # daemon: a failure on either side never hangs the exit
        th.start()
        n_in = n_out = 0
        caller = torch.cuda.current_stream(dev)

        def ready():
            out = []
            with cv:
                while done:
                    out.append(done.popleft())
            return out

        def hand(item):
            # the caller's stream waits for the decode; the tensors were allocated on the pipeline's streams, so the caching allocator is
            # told that the caller's stream uses them too (it must not hand their memory to the next item while the caller still reads)
            job, enc, dec, ev = item
            cur = torch.cuda.current_stream(dev)
            cur.wait_event(ev)
            for d in (dec if isinstance(dec, (list, tuple)) else [dec]):
                d["x_hat"].record_stream(cur)
            for e in (enc if isinstance(enc, (list, tuple)) else [enc]):
                for m in e.get("masks", []):
                    m.record_stream(cur)
            return job, enc, dec
        try:
            for job in jobs:
                if state["err"] is not None:
                    break
                self.s_enc.wait_stream(caller)                   # the job's input was produced on the caller's stream
                with torch.cuda.stream(self.s_enc):
                    enc = self._encode(self.enc, job)
                if on_encoded is not None:
                    on_encoded(job, enc)
                q_in.put((job, enc))                             # blocks while the decoder is queue_depth items behind
                n_in += 1
                for item in ready():
                    n_out += 1
                    yield hand(item)
            q_in.put(None)
            while n_out < n_in and state["err"] is None:
                with cv:
                    while not done and state["err"] is None:
                        if not cv.wait(timeout=600):
                            raise PipelineError("decoder thread made no progress for 600 s")
                for item in ready():
                    n_out += 1
                    yield hand(item)
            if state["err"] is not None:
                raise state["err"]
        finally:
            try:
                q_in.put_nowait(None)                            # (a second sentinel is harmless; after an exception it is the first)
            except queue.Full:
                # the caller abandoned the generator with the queue full: make room -- the decoder drops what it has not started
                try:
                    while True:
                        q_in.get_nowait()
                except queue.Empty:
                    pass
                q_in.put(None)
            th.join(timeout=600)
            caller.wait_stream(self.s_enc)
            caller.wait_stream(self.s_dec)
            self._busy.release()
            if th.is_alive():
                raise PipelineError("decoder thread did not finish")

    def _code_pairs(self, jobs, on_encoded=None):
        """code() with n_pairs > 1: every pair runs the single-pair schedule -- its encoder thread takes the next job of the stream, encodes
        it and queues it for the pair's decoder thread -- and the caller's generator hands the results out in job order.  The jobs
        iterator is advanced under a lock by whichever encoder thread is free (so a lazily built job, e.g. the harness's padding, is
        prepared on the stream that will consume it); at most n_pairs * (queue_depth + 1) jobs are taken beyond the last one handed out.
        `on_encoded` runs on the encoder threads."""
        import torch
        if not self._busy.acquire(blocking=False):
            raise PipelineError("this CodecPipeline is already running a code() loop (one schedule per pipeline object)")
        it = iter(jobs)
        cv = threading.Condition()
        st = {"next": 0, "out": 0, "done": False, "err": None, "stop": False}
        results = {}
        window = self.n_pairs * (self.queue_depth + 1)
        dev = self.device
        caller = torch.cuda.current_stream(dev)
        queues = [queue.Queue(maxsize=self.queue_depth) for _ in self.pairs]

        def fail(e):
            with cv:
                if st["err"] is None:
                    st["err"] = e
                st["stop"] = True
                cv.notify_all()

        def take():
            with cv:
                while not st["stop"] and not st["done"] and st["next"] - st["out"] >= window:
                    cv.wait(timeout=1.0)
                if st["stop"] or st["done"]:
                    return None
                try:
                    job = next(it)
                except StopIteration:
                    st["done"] = True
                    cv.notify_all()
                    return None
                seq = st["next"]
                st["next"] += 1
                return seq, job

        def encoder(r):
            enc_net, _, s_enc, _ = self.pairs[r]
            try:
                torch.cuda.set_device(dev)
                s_enc.wait_stream(caller)
                with torch.cuda.stream(s_enc):
                    while True:
                        t = take()
                        if t is None:
                            break
                        seq, job = t
                        enc = self._encode(enc_net, job)
                        if on_encoded is not None:
                            on_encoded(job, enc)
                        queues[r].put((seq, job, enc))
            except BaseException as e:
                fail(e)
            finally:
                queues[r].put(None)

        def decoder(r):
            _, dec_net, _, s_dec = self.pairs[r]
            try:
                torch.cuda.set_device(dev)
                with torch.cuda.stream(s_dec):
                    while True:
                        item = queues[r].get()
                        if item is None:
                            return
                        if st["stop"]:
                            continue                       # drain: the producer must never block on a dead consumer
                        seq, job, enc = item
                        e0 = None
                        if job.get("time_decode"):
                            e0 = torch.cuda.Event(enable_timing=True)
                            e0.record(s_dec)
                        dec = self._decode(dec_net, job, enc)
                        ev = torch.cuda.Event(enable_timing=e0 is not None)
                        ev.record(s_dec)
                        if e0 is not None:
                            job["_dec_events"] = (e0, ev)
                        with cv:
                            results[seq] = (job, enc, dec, ev)
                            cv.notify_all()
            except BaseException as e:
                fail(e)
                while queues[r].get() is not None:
                    pass

        threads = [threading.Thread(target=encoder, args=(r,), name=f"pcodec-encoder-{r}", daemon=True) for r in range(self.n_pairs)] + \
                  [threading.Thread(target=decoder, args=(r,), name=f"pcodec-decoder-{r}", daemon=True) for r in range(self.n_pairs)]
        for t in threads:
            t.start()
        try:
            while True:
                with cv:
                    while st["out"] not in results and st["err"] is None and not (st["done"] and st["out"] >= st["next"]):
                        if not cv.wait(timeout=600):
                            raise PipelineError("the pipeline made no progress for 600 s")
                    if st["err"] is not None:
                        raise st["err"]
                    if st["out"] not in results:
                        break                              # every job taken has been handed out and the stream is exhausted
                    job, enc, dec, ev = results.pop(st["out"])
                    st["out"] += 1
                    cv.notify_all()
                cur = torch.cuda.current_stream(dev)
                cur.wait_event(ev)
                for d in (dec if isinstance(dec, (list, tuple)) else [dec]):
                    d["x_hat"].record_stream(cur)
                for e in (enc if isinstance(enc, (list, tuple)) else [enc]):
                    for m in e.get("masks", []):
                        m.record_stream(cur)
                yield job, enc, dec
        finally:
            with cv:
                st["stop"] = True
                cv.notify_all()
            for t in threads:
                t.join(timeout=600)
            for _, _, s_e, s_d in self.pairs:
                caller.wait_stream(s_e)
                caller.wait_stream(s_d)
            self._busy.release()
            if any(t.is_alive() for t in threads):
                raise PipelineError("a pipeline thread did not finish")

    @staticmethod
    def decode_ms(job):
        """Milliseconds the decode stream spent on a job that was submitted with ``"time_decode": True`` (from the end of the item before
        it -- or the moment it was queued on an idle stream -- to the end of its own x_hat; host entropy decoding inside the call
        included).  Waits for the job's decode to finish."""
        e0, e1 = job["_dec_events"]
        e1.synchronize()
        return e0.elapsed_time(e1)

    def run(self, jobs):
        """code() collected into a list of (job, enc, dec)."""
        return list(self.code(jobs))

    # ------------------------------------------------------------------ measurement aid (bench.py roofline leg)
    def profile_conv_in_schedule(self, jobs):
        """Run `jobs` through code() with every MFMA-conv launch of BOTH objects bracketed by HIP events, the schedule left alone
        (pc_codec_set_option "profile_in_schedule"), and fold the two objects' launch intervals into one timeline.  Returns a dict:
        launches, algorithmic FLOPs, conv-busy ms (time during which at least one conv kernel runs), the sum of the launch durations,
        the window (first start .. last end) and the mean number of conv kernels in flight."""
        import ctypes as C
        import numpy as np
        import torch
        from ._lib import check, lib
        L = lib()
        check(L.pc_profile_set_epoch(self.device.index or 0), "pc_profile_set_epoch")
        for net in self.objects:
            net.set_option("profile_in_schedule", 1)
            check(L.pc_codec_profile_begin(net._h), "pc_codec_profile_begin")
        try:
            n_jobs = sum(1 for _ in self.code(jobs))
            torch.cuda.synchronize(self.device)
        finally:
            iv = []
            tot_fl = tot_by = 0.0
            for net in self.objects:
                nl, ms, fl, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
                check(L.pc_codec_profile_end(net._h, C.byref(nl), C.byref(ms), C.byref(fl)), "pc_codec_profile_end")
                check(L.pc_codec_profile_bytes(net._h, C.byref(by)), "pc_codec_profile_bytes")
                n = C.c_size_t()
                check(L.pc_codec_profile_intervals(net._h, None, None, None, 0, C.byref(n)), "pc_codec_profile_intervals")
                t0, t1, f = (np.zeros(max(1, n.value)) for _ in range(3))
                check(L.pc_codec_profile_intervals(net._h, t0.ctypes.data_as(C.c_void_p), t1.ctypes.data_as(C.c_void_p), f.ctypes.data_as(C.c_void_p),
                                                   n.value, C.byref(n)), "pc_codec_profile_intervals")
                iv.append((t0[:n.value], t1[:n.value], f[:n.value]))
                tot_fl += fl.value
                tot_by += by.value
                net.set_option("profile_in_schedule", 0)
        t0 = np.concatenate([a for a, _, _ in iv])
        t1 = np.concatenate([b for _, b, _ in iv])
        out = fold_intervals(t0, t1)
        out.update({"jobs": n_jobs, "launches": int(t0.size), "algorithmic_flops": tot_fl, "algorithmic_bytes": tot_by})
        return out


def fold_intervals(t0, t1):
    """Union length of the intervals [t0[i], t1[i]) (ms), their summed length, the window they span and the mean number in flight."""
    import numpy as np
    if len(t0) == 0:
        return {"busy_ms": 0.0, "sum_ms": 0.0, "window_ms": 0.0, "mean_in_flight": 0.0}
    order = np.argsort(t0, kind="stable")
    a, b = np.asarray(t0)[order], np.asarray(t1)[order]
    busy, cur0, cur1 = 0.0, a[0], b[0]
    for s, e in zip(a[1:], b[1:]):
        if s > cur1:
            busy += cur1 - cur0
            cur0, cur1 = s, e
        elif e > cur1:
            cur1 = e
    busy += cur1 - cur0
    total = float((b - a).sum())
    return {"busy_ms": float(busy), "sum_ms": total, "window_ms": float(b.max() - a.min()), "mean_in_flight": float(total / busy) if busy > 0 else 0.0}
