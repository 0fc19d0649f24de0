"""The pipelined inference driver behind `test.py --in_flight K` (K > 1).

The reference's loop (test.py:33-63 over DataLoader(batch_size=opt.batchSize, num_workers=nThreads), data/__init__.py:41-48) handles one batch at
a time: decode -> H2D -> forward -> D2H -> PNG encode, each waiting for the one before.  bench.py's throughput presupposes something else: several
forwards in flight on launch-plan replicas (hipnet.dec_ipt.replica: own workspace, shared packed weights), replayed from hipGraphs, one stream and one
hardware queue each.  This module is that loop for real files:

  DataLoader workers (--nThreads)   PNG decode -> (B,H,W,3) uint8 (with --u8_input) or normalised float batches
  main thread, slot k = batch % K   pinned host buffer -> cudaMemcpyAsync H2D -> graph replay of replica k -> tensor2im bytes (written by the tails' last
                                    launch where the net does that itself, by the harness's own device pass otherwise) -> cudaMemcpyAsync D2H into pinned
                                    memory -> event; all on the slot's stream, nothing waits on the host until the slot comes round again
  writers (--writers N)             PNG encode + file write of a finished slot: N threads (PIL releases the GIL while it compresses, but 32 threads measured 527 images/s =
                                    16 each against 33 for one alone: GIL contention around the compressor), or with --writer_procs N processes forked BEFORE the model
                                    exists (start_writer_processes) that take the images from a ring of shared-memory slots -- encode then scales with the host's cores

Same files, byte for byte, as the sequential loop (tests/test_hip_net.py::test_pipelined_cli_writes_the_same_pngs): the arithmetic is the same launch plan
on the same weights (replicas differ in workspace only), and the bytes of a visual come from the same device passes (`util.tensor2im`'s arithmetic).
`--precision half` checks (models/model_iid_dehazing.py: first batch and every --half_guard_every-th) run through the model's own sequential path
after the pipeline has drained, so a fallback to fp32 is decided exactly as in the sequential loop.
"""
import ntpath
import os
import time
from collections import OrderedDict
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import ops
from .util import util


def _save_png(arr, path):
    util.save_image(arr, path)


# ---- writer PROCESSES (round 6): forked right after option parsing (test.py), like the DataLoader's decode workers -- forking a process that holds its GPU working set
# stalls the next launch 10-16 s on MI355X / ROCm 7.2.  Images travel through an anonymous shared mapping made before the fork (a ring of image-sized slots): the parent
# copies a finished image into a free slot (0.8 MB memcpy) and sends (slot, shape, path); a slot returns to the free list when its file is written, which also bounds the
# backlog in bytes (the thread pool's 4096-future bound let ~3 GB of image copies queue up).
_WRITERS = None


def _save_png_from_ring(off, shape, path, token):
    arr = np.frombuffer(_WRITERS["ring"], dtype=np.uint8, count=int(np.prod(shape)), offset=off).reshape(shape)
    util.save_image(arr, path)
    return token


def start_writer_processes(n, image_edge, batch=8, labels=1, in_flight=4, slots=0):
    """fork `n` PNG writer processes that read their images from one anonymous shared mapping made here, BEFORE the first HIP call:
      * `bslots` BATCH slots of labels x batch x (edge, edge, 3) uint8: the pipelined loop registers this part as pinned host memory (hipHostRegister) and lets the
        device-to-host copies land in it -- a writer encodes an image straight from where the DMA put it, the main thread copies nothing;
      * `slots` single-image slots behind them for the batches that go through the model's sequential path (--precision half checks): copied in by the main thread."""
    global _WRITERS
    import mmap
    import multiprocessing as mp
    if _WRITERS is not None or n <= 0:
        return _WRITERS
    img = int(image_edge) * int(image_edge) * 3
    bbytes = int(labels) * int(batch) * img
    bslots = max(2 * int(in_flight), min(4 * int(in_flight), (3 << 29) // bbytes))
    slots = int(slots) or 2 * int(batch) * int(labels)
    ring = mmap.mmap(-1, bslots * bbytes + slots * img)      # anonymous + shared: inherited by the forked workers
    _WRITERS = {"ring": ring, "img_bytes": img, "batch_bytes": bbytes, "bslots": bslots, "slots": slots, "slot0": bslots * bbytes, "n": n,
                "batch": int(batch), "labels": int(labels)}
    _WRITERS["pool"] = mp.get_context("fork").Pool(n)       # the workers read _WRITERS through the fork
    return _WRITERS


def stop_writer_processes():
    global _WRITERS
    if _WRITERS is not None:
        _WRITERS["pool"].close()
        _WRITERS["pool"].join()
        _WRITERS["ring"].close()
        _WRITERS = None


class _Buffers:
    """what a slot needs for ONE batch shape: pinned + device input, the graph that reads it, the device outputs the graph writes, pinned images"""
    def __init__(self):
        self.x_host = self.x_dev = None
        self.gid = None
        self.outs = None         # device outputs the graph writes: [xr, xs, xd]
        self.host = None         # label -> pinned (B,H,W,3) uint8


class _Slot:
    def __init__(self):
        self.by_shape = {}       # (tuple(batch shape), dtype) -> _Buffers (the full batch, and the ragged last one)
        self.cur = None          # _Buffers of the batch in flight
        self.event = None
        self.paths = None        # image paths of the batch in flight (None = slot idle)
        self.bslot = None        # writer processes: the batch slot of the shared ring this batch's images land in
        self.stream = None


class PipelinedRunner:
    def __init__(self, model, opt, image_dir):
        if not torch.cuda.is_available():
            raise RuntimeError("--in_flight needs a GPU")
        self.model, self.opt, self.image_dir = model, opt, image_dir
        self.net = model.netG
        self.K = int(opt.in_flight)
        self.dev = model.device
        self.slots = [_Slot() for _ in range(self.K)]
        for s in self.slots:
            s.stream = torch.cuda.Stream(self.dev)
        self.pool = ThreadPoolExecutor(max(1, int(getattr(opt, 'writers', 8))))
        self.pending = []
        # writer processes, if test.py forked them (--writer_procs): free ring slots + what came back from the workers
        self.procs = _WRITERS
        if self.procs is not None:
            import queue
            import threading
            p = self.procs
            self.free = queue.Queue()                          # single-image slots (sequential-path batches)
            for i in range(p["slots"]):
                self.free.put(i)
            self.bfree = queue.Queue()                         # batch slots (graph batches: the D2H copies land in them)
            for i in range(p["bslots"]):
                self.bfree.put(i)
            self.bleft = [0] * p["bslots"]                     # images of a batch slot still with the writers
            self.block = threading.Lock()
            self.proc_errors = []
            self.ring_t = torch.frombuffer(p["ring"], dtype=torch.uint8)
            nbytes = p["bslots"] * p["batch_bytes"]
            rc = torch.cuda.cudart().cudaHostRegister(self.ring_t.data_ptr(), nbytes, 0)
            self.ring_pinned = int(rc) == 0 and self.ring_t[:nbytes].is_pinned()
            self.stats_ring = {"batch_slots": p["bslots"], "registered_as_pinned": bool(self.ring_pinned)}
        self.labels = ['fake_A'] if opt.out_all else list(model.visual_names)
        self.stats = {"batches": 0, "images": 0, "graph_batches": 0, "sequential_batches": 0}
        # several forwards in flight want ONE serial chain of launches per forward (what bench.py replays): 2.26 against 2.38 ms per step with
        # three in flight, because a fork / join inside each of several concurrent graphs crosses hardware queues (DESIGN: launch plan)
        self._plan_was = self.net.serial_plan
        self.net.serial_plan = True
        self._dtype_built = None

    # ---- slot buffers + graph for a batch shape ------------------------------------------------------------------------------------------------
    def _build(self, k, batch):
        """buffers + graph of slot k for this batch shape (kept: a slot serves the full batch shape and the ragged last one)"""
        s = self.slots[k]
        b = _Buffers()
        self.net.replica = k
        self.net.output_u8 = bool(getattr(self.model, '_u8_out', False))
        b.x_host = torch.empty(batch.shape, dtype=batch.dtype).pin_memory()
        b.x_dev = torch.empty(batch.shape, dtype=batch.dtype, device=self.dev)
        b.x_dev.copy_(batch)
        with torch.cuda.stream(s.stream):
            self.net(b.x_dev)                                  # builds this replica's plan (and packs the weights the first time)
            if self.net.output_u8 and not self.net.writes_u8_natively():
                # (fp32 plans, small images) the uint8 images are separate passes behind the forward: the graph holds the forward and writes float
                # outputs, _visual_u8 runs those passes behind every replay
                self.net.output_u8 = False
            b.gid, b.outs = self.net.capture(b.x_dev)
        s.stream.synchronize()
        self.net.replica = 0
        B = batch.shape[0]
        n = self.net.cfg.image_size
        b.host = {lab: torch.empty(B, n, n, 3, dtype=torch.uint8).pin_memory() for lab in self.labels}
        if s.event is None:
            s.event = torch.cuda.Event()
        s.by_shape[(tuple(batch.shape), batch.dtype)] = b
        self._dtype_built = self.net.compute_dtype
        return b

    def _reset(self):
        """the net dropped its plans or changed its compute type (a precision fallback): every slot is rebuilt at its next use"""
        for s in self.slots:
            s.by_shape = {}

    def warm_up(self, batch_shapes, dtype):
        """graphs, pinned buffers and workspaces of every slot for the batch shapes the run will see (the full batch and the ragged last one), before
        the DataLoader forks its workers (see DECHLGVIT.warm_up: device allocations with forked children alive take seconds each)"""
        for shape in batch_shapes:
            dummy = torch.zeros(shape, dtype=dtype)
            for k in range(self.K):
                self._build(k, dummy)

    def _visual_u8(self, s, lab, B):
        """device (B,H,W,3) uint8 of one visual of the batch slot `s` just computed, on the current (= the slot's) stream: the bytes util.tensor2im gives"""
        s = s.cur
        if lab == 'real_B':
            x = s.x_dev
            if x.dtype == torch.uint8:                         # what the model shows for --u8_input: the normalised float image (model_iid_dehazing.set_input)
                x = (x.permute(0, 3, 1, 2).float() / 255.0 - 0.5) / 0.5
            src = x
        else:
            src = s.outs[{'fake_R': 0, 'fake_S': 1, 'fake_A': 2}[lab]]
        if src.dtype == torch.uint8:
            return src
        return torch.stack([ops.tensor2im_u8(src[b].float().contiguous()) for b in range(B)])

    # ---- the loop ------------------------------------------------------------------------------------------------------------------------------
    def _retire(self, s):
        """slot's batch is on the host: hand its images to the writers"""
        if s.paths is None:
            return
        s.event.synchronize()
        if self.procs is not None and s.bslot is not None:
            # the images already sit in the slot's batch slot of the shared ring (the D2H copies landed there): the writers get (offset, shape, path), the batch slot
            # returns to the free list when its last file is written
            p, bs, B = self.procs, s.bslot, len(s.paths)
            with self.block:
                self.bleft[bs] = B * len(self.labels)
            n = self.net.cfg.image_size
            for li, lab in enumerate(self.labels):
                for i, path in enumerate(s.paths):
                    name = os.path.splitext(ntpath.basename(path))[0]
                    off = bs * p["batch_bytes"] + (li * p["batch"] + i) * p["img_bytes"]
                    p["pool"].apply_async(_save_png_from_ring, (off, (n, n, 3), os.path.join(self.image_dir, '%s_%s.png' % (name, lab)), bs),
                                          callback=self._batch_image_done, error_callback=lambda e, bs=bs: self._batch_image_failed(e, bs))
            s.paths = None
            s.bslot = None
            if self.proc_errors:
                raise self.proc_errors[0]
            return
        own = {lab: np.array(s.cur.host[lab].numpy()) for lab in self.labels}      # one copy per visual: the pinned buffers are reused as soon as this returns
        for i, path in enumerate(s.paths):
            name = os.path.splitext(ntpath.basename(path))[0]
            for lab in self.labels:
                self.pending.append(self.pool.submit(_save_png, own[lab][i], os.path.join(self.image_dir, '%s_%s.png' % (name, lab))))
        s.paths = None
        # the backlog is bounded by a few rounds of slots (it was 4096 futures = ~3 GB of image copies at 512 x 512: ADVICE r05), and finished futures are
        # checked at every retire so that a writer's exception surfaces at once
        bound = 8 * self.K * max(1, len(own[self.labels[0]])) * len(self.labels)
        while self.pending and self.pending[0].done():
            self.pending.pop(0).result()
        if len(self.pending) > bound:
            self._reap(keep=bound // 2)

    def _batch_image_done(self, bs):
        with self.block:
            self.bleft[bs] -= 1
            free = self.bleft[bs] == 0
        if free:
            self.bfree.put(bs)

    def _batch_image_failed(self, e, bs):
        self.proc_errors.append(e)
        self._batch_image_done(bs)

    def _host_views(self, bs, B):
        """label -> (B, n, n, 3) uint8 tensor over batch slot `bs` of the shared ring"""
        p, n = self.procs, self.net.cfg.image_size
        out = {}
        for li, lab in enumerate(self.labels):
            o = bs * p["batch_bytes"] + li * p["batch"] * p["img_bytes"]
            out[lab] = self.ring_t[o:o + B * p["img_bytes"]].view(B, n, n, 3)
        return out

    def _submit_proc(self, arr, path):
        """one image of a sequential-path batch to the writer processes: copied into a free single-image slot (blocks while all are in use), then (offset, shape, path)"""
        if self.proc_errors:
            raise self.proc_errors[0]
        slot = self.free.get()
        p = self.procs
        off = p["slot0"] + slot * p["img_bytes"]
        np.frombuffer(p["ring"], dtype=np.uint8, count=arr.size, offset=off).reshape(arr.shape)[...] = arr

        def fail(e, slot=slot):
            self.proc_errors.append(e)
            self.free.put(slot)
        p["pool"].apply_async(_save_png_from_ring, (off, tuple(arr.shape), path, slot), callback=self.free.put, error_callback=fail)

    def _reap(self, keep=0):
        while len(self.pending) > keep:
            self.pending.pop(0).result()                      # re-raises a writer's exception here
        if self.procs is not None and keep == 0:
            for q, count in ((self.free, self.procs["slots"]), (self.bfree, self.procs["bslots"])):
                held = []
                while len(held) < count:                        # every slot back in its free list = every file written
                    held.append(q.get())
                for i in held:
                    q.put(i)
            if self.proc_errors:
                raise self.proc_errors[0]

    def drain(self):
        for s in self.slots:
            self._retire(s)

    def _sequential(self, data, j):
        """one batch through the model's own path (the --precision half checks live there), files through the writers"""
        tm = self.stats.setdefault("sequential_seconds", {"drain": 0.0, "set_input": 0.0, "test": 0.0, "images": 0.0})
        t0 = time.perf_counter()
        self.drain()
        t1 = time.perf_counter()
        self.net.replica = 0
        self.model._batch_index = j
        self.model.set_input(data)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        self.model.test(self.opt)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        visuals = self.model.get_current_visuals()
        paths = self.model.get_image_paths()
        for i, path in enumerate(paths):
            name = os.path.splitext(ntpath.basename(path))[0]
            for lab in self.labels:
                arr = util.tensor2im(visuals[lab][i, :, :, :])
                if self.procs is not None and arr.nbytes <= self.procs["img_bytes"]:
                    self._submit_proc(np.ascontiguousarray(arr), os.path.join(self.image_dir, '%s_%s.png' % (name, lab)))
                else:
                    self.pending.append(self.pool.submit(_save_png, arr, os.path.join(self.image_dir, '%s_%s.png' % (name, lab))))
        self.stats["sequential_batches"] += 1
        tm["drain"] += t1 - t0; tm["set_input"] += t2 - t1; tm["test"] += t3 - t2; tm["images"] += time.perf_counter() - t3
        if self._dtype_built is not None and self.net.compute_dtype != self._dtype_built:
            self._reset()

    def run(self, dataset, how_many=float('inf')):
        t0 = time.perf_counter()
        j = 0
        tm = {"wait_data": 0.0, "retire": 0.0, "build": 0.0, "stage_in": 0.0, "enqueue": 0.0, "sequential": 0.0}
        self.stats["main_thread_seconds"] = tm
        clock = time.perf_counter
        t_prev = clock()
        for i, data in enumerate(dataset):
            tm["wait_data"] += clock() - t_prev
            if i >= how_many:
                break
            batch = data['B']
            paths = list(data['B_paths'])
            if self.model.guard_due(j):
                t1 = clock()
                self._sequential(data, j)
                tm["sequential"] += clock() - t1
            else:
                k = j % self.K
                s = self.slots[k]
                t1 = clock()
                self._retire(s)                                # waits for THIS slot's previous batch only
                t2 = clock()
                tm["retire"] += t2 - t1
                if self.net.compute_dtype != self._dtype_built:
                    self.drain()
                    self._reset()
                b = s.by_shape.get((tuple(batch.shape), batch.dtype)) or self._build(k, batch)
                s.cur = b
                t3 = clock()
                tm["build"] += t3 - t2
                src = batch if batch.is_pinned() else b.x_host.copy_(batch)     # the loader's pin thread has already staged it (CustomDatasetDataLoader: pin_memory)
                b.keep = batch                                                    # alive until the slot comes round again: the H2D copy reads it asynchronously
                B = batch.shape[0]
                t4 = clock()
                tm["stage_in"] += t4 - t3
                with torch.cuda.stream(s.stream):
                    b.x_dev.copy_(src, non_blocking=True)
                    self.net.replay(b.gid)
                    host = b.host
                    if self.procs is not None and B <= self.procs["batch"]:
                        s.bslot = self.bfree.get()              # blocks while the writers hold every batch slot: the backlog is the ring
                        host = self._host_views(s.bslot, B)
                    for lab in self.labels:
                        host[lab].copy_(self._visual_u8(s, lab, B), non_blocking=True)
                    s.event.record(s.stream)
                s.paths = paths
                self.model.note_unchecked(paths)
                self.stats["graph_batches"] += 1
                tm["enqueue"] += clock() - t4
            self.stats["batches"] += 1
            self.stats["images"] += len(paths)
            j += 1
            t_prev = clock()
        t1 = clock()
        self.drain()
        self._reap()
        torch.cuda.synchronize()
        tm["drain_and_writers"] = clock() - t1
        for k in tm:
            tm[k] = round(tm[k], 3)
        self.stats["seconds"] = time.perf_counter() - t0
        if self.procs is not None:
            self.stats["writer_ring"] = self.stats_ring
        return self.stats

    def close(self):
        self._reap()
        self.pool.shutdown(wait=True)
        if self.procs is not None and self.ring_pinned:
            torch.cuda.synchronize()
            torch.cuda.cudart().cudaHostUnregister(self.ring_t.data_ptr())
            self.ring_pinned = False
        self.net.replica = 0
        self.net.serial_plan = self._plan_was
