"""The pipelined inference driver behind `test.py --in_flight K` (K > 1).

The reference's loop (test.py:33-63 over DataLoader(batch_size=opt.batchSize, num_workers=nThreads), data/__init__.py:41-48) handles one batch at
a time: decode -> H2D -> forward -> D2H -> PNG encode, each waiting for the one before.  bench.py's throughput presupposes something else: several
forwards in flight on launch-plan replicas (hipnet.dec_ipt.replica: own workspace, shared packed weights), replayed from hipGraphs, one stream and one
hardware queue each.  This module is that loop for real files:

  DataLoader workers (--nThreads)   PNG decode -> (B,H,W,3) uint8 (with --u8_input) or normalised float batches
  main thread, slot k = batch % K   pinned host buffer -> cudaMemcpyAsync H2D -> graph replay of replica k -> tensor2im bytes (written by the tails' last
                                    launch where the net does that itself, by the harness's own device pass otherwise) -> cudaMemcpyAsync D2H into pinned
                                    memory -> event; all on the slot's stream, nothing waits on the host until the slot comes round again
  writer threads (--writers)        PNG encode + file write of a finished slot (PIL releases the GIL while it compresses)

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
        own = {lab: np.array(s.cur.host[lab].numpy()) for lab in self.labels}      # one copy per visual: the pinned buffers are reused as soon as this returns
        for i, path in enumerate(s.paths):
            name = os.path.splitext(ntpath.basename(path))[0]
            for lab in self.labels:
                self.pending.append(self.pool.submit(_save_png, own[lab][i], os.path.join(self.image_dir, '%s_%s.png' % (name, lab))))
        s.paths = None
        if len(self.pending) > 4096:
            self._reap(keep=1024)

    def _reap(self, keep=0):
        while len(self.pending) > keep:
            self.pending.pop(0).result()                      # re-raises a writer's exception here

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
                    for lab in self.labels:
                        b.host[lab].copy_(self._visual_u8(s, lab, B), non_blocking=True)
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
        return self.stats

    def close(self):
        self._reap()
        self.pool.shutdown(wait=True)
        self.net.replica = 0
        self.net.serial_plan = self._plan_was
