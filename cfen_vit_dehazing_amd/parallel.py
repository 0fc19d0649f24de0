"""Data-parallel sharding of image batches: one process per GPU, weights replicated, no exchange inside
the forward, ONE all-gather of the per-rank output slab (RCCL over xGMI; `nccl` backend == RCCL on ROCm).

The reference's only parallelism is single-process nn.DataParallel (scatter batch / gather outputs,
models/networks_iid_hlgvit_crs_gd4_cfs_v3.py:77-83); every op of the forward is per-sample once ActNorm
is initialised, so sharding dim 0 is exact.  Instead of a gather to one master GPU the
slab [xr | xs | xd] of each rank is all-gathered, and the collective of batch i runs on a side stream
underneath the forward of batch i+1.
"""
import torch
import torch.distributed as dist


def shard_range(total, world, rank):
    """Contiguous [lo, hi) slice of a global batch for `rank` (sizes differ by at most one)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def even_shard(total, world, rank):
    """shard_range for the all-gather path: all_gather_into_tensor and merge_gathered need equally sized per-rank slabs, so a
    global batch that does not divide by the world size is refused here instead of hanging in the collective."""
    if total % world:
        raise ValueError("global batch %d is not divisible by world size %d: OutputGatherer needs equal per-rank slabs "
                         "(pad the batch, or gather with per-rank sizes)" % (total, world))
    return shard_range(total, world, rank)


def dist_env():
    """(rank, world, local_rank) of a process started by `python -m torch.distributed.run`; (0, 1, 0) when started plainly.  The launcher exports
    RANK, WORLD_SIZE and LOCAL_RANK together (and TORCHELASTIC_RUN_ID): a scheduler that sets only some of them (SLURM / k8s job arrays export
    WORLD_SIZE or RANK for their own purposes) does NOT switch the sharded mode on.  Reads the environment only: safe before any GPU call."""
    import os
    env = os.environ
    if not (all(k in env for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK")) or "TORCHELASTIC_RUN_ID" in env):
        return 0, 1, 0
    return int(env.get("RANK", "0")), int(env.get("WORLD_SIZE", "1")), int(env.get("LOCAL_RANK", "0"))


def shard_items(items, world, rank):
    """the contiguous slice of a dataset's item list one rank works on (test.py under torch.distributed.run: every rank dehazes its own
    images and writes its own files, so slices may differ in length by one and no collective is needed)"""
    lo, hi = shard_range(len(items), world, rank)
    return items[lo:hi]


def split_slab(slab, batch, n):
    """flat [xr | xs | xd] slab of one rank -> views (B,3,n,n), (B,1,n,n), (B,3,n,n)."""
    px = batch * n * n
    return slab[:3 * px].view(batch, 3, n, n), slab[3 * px:4 * px].view(batch, 1, n, n), slab[4 * px:7 * px].view(batch, 3, n, n)


def merge_gathered(gathered, world, batch, n):
    """all-gathered buffer (world slabs back to back) -> global-batch tensors in rank order."""
    per = 7 * batch * n * n
    parts = [split_slab(gathered[r * per:(r + 1) * per], batch, n) for r in range(world)]
    return [torch.cat([p[i] for p in parts], 0) for i in range(3)]


class OutputGatherer:
    """Asynchronous all-gather of equally sized per-rank slabs, ONE SLOT PER OUTPUT SLAB of the caller's rotation.

    A slot owns what the collective of one slab needs: the fp16 stage buffer, the gathered buffer and the handle of the collective that last used them.
    `slots` is the number of slabs the compute lanes rotate over (bench.py: max(2, forwards in flight)), so step i uses slot i % slots and the only
    thing step i ever waits for is the gather of step i - slots, the last user of that slot -- with three slabs three forwards really are in flight.
    (Round 4 kept two slots for three slabs; forward i then waited for gather i - 2 and at most two forwards overlapped per rank: ADVICE r04.)

    No communication stream of its own (round 5): `launch` is called in the LANE's stream context right behind the forward.  The wire-type conversion
    runs on the lane (30 us, stream-ordered behind the forward, so the slab is free for the lane's next forward without any fence) and the collective is
    handed to torch.distributed asynchronously: ProcessGroupNCCL runs it on its own internal stream behind an event of the lane.  Busy hardware queues
    per rank = the lanes + that one stream.  Measured on one MI355X with a world-1 RCCL communicator (bench.py extra_configs.gather_overhead_1gpu): a
    separate communication stream in front of torch's internal one -- five busy queues -- cost 0.72 ms per step (2.21 -> 2.93 ms with three lanes).

    `dtype` is the wire type.  With the fp16 compute path the wire is fp16 (outputs are tanh values in (-1, 1): the 2^-11 rounding is below the fp16
    path's own error), which halves the xGMI traffic -- 29 MB instead of 59 MB per rank at 8 images of 512x512.  Since round 6 the fused tail launch
    writes that type itself (dec_ipt.output_f16): a slab that already has the wire type is gathered as it is; an fp32 slab is still converted on the
    lane first."""

    def __init__(self, world, numel, device, dtype=torch.float32, slots=2):
        if slots < 1:
            raise ValueError("OutputGatherer needs at least one slot")
        self.world, self.numel, self.dtype, self.slots = world, numel, dtype, slots
        self.cuda = torch.device(device).type == "cuda"
        if world > 1 and dist.is_initialized():
            # every rank must bring the same slab size: all_gather_into_tensor with unequal inputs does not fail, it hangs
            lohi = torch.tensor([numel, -numel], dtype=torch.int64, device=device)
            dist.all_reduce(lohi, op=dist.ReduceOp.MAX)
            if int(lohi[0]) != numel or int(-lohi[1]) != numel:
                raise ValueError("OutputGatherer: ranks hold slabs of %d .. %d elements; shard the global batch with parallel.even_shard"
                                 % (int(-lohi[1]), int(lohi[0])))
        self.bufs = [torch.empty(world * numel, dtype=dtype, device=device) for _ in range(slots)]
        self.stage = [torch.empty(numel, dtype=dtype, device=device) for _ in range(slots)] if dtype != torch.float32 else None   # wire-type copies of fp32 slabs
        self.direct = [True] * slots          # slot -> the collective last launched for it reads the caller's slab itself (no stage copy in between)
        self.work = [None] * slots            # slot -> handle of the collective that last read stage[slot] / the slab and wrote bufs[slot]

    def _settle(self, slot):
        """the current stream waits (on the device, not the host) until the collective that last used `slot` is done with its buffers"""
        if self.cuda and self.work[slot] is not None:
            self.work[slot].wait()
            self.work[slot] = None

    def before_write(self, slot):
        """Call on the lane's stream before the forward overwrites the slab last handed to launch(slot).  With a converting gatherer the slab was
        released by a copy on this same lane (nothing to wait for); an fp32 wire reads the slab itself."""
        if self.direct[slot]:
            self._settle(slot)

    def launch(self, slab, slot):
        """Call on the lane's stream right behind the forward that wrote `slab`.  Returns the gathered buffer (valid after wait_all / the next _settle)."""
        if slab.numel() != self.numel:
            raise ValueError("slab has %d elements, the gatherer was built for %d per rank (every rank must hand over the same size)"
                             % (slab.numel(), self.numel))
        convert = self.stage is not None and slab.dtype != self.dtype
        if not self.cuda:
            self.direct[slot] = not convert
            src = self.stage[slot].copy_(slab) if convert else slab
            dist.all_gather_into_tensor(self.bufs[slot], src)
            return self.bufs[slot]
        self._settle(slot)                                # stage[slot] / bufs[slot] are free again
        self.direct[slot] = not convert
        src = self.stage[slot].copy_(slab) if convert else slab
        self.work[slot] = dist.all_gather_into_tensor(self.bufs[slot], src, async_op=True)
        return self.bufs[slot]

    def wait_all(self):
        """the current stream waits for every collective still out"""
        for slot in range(self.slots):
            self._settle(slot)
