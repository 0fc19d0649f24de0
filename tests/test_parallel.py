"""Multi-process data-parallel path on CPU (gloo, world_size 2): sharding + one all-gather of the
per-rank output slab must reproduce the single-process result in rank order.  The forward itself is
stood in for by the CPU oracle here (the HIP forward needs a GPU); the sharding / gather / merge code is
exactly what bench.py and the multi-GPU harness run with the `nccl` (RCCL) backend."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cfen_vit_dehazing_amd.parallel import shard_range, split_slab, merge_gathered, OutputGatherer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_covers_batch():
    for total in (1, 7, 8, 64, 13):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cfen_oracle
    from cfen_vit_dehazing_amd.config import NetConfig
    from cfen_vit_dehazing_amd.manifest import generate_state_dict, synthetic_input
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    cfg = NetConfig(24, 2, patch_size=8, load_size=64)
    sd = generate_state_dict(cfg, seed=0, with_dead=False)
    total, n = 4, cfg.image_size
    x = synthetic_input(total, cfg)
    lo, hi = shard_range(total, world, rank)
    with torch.no_grad():
        outs = cfen_oracle.forward(sd, x[lo:hi], cfg.num_heads, cfg.patch_size)
    B = hi - lo
    slab = torch.cat([o.reshape(-1) for o in outs])
    assert [tuple(v.shape) for v in split_slab(slab, B, n)] == [(B, 3, n, n), (B, 1, n, n), (B, 3, n, n)]
    g = OutputGatherer(world, slab.numel(), "cpu")
    g.before_write(0)
    gathered = g.launch(slab, 0)
    g.wait_all()
    merged = merge_gathered(gathered, world, B, n)
    # fp16 wire type (what bench.py uses with the fp16 compute path): converted on the way out, 2^-11 rounding of values in (-1, 1)
    g16 = OutputGatherer(world, slab.numel(), "cpu", torch.float16)
    g16.before_write(1)
    merged16 = merge_gathered(g16.launch(slab, 1), world, B, n)
    g16.wait_all()
    if rank == 0:
        with torch.no_grad():
            whole = cfen_oracle.forward(sd, x, cfg.num_heads, cfg.patch_size)
        err16 = max(float((a.float() - b).abs().max()) for a, b in zip(merged16, whole))
        assert merged16[0].dtype == torch.float16 and err16 <= 2.0 ** -11, err16
        q.put(max(float((a - b).abs().max()) for a, b in zip(merged, whole)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_allgather_equals_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=5) <= 1e-5
