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

from cfen_vit_dehazing_amd.parallel import shard_range, even_shard, split_slab, merge_gathered, OutputGatherer

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_covers_batch():
    for total in (1, 7, 8, 64, 13):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_uneven_global_batch_is_refused_before_the_collective():
    assert even_shard(64, 8, 3) == (24, 32)
    with pytest.raises(ValueError):
        even_shard(13, 8, 0)
    g = OutputGatherer(1, 16, "cpu")
    with pytest.raises(ValueError):
        g.launch(torch.zeros(15), 0)


def _gpu_worker(rank, world, port, q):
    """the CUDA branch of OutputGatherer (side stream, event hand-off, fp16 stage conversion, async RCCL all_gather_into_tensor)"""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    n, B = 64, 2
    g = torch.Generator(); g.manual_seed(7)
    slabs = [(torch.rand(7 * B * n * n, generator=g) * 2 - 1) for _ in range(world)]     # every rank can rebuild all slabs
    mine = slabs[rank].to(dev)
    worst = 0.0
    lanes = [torch.cuda.Stream(dev) for _ in range(3)]
    for wire in (torch.float32, torch.float16):
        # bench.py's rotation: three slabs on three lane streams, one gather slot per slab, launch() in the lane's context right behind the "forward"
        # (here: a rewrite of the slab with values that depend on the step, so a gather that read a slab or a stage too late or too early is seen)
        og = OutputGatherer(world, mine.numel(), dev, wire, slots=3)
        mines = [torch.empty_like(mine) for _ in range(3)]
        steps = 8
        for it in range(steps):
            slot = it % 3
            with torch.cuda.stream(lanes[slot]):
                og.before_write(slot)
                mines[slot].copy_(slabs[rank].to(dev) * (1.0 - it / 16.0))     # exact in fp16 and fp32: a power-of-two step
                og.launch(mines[slot], slot)
        for s in lanes:
            with torch.cuda.stream(s):
                og.wait_all()
        torch.cuda.synchronize()
        for it in range(steps - 3, steps):
            out = og.bufs[it % 3]
            want = torch.cat(slabs).to(dev) * (1.0 - it / 16.0)
            err = float((out.float() - want).abs().max())
            assert out.dtype == wire and err <= (0.0 if wire == torch.float32 else 2.0 ** -11), (wire, it, err)
            worst = max(worst, err)
    if rank == 0:
        q.put(worst)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2])
def test_output_gatherer_cuda_branch_over_rccl(world):
    """world 1: a single-rank RCCL communicator drives the whole CUDA branch (side stream, event hand-off, wire conversion) on the one-GPU
    test box; world 2: two ranks over xGMI -- skipped with the reason when the box shows fewer than two GPUs, so a run on a multi-GPU
    node exercises rank ordering across devices"""
    if torch.cuda.device_count() < world:
        pytest.skip("needs %d GPUs, this box shows %d" % (world, torch.cuda.device_count()))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 17) % 2000
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=5) <= 2.0 ** -11


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


def _rotation_worker(rank, world, port, q, slab_dtype=torch.float32):
    """bench.py's step loop of the sharded run on CPU tensors: `nslab` = 3 output slabs rotate (three forwards in flight per rank) and the gatherer has
    one slot per slab (slot = step % nslab), so the fence in front of step i is the gather of step i - nslab, the last reader of that slab"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, n, nslab, steps = 2, 8, 3, 7
    slabs = [torch.empty(7 * B * n * n, dtype=slab_dtype) for _ in range(nslab)]      # fp16 slabs: what dec_ipt.output_f16 writes -- gathered as they are (round 6)
    g = OutputGatherer(world, slabs[0].numel(), "cpu", torch.float16, slots=nslab)
    assert len(g.bufs) == len(g.stage) == len(g.work) == nslab
    for i in range(steps):
        s = slabs[i % nslab]
        g.before_write(i % nslab)
        s.copy_(torch.arange(s.numel(), dtype=torch.float32) % 31 / 64 + rank / 4 + i / 128)      # what step i of this rank "computed" (fp16-exact values)
        g.launch(s, i % nslab)
        assert g.direct[i % nslab] == (slab_dtype == torch.float16)                                # the collective reads the slab itself only when it has the wire type
    g.wait_all()
    ok = True
    for back in range(nslab):          # the last nslab steps each still sit in their own slot
        i = steps - 1 - back
        merged = merge_gathered(g.bufs[i % nslab].float(), world, B, n)
        want = [torch.arange(slabs[0].numel(), dtype=torch.float32) % 31 / 64 + r / 4 + i / 128 for r in range(world)]
        ok = ok and all(torch.equal(torch.cat([split_slab(want[r], B, n)[k] for r in range(world)], 0), merged[k]) for k in range(3))
    if rank == 0:
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("slab_dtype", [torch.float32, torch.float16])
def test_slab_rotation_of_the_sharded_bench_loop_gathers_the_last_step(slab_dtype):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000 + (7 if slab_dtype == torch.float16 else 0)
    procs = [ctx.Process(target=_rotation_worker, args=(r, 2, port, q, slab_dtype)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _harness_worker(rank, world, port, root, q):
    """what `python -m torch.distributed.run --nproc-per-node 2 test.py ...` does before its first GPU call: the options pick the GPU of
    LOCAL_RANK, the dataset is this rank's contiguous slice"""
    sys.path.insert(0, ROOT)
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    from cfen_vit_dehazing_amd.options.test_options import TestOptions
    from cfen_vit_dehazing_amd import data as cdata
    opt = TestOptions().parse(['--dataroot', os.path.join(root, 'data'), '--checkpoints_dir', os.path.join(root, 'ckpt'), '--name', 'shard_unit',
                               '--n_feats', '24', '--hidden_dim_ratio', '4', '--sb', '--gpu_ids', '0,1', '--how_many', '6'])
    assert (opt.dist_rank, opt.dist_world, opt.gpu_ids) == (rank, world, [rank])       # one GPU per process, whatever --gpu_ids said
    mine = [os.path.basename(p) for b in cdata.CreateDataLoader(opt).load_data() for p in b['B_paths']]
    dist.init_process_group("gloo", rank=rank, world_size=world)
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    if rank == 0:
        q.put(everyone)
    dist.barrier()
    dist.destroy_process_group()


def test_test_py_shards_the_dataset_over_ranks(tmp_path):
    """world-size-2 gloo run of the harness's sharding: the two ranks' slices are contiguous, disjoint, in order and cover exactly the
    first --how_many images; the options file is written once"""
    import numpy as np
    from PIL import Image
    hazy = tmp_path / 'data' / 'hazy'
    hazy.mkdir(parents=True)
    names = ['img_%04d.png' % i for i in range(7)]
    for nm in names:
        Image.fromarray(np.zeros((8, 8, 3), dtype=np.uint8)).save(hazy / nm)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 911) % 2000
    procs = [ctx.Process(target=_harness_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    a, b = q.get(timeout=5)
    assert a + b == names[:6] and len(a) == len(b) == 3
    assert os.path.exists(tmp_path / 'ckpt' / 'shard_unit' / 'opt.txt')


def test_gatherer_refuses_unequal_slabs_instead_of_hanging():
    """two gloo ranks that bring slabs of different sizes: the constructor's size agreement raises on both"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 1222) % 2000
    procs = [ctx.Process(target=_unequal_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(2)) == ["refused", "refused"]


def _unequal_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        OutputGatherer(world, 16 + rank, "cpu")
        q.put("built")
    except ValueError:
        q.put("refused")
    dist.barrier()
    dist.destroy_process_group()
