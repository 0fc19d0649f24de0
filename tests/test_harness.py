"""Host-side mirror of the reference's inference harness (options / data / BaseModel / save_images):
behaviour pinned by vectors captured from the reference (tensor2im) and by its documented conventions."""
import os
import sys

import numpy as np
import pytest
import torch
from PIL import Image

from cfen_vit_dehazing_amd.config import NetConfig
from cfen_vit_dehazing_amd.hipnet import dec_ipt
from cfen_vit_dehazing_amd.manifest import generate_state_dict, state_manifest
from cfen_vit_dehazing_amd.options.test_options import TestOptions
from cfen_vit_dehazing_amd.util import util
from cfen_vit_dehazing_amd.util.visualizer import save_images
from cfen_vit_dehazing_amd import data as cdata
from cfen_vit_dehazing_amd.models import create_model


def parse(tmp_path, extra=()):
    argv = ['--dataroot', str(tmp_path / 'data'), '--checkpoints_dir', str(tmp_path / 'ckpt'), '--results_dir', str(tmp_path / 'res'),
            '--name', 'iid_hlgvit_crs_gd4_cfs_v3_unit', '--n_feats', '24', '--hidden_dim_ratio', '4', '--sb', '--out_all'] + list(extra)
    return TestOptions().parse(argv)


def test_tensor2im_matches_reference_vectors(golden_dir):
    kat = np.load(golden_dir + "/ops_kat.npz")
    assert np.array_equal(util.tensor2im(torch.from_numpy(kat["t2i/x"])), kat["t2i/y"])      # truncation, 1-channel tiling
    assert np.array_equal(util.tensor2im(torch.from_numpy(kat["t2i/x3"])), kat["t2i/y3"])
    # an image the generator already wrote as tensor2im's bytes (dec_ipt.output_u8: (H,W,3) uint8) passes through unchanged
    img = torch.from_numpy(kat["t2i/y3"].copy())
    assert img.dtype == torch.uint8 and np.array_equal(util.tensor2im(img), kat["t2i/y3"])


def test_options_defaults_and_readme_command(tmp_path):
    opt = parse(tmp_path, ['--which_epoch', '32', '--some_training_flag', '7'])
    assert (opt.n_feats, opt.hidden_dim_ratio, opt.patch_size, opt.loadSize, opt.patch_dim, opt.num_heads) == (24, 4, 32, 256, 2, 4)
    assert opt.model == 'dec_vit' and opt.model_G == 'iid_hlgvit_crs_gd4_cfs_v3' and opt.dataset_mode == 'dec_vit'
    assert opt.gpu_ids == [0] and opt.isTrain is False and opt.which_epoch == '32' and opt.batchSize == 1
    assert os.path.exists(tmp_path / 'ckpt' / opt.name / 'opt.txt')                           # base_options.py:241-248
    d = TestOptions().parse(['--dataroot', 'x', '--checkpoints_dir', str(tmp_path / 'c2')])
    assert (d.n_feats, d.hidden_dim_ratio, d.how_many, d.which_epoch, d.phase) == (32, 6, 924, 'latest', 'test')


def test_dataset_reads_hazy_folder_in_order(tmp_path):
    hazy = tmp_path / 'data' / 'hazy'
    hazy.mkdir(parents=True)
    rs = np.random.RandomState(0)
    imgs = {}
    for nm in ('b_0002.png', 'a_0001.png', 'c.jpg.txt'):
        if nm.endswith('.txt'):
            (hazy / nm).write_text('not an image')
            continue
        a = rs.randint(0, 256, (16, 24, 3), dtype=np.uint8)
        Image.fromarray(a).save(hazy / nm)
        imgs[nm] = a
    opt = parse(tmp_path, ['--batchSize', '2'])
    loader = cdata.CreateDataLoader(opt).load_data()
    batches = list(loader)
    assert len(loader) == 2 and len(batches) == 1
    b = batches[0]
    assert [os.path.basename(p) for p in b['B_paths']] == ['a_0001.png', 'b_0002.png']            # sorted (dec_vit_data.py:32)
    want = torch.from_numpy(imgs['a_0001.png']).permute(2, 0, 1).float() / 255 * 2 - 1
    assert b['B'].shape == (2, 3, 16, 24) and torch.allclose(b['B'][0], want, atol=1e-6)
    with pytest.raises(ValueError):
        opt.dataset_mode = 'vit'
        cdata.CreateDataset(opt)


def test_save_images_naming(tmp_path):
    from collections import OrderedDict
    vis = OrderedDict(fake_A=torch.zeros(2, 3, 4, 4), fake_S=torch.ones(2, 1, 4, 4) * 0.5)
    save_images(str(tmp_path), vis, ['/x/y/img_0001.png', 'C:\\d\\img_0002.jpg'])
    assert sorted(os.listdir(tmp_path)) == ['img_0001_fake_A.png', 'img_0001_fake_S.png', 'img_0002_fake_A.png', 'img_0002_fake_S.png']
    a = np.asarray(Image.open(tmp_path / 'img_0001_fake_A.png'))
    assert a.shape == (4, 4, 3) and int(a[0, 0, 0]) == 127                                       # (0+1)/2*255 = 127.5 -> 127
    s = np.asarray(Image.open(tmp_path / 'img_0002_fake_S.png'))
    assert s.shape == (4, 4, 3) and int(s[0, 0, 0]) == 191


def test_module_state_dict_is_reference_compatible(tmp_path):
    cfg = NetConfig(8, 2, patch_size=8, load_size=64)              # small n_feats: same key structure, 30x fewer bytes
    net = dec_ipt(cfg)
    keys = [k for k, _, _ in state_manifest(cfg)]
    assert list(net.state_dict().keys()) == keys                   # same keys, same order as dec_ipt(opt).state_dict()
    sd = generate_state_dict(cfg)
    net.load_state_dict(sd, strict=True)
    for k, v in net.state_dict().items():
        assert torch.equal(v, sd[k]), k
    bad = dict(sd)
    bad.pop('head.0.0.weight')
    with pytest.raises(RuntimeError):
        net.load_state_dict(bad, strict=True)                      # strict, like base_model.py:131
    bad = dict(sd)
    bad['head.0.0.weight'] = torch.zeros(4, 3, 3, 3)
    with pytest.raises(RuntimeError):
        net.load_state_dict(bad)


def test_create_model_errors_and_checkpoint_roundtrip(tmp_path):
    opt = parse(tmp_path, ['--loadSize', '64', '--patch_size', '8', '--gpu_ids', '-1'])
    opt.n_feats, opt.hidden_dim_ratio = 8, 2                       # keep the checkpoint files small
    opt.model = 'vit'
    with pytest.raises(NotImplementedError):
        create_model(opt)                                          # models/__init__.py:26
    opt.model = 'dec_vit'
    model = create_model(opt)
    assert model.name() == 'DECHLGVIT' and model.model_names == ['G'] and model.visual_names == ['fake_A', 'real_B', 'fake_R', 'fake_S']
    assert str(model.device) == 'cpu' and model.save_dir == os.path.join(opt.checkpoints_dir, opt.name)
    with pytest.raises(FileNotFoundError):
        model.setup(opt)                                           # no checkpoint yet
    sd = generate_state_dict(NetConfig(8, 2, patch_size=8, load_size=64), seed=3)
    torch.save(sd, os.path.join(model.save_dir, 'latest_net_G.pth'))
    model.setup(opt)
    assert torch.equal(model.netG.state_dict()['tail_S.0.4.weight'], sd['tail_S.0.4.weight'])
    model.save_networks('7')
    again = torch.load(os.path.join(model.save_dir, '7_net_G.pth'))
    assert list(again.keys()) == list(sd.keys()) and torch.equal(again['head.0.0.bias'], sd['head.0.0.bias'])
    # CPU tensors are refused loudly: there is no fallback path
    model.set_input({'B': torch.zeros(1, 3, 128, 128), 'B_paths': ['x.png']})
    with pytest.raises(Exception):
        model.test(opt)
    opt.model_G = 'iid_cnn_crs'
    m2 = create_model(opt)
    assert not hasattr(m2, 'netG')                                 # reference: silently undefined -> AttributeError later


@pytest.mark.parametrize("model_g,variant,nkeys", [("iid_hlgvit_crs_gd4_cfs_v3", "v3", 958), ("iid_hlgvit_crs_gd4_cfs", "cfs", 934),
                                                   ("iid_hlgvit_crs_gd4", "crs", 950), ("iid_hlgvit_crs_gd4_cfs_v5", "v5", 1078)])
def test_model_G_selects_the_generator_variant(tmp_path, model_g, variant, nkeys):
    """models/model_iid_dehazing.py:50-95: --model_G picks the generator; the four built ones give a module with the reference's state_dict
    (checkpoint round trip incl. the never-read entries: decoder.*, query_embed, sub/add_mean, crs_gd4's SpatialPyramid)"""
    opt = parse(tmp_path, ['--loadSize', '64', '--patch_size', '8', '--gpu_ids', '-1', '--model_G', model_g])
    opt.n_feats, opt.hidden_dim_ratio = 8, 2
    model = create_model(opt)
    cfg = model.netG.cfg
    assert cfg.variant == variant and cfg.image_size == (64 if variant in ("cfs", "crs") else 128)
    sd = generate_state_dict(cfg, seed=5)
    assert len(sd) == nkeys
    torch.save(sd, os.path.join(model.save_dir, 'latest_net_G.pth'))
    model.setup(opt)
    back = model.netG.state_dict()
    assert list(back.keys()) == list(sd.keys())
    for k in sd:
        assert torch.equal(back[k], sd[k]), k


# ---- round 5: the --precision half checks (first batch + every N-th), the pipelined driver's options ------------------------------------------

class _FakeNet:
    """stands in for hipnet.dec_ipt where only the guard's bookkeeping is under test (no GPU)"""
    compute_dtype = torch.float16
    output_u8 = False

    def set_compute_dtype(self, d):
        self.compute_dtype = {"fp32": torch.float32, "fp16": torch.float16}[d]


def _guard_model(tmp_path, every, world=1, rank=0):
    from types import SimpleNamespace
    from cfen_vit_dehazing_amd.models.model_iid_dehazing import DECHLGVIT
    m = DECHLGVIT.__new__(DECHLGVIT)
    m.opt = SimpleNamespace(precision='half', no_half_guard=False, half_guard_every=every, dist_world=world, dist_rank=rank, isTrain=False,
                            phase='test', results_dir=str(tmp_path / 'res'), name='guard_unit', which_epoch='latest')
    m.device = torch.device('cpu')
    m.netG = _FakeNet()
    m._half_guard = True
    m._guard_every = every
    m._batch_index, m._since_check, m.redo_paths, m.half_guard_log, m._checks_planned, m._checks_done = 0, [], [], [], None, 0
    return m


def test_half_guard_schedule_first_batch_and_every_nth(tmp_path):
    m = _guard_model(tmp_path, every=4)
    assert [j for j in range(13) if m.guard_due(j)] == [0, 4, 8, 12]
    assert [m.checks_for(n) for n in (0, 1, 4, 5, 8, 9)] == [0, 1, 1, 2, 2, 3]
    m0 = _guard_model(tmp_path, every=0)
    assert [j for j in range(9) if m0.guard_due(j)] == [0] and m0.checks_for(100) == 1
    m.plan_half_guard(9)
    assert m._checks_planned == 3
    # batches noted as unchecked become redo work when a later check fails, and the model leaves fp16
    m.note_unchecked(['a.png', 'b.png'])
    m.note_unchecked(['c.png'])
    m._fall_back(0.4, batch_index=4)
    assert m.redo_paths == ['a.png', 'b.png', 'c.png'] and m.netG.compute_dtype == torch.float32 and not m.guard_due(8)
    txt = open(tmp_path / 'res' / 'guard_unit' / 'test_latest' / 'precision.txt').read()
    assert 'precision: single' in txt and 'fell_back_at_batch: 4' in txt


def _guard_rank(rank, world, port, tmp, q):
    import torch.distributed as dist
    from pathlib import Path
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _guard_model(Path(tmp), every=2, world=world, rank=rank)
    # without a plan only the first check is common to all ranks
    assert [j for j in range(6) if m.guard_due(j)] == [0]
    nb = 5 if rank == 0 else 3                      # slices differ: rank 0 checks batches 0, 2, 4; rank 1 batches 0, 2
    m.plan_half_guard(nb)
    assert m._checks_planned == 3
    worsts = []
    for j in range(nb):
        if m.guard_due(j):
            mine = 1e-3 if not (rank == 0 and j == 4) else 0.5      # rank 0's LAST check fails, after rank 1 has finished its slice
            m._settle_check(j, mine)
            worsts.append(m.half_guard_max_abs)
        else:
            m.note_unchecked(['r%d_b%d.png' % (rank, j)])
    m.finish_half_guard()                            # rank 1 joins the third check here and learns of the failure
    q.put((rank, worsts, m.netG.compute_dtype == torch.float32, sorted(m.redo_paths)))
    dist.barrier()
    dist.destroy_process_group()


def test_half_guard_ranks_with_unequal_slices_agree_on_every_check(tmp_path):
    """world-size-2 gloo run of the guard's collectives (what test.py does under torch.distributed.run): the ranks pair up their checks although one
    holds more batches, and a failure on one rank's last check moves BOTH to fp32 and marks each rank's unchecked batches for redoing"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() + 333) % 2000
    procs = [ctx.Process(target=_guard_rank, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    res = dict((r, rest) for r, *rest in (q.get(timeout=5) for _ in range(2)))
    assert res[0][0] == [1e-3, 1e-3, 0.5] and res[1][0] == [1e-3, 1e-3]
    assert res[0][1] and res[1][1]                                  # both ranks ended in fp32
    assert res[0][2] == ['r0_b3.png'] and res[1][2] == []           # rank 1's batch 1 was covered by its passed check at batch 2


def test_in_flight_option_sets_the_hardware_queues_before_the_first_hip_call(tmp_path, monkeypatch):
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    opt = parse(tmp_path, [])
    assert opt.in_flight == 1 and "GPU_MAX_HW_QUEUES" not in os.environ and opt.half_guard_every == 32
    opt = parse(tmp_path, ['--in_flight', '4', '--batchSize', '8', '--writers', '3'])
    assert (opt.in_flight, opt.writers, os.environ.get("GPU_MAX_HW_QUEUES")) == (4, 3, "8")
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "6")                    # an explicit choice of the user is kept
    parse(tmp_path, ['--in_flight', '2'])
    assert os.environ["GPU_MAX_HW_QUEUES"] == "6"
    with pytest.raises(ValueError):
        parse(tmp_path, ['--in_flight', '0'])


def test_plan_info_says_which_launch_plan_and_why(monkeypatch):
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    monkeypatch.delenv("CFEN_SERIAL", raising=False)
    net = dec_ipt(NetConfig(24, 4, patch_size=8, load_size=64))
    assert net.plan_info()["lanes_per_forward"] == 2 and "default" in net.plan_info()["why"]
    net.serial_plan = True
    assert net.plan_info()["lanes_per_forward"] == 1 and "caller" in net.plan_info()["why"]
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
    net8 = dec_ipt(NetConfig(24, 4, patch_size=8, load_size=64))
    assert net8.plan_info()["lanes_per_forward"] == 1 and "GPU_MAX_HW_QUEUES" in net8.plan_info()["why"]


def test_writer_processes_encode_from_the_shared_ring(tmp_path):
    """pipeline.start_writer_processes (test.py --writer_procs, round 6): workers forked here read an image from the shared mapping at the offset they are handed and write
    the PNG util.save_image would; batch slots sit in front of the single-image slots; --png_compress_level changes the file's bytes, not its pixels"""
    from cfen_vit_dehazing_amd import pipeline
    n = 16
    w = pipeline.start_writer_processes(2, n, batch=2, labels=1, in_flight=2)
    try:
        assert w["bslots"] >= 4 and w["slot0"] == w["bslots"] * w["batch_bytes"] and w["img_bytes"] == n * n * 3
        rs = np.random.RandomState(3)
        imgs = [rs.randint(0, 256, size=(n, n, 3)).astype(np.uint8) for _ in range(3)]
        offs = [1 * w["batch_bytes"] + 1 * w["img_bytes"], w["slot0"], w["slot0"] + 3 * w["img_bytes"]]      # image 1 of batch slot 1, single slots 0 and 3
        res = []
        for k, (im, off) in enumerate(zip(imgs, offs)):
            np.frombuffer(w["ring"], dtype=np.uint8, count=im.size, offset=off).reshape(im.shape)[...] = im
            res.append(w["pool"].apply_async(pipeline._save_png_from_ring, (off, im.shape, str(tmp_path / ("w%d.png" % k)), 40 + k)))
        assert [r.get(timeout=60) for r in res] == [40, 41, 42]                                                   # the token comes back (it frees the slot)
        for k, im in enumerate(imgs):
            util.save_image(im, str(tmp_path / ("d%d.png" % k)))
            assert open(tmp_path / ("w%d.png" % k), "rb").read() == open(tmp_path / ("d%d.png" % k), "rb").read()
    finally:
        pipeline.stop_writer_processes()
    assert pipeline._WRITERS is None
    was = util.PNG_COMPRESS_LEVEL
    try:
        opt = TestOptions().parse(['--dataroot', str(tmp_path), '--name', 'x', '--gpu_ids', '-1', '--png_compress_level', '1', '--checkpoints_dir', str(tmp_path / 'ckpt')])
        assert opt.png_compress_level == 1 and util.PNG_COMPRESS_LEVEL == 1
        smooth = np.tile(np.arange(64, dtype=np.uint8)[None, :, None], (64, 1, 3))
        util.save_image(smooth, str(tmp_path / "l1.png"))
        util.PNG_COMPRESS_LEVEL = None
        util.save_image(smooth, str(tmp_path / "l6.png"))
        assert np.array_equal(np.asarray(Image.open(tmp_path / "l1.png")), np.asarray(Image.open(tmp_path / "l6.png")))
        with pytest.raises(ValueError):
            TestOptions().parse(['--dataroot', str(tmp_path), '--name', 'x', '--gpu_ids', '-1', '--png_compress_level', '12'])
    finally:
        util.PNG_COMPRESS_LEVEL = was
