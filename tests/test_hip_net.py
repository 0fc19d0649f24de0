"""GPU parity of the whole generator forward (C ABI cfen_net_forward) against
  (1) golden vectors produced by the imported reference (tests/golden/net_*.npz), and
  (2) the CPU oracle run here on the same seeded weights / inputs.

Tolerances (BASELINE.json north_star): fp32 path <= 1e-3 max-abs on the outputs (we hold 2e-4 on every
stage); fp16 path: PSNR / SSIM of the outputs against a reference target move by <= 0.01 dB / 1e-4
relative to the fp32 reference outputs."""
import numpy as np
import pytest
import torch

import cfen_oracle
from cfen_vit_dehazing_amd.config import NetConfig
from cfen_vit_dehazing_amd.hipnet import dec_ipt
from cfen_vit_dehazing_amd.manifest import generate_state_dict, synthetic_input
from cfen_vit_dehazing_amd import ops
from helpers import load_net_fixture, check_outputs, check_stages, weight_mode, sample_idx

pytestmark = pytest.mark.gpu

# max-abs bar of the fp16 path on outputs in (-1, 1) with the seeded weights: measured 8.7e-4 .. 9.9e-4 at 512 x 512 in every record of rounds 1-4; the bar sits
# 5x above that (it was 3e-2: a kernel regression costing a factor 10 in accuracy passed every gate -- VERDICT r04)
FP16_BAR = 5e-3
TAIL_FUSED_DEFAULT = 2   # "net.tail_fused" of the shipped library (restored by tests that change it)


_SD_CACHE = {}


def cached_state_dict(cfg, seed=0, mode="trained"):
    """generate_state_dict is deterministic and takes 4 s at 512 x 512 (it also draws the 0.2 G never-read parameters of a real checkpoint): the last few are kept
    (tests only read them -- load_state_dict copies)"""
    key = (repr(cfg), seed, mode)
    if key not in _SD_CACHE:
        if len(_SD_CACHE) >= 3:
            _SD_CACHE.pop(next(iter(_SD_CACHE)))
        _SD_CACHE[key] = generate_state_dict(cfg, seed=seed, mode=mode)
    return _SD_CACHE[key]


def make_net(cfg, dtype, seed=0, mode="trained", sd=None):
    net = dec_ipt(cfg, compute_dtype=dtype)
    net.load_state_dict(sd if sd is not None else cached_state_dict(cfg, seed=seed, mode=mode), strict=True)
    return net.to("cuda:0")


def gpu_stages(net, z):
    st = {}
    xf = net.stage("ds_conv_e01")
    for n in [str(s) for s in z["stage_names"]]:
        if n.startswith("tail_"):
            continue
        t = net.stage(n)
        if n.startswith("lgcat_conv_d01"):
            t = t - xf                      # the plan folds `+ xf` (v3:696,852,1008) into this stage's epilogue
        st[n] = t
    return st


@pytest.mark.parametrize("name", ["tiny_nf24_hdr4", "tiny_nf24_hdr2", "small_nf24_hdr4"])
def test_fp32_matches_reference_vectors_all_stages(name):
    cfg, batch, z = load_net_fixture(name)
    net = make_net(cfg, "fp32")
    x = synthetic_input(batch, cfg).to("cuda:0")
    outs = net(x)
    st = gpu_stages(net, z)
    for nm, o in zip(("tail_R", "tail_S", "tail_D"), outs):
        st[nm] = o
    worst = check_stages(z, st, 2e-4, rel_sum=2e-4)
    wo = check_outputs(z, outs, 1e-4)
    print("%s fp32: worst stage sample diff %.2e, outputs %.2e" % (name, worst, wo))


def test_fp32_full512_matches_reference_vectors():
    cfg, batch, z = load_net_fixture("full512_nf24_hdr4")
    net = make_net(cfg, "fp32")
    outs = net(synthetic_input(batch, cfg).to("cuda:0"))
    st = gpu_stages(net, z)
    for nm, o in zip(("tail_R", "tail_S", "tail_D"), outs):
        st[nm] = o
    check_stages(z, st, 3e-4, rel_sum=2e-4)
    wo = check_outputs(z, outs, 1e-4)
    assert wo <= 1e-3                               # the north_star bar
    print("full512 fp32 outputs max-abs vs reference %.2e" % wo)


def test_fp32_hdr2_and_1024_configs():
    for name in ("full512_nf24_hdr2", "full1024_nf24_hdr4"):
        cfg, batch, z = load_net_fixture(name)
        net = make_net(cfg, "fp32")
        outs = net(synthetic_input(batch, cfg).to("cuda:0"))
        check_outputs(z, outs, 2e-4)
        del net
        torch.cuda.empty_cache()


def _psnr_ssim_delta(ref_outs, got_outs, target):
    res = []
    for r, g in zip(ref_outs, got_outs):
        t = target[:, :r.shape[1]]
        res.append((abs(cfen_oracle.psnr(r, t) - cfen_oracle.psnr(g, t)), abs(cfen_oracle.ssim(r, t) - cfen_oracle.ssim(g, t)),
                    cfen_oracle.psnr(r, g), float((r - g).abs().max())))
    return res


@pytest.mark.parametrize("name", ["tiny_nf24_hdr4", "small_nf24_hdr4"])
def test_fp16_psnr_ssim_against_oracle(name):
    cfg, batch, z = load_net_fixture(name)
    sd = generate_state_dict(cfg, seed=0, with_dead=False)
    x = synthetic_input(batch, cfg)
    with torch.no_grad():
        ref = cfen_oracle.forward(sd, x, cfg.num_heads, cfg.patch_size)
    net = make_net(cfg, "fp16")
    got = [o.cpu() for o in net(x.to("cuda:0"))]
    # "ground truth" stand-in: the (clean) input itself; what matters is that both outputs score the same
    for dp, ds, p, mx in _psnr_ssim_delta(ref, got, x):
        print("%s fp16: dPSNR %.4f dB, dSSIM %.2e, PSNR(fp16 vs ref) %.1f dB, max-abs %.2e" % (name, dp, ds, p, mx))
        assert dp <= 0.01 and ds <= 1e-4
        assert p >= 50.0 and mx <= FP16_BAR


def test_fp16_full512_against_reference_vectors():
    cfg, batch, z = load_net_fixture("full512_nf24_hdr4")
    net = make_net(cfg, "fp16")
    outs = net(synthetic_input(batch, cfg).to("cuda:0"))
    worst = check_outputs(z, outs, FP16_BAR)
    print("full512 fp16 outputs max-abs vs reference %.2e" % worst)
    for nm, o in zip(("xr", "xs", "xd"), outs):
        stat = z["stat/" + nm]
        assert abs(float(o.mean()) - stat[0]) < 2e-3 and abs(float(o.std()) - stat[1]) < 2e-3


def test_batch_invariance_and_determinism():
    cfg = NetConfig(24, 4, patch_size=8, load_size=64)
    net = make_net(cfg, "fp32")
    x = synthetic_input(3, cfg).to("cuda:0")
    a = [o.clone() for o in net(x)]
    b = [o.clone() for o in net(x)]
    one = net(x[2:3])
    for u, v, w in zip(a, b, one):
        assert torch.equal(u, v)                                  # bitwise reproducible
        assert float((u[2:3] - w).abs().max()) <= 1e-5            # sharding the batch does not change results


def test_wrong_input_size_raises():
    cfg = NetConfig(24, 4, patch_size=8, load_size=64)
    net = make_net(cfg, "fp32")
    with pytest.raises(RuntimeError):
        net(torch.zeros(1, 3, 64, 64, device="cuda:0"))
    with pytest.raises(Exception):
        net(torch.zeros(1, 3, 128, 128))                          # CPU tensor: no fallback


@pytest.mark.parametrize("load_size,nimg,precision,levels", [(64, 3, "single", 1), (256, 1, "half", 6)])
def test_cli_end_to_end_writes_reference_named_pngs(tmp_path, load_size, nimg, precision, levels):
    """python test.py with the reference's README flags on a synthetic checkpoint: PNGs land where the
    reference puts them and equal tensor2im(oracle output) up to one grey level (fp16: 6 levels = 0.047 of the [-1, 1] range, the fp16
    output bar of test_fp16_full512_against_reference_vectors, and 99 % of the pixels within one level).  load_size 256 = the benchmark's
    512 x 512 images through the CLI, not only through the module."""
    import os
    import subprocess
    import sys
    from PIL import Image
    from cfen_vit_dehazing_amd.util import util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = NetConfig(24, 4, patch_size=load_size // 8, load_size=load_size)   # the reference nests its crops: loadSize = 8 * patch_size
    sd = generate_state_dict(cfg, seed=0)
    name = "iid_hlgvit_crs_gd4_cfs_v3_synthetic"
    os.makedirs(tmp_path / "ckpt" / name)
    torch.save(sd, tmp_path / "ckpt" / name / "32_net_G.pth")
    os.makedirs(tmp_path / "data" / "hazy")
    rs = np.random.RandomState(0)
    imgs = []
    for i in range(nimg):
        a = rs.randint(0, 256, (2 * load_size, 2 * load_size, 3), dtype=np.uint8)
        Image.fromarray(a).save(tmp_path / "data" / "hazy" / ("syn_%04d.png" % (i + 1)))
        imgs.append(a)
    cmd = [sys.executable, os.path.join(root, "test.py"), "--dataroot", str(tmp_path / "data"), "--name", name, "--n_feats", "24",
           "--hidden_dim_ratio", "4", "--sb", "--out_all", "--which_epoch", "32", "--loadSize", str(load_size), "--patch_size", str(load_size // 8),
           "--checkpoints_dir", str(tmp_path / "ckpt"), "--results_dir", str(tmp_path / "res"), "--precision", precision]
    subprocess.check_call(cmd, cwd=str(tmp_path))
    out_dir = tmp_path / "res" / name / "test_32" / "images"
    assert sorted(os.listdir(out_dir)) == ["syn_%04d_fake_A.png" % (i + 1) for i in range(nimg)]
    live = {k: v for k, v in sd.items()}
    for i, a in enumerate(imgs):
        x = (torch.from_numpy(a).permute(2, 0, 1).float() / 255 - 0.5) / 0.5
        with torch.no_grad():
            xd = cfen_oracle.forward(live, x[None], cfg.num_heads, cfg.patch_size)[2]
        want = util.tensor2im(xd[0]).astype(np.int32)
        got = np.asarray(Image.open(out_dir / ("syn_%04d_fake_A.png" % (i + 1)))).astype(np.int32)
        assert got.shape == want.shape and np.abs(got - want).max() <= levels
        assert (got != want).mean() < 0.01 if levels == 1 else (np.abs(got - want) > 1).mean() < 0.01


def test_native_graph_replay_equals_eager():
    """cfen_net_graph_capture builds the multi-lane plan as explicit hipGraph nodes; replay must be bitwise eager."""
    cfg = NetConfig(24, 4, patch_size=8, load_size=64)
    net = make_net(cfg, "fp16")
    x = synthetic_input(2, cfg).to("cuda:0")
    eager = [o.clone() for o in net(x)]
    gid, outs = net.capture(x)
    for o in outs:
        o.zero_()
    net.replay(gid)
    torch.cuda.synchronize()
    for a, b in zip(eager, outs):
        assert torch.equal(a, b)
    x.copy_(synthetic_input(2, cfg, seed0=5).to("cuda:0"))      # same buffer, new content
    net.replay(gid)
    want = [o.clone() for o in outs]
    again = net(x)
    for a, b in zip(want, again):
        assert torch.equal(a, b)


def test_profile_entries_label_every_launch():
    """cfen_net_profile_entry: one labelled record per launch; the decoders' layers go out as grouped launches (x3)"""
    cfg = NetConfig(24, 4, patch_size=8, load_size=64)
    net = make_net(cfg, "fp16")
    x = synthetic_input(2, cfg).to("cuda:0")
    prof = net.profile(x)
    launches = prof["launches"]
    assert sum(prof[c][2] for c in net.KERNEL_CLASSES) == len(launches) > 50
    labels = [l[0] for l in launches]
    assert labels[0].startswith("head.0.0") and any("(x3)" in l for l in labels) and any("tail_R.conv7" in l for l in labels)
    assert all(l[3] > 0 for l in launches)
    assert abs(sum(l[2] for l in launches) - sum(prof[c][1] for c in net.KERNEL_CLASSES)) < 1e-3 * sum(prof[c][1] for c in net.KERNEL_CLASSES)
    # round 4: every record also names the DEVICE KERNEL(S) that ran it (cfen_net_profile_entry_kernel) and, for the token GEMMs, algorithmic bytes
    kernels = [l[4] for l in launches]
    assert all(k.startswith("k_") for k in kernels), [k for k in kernels if not k.startswith("k_")]
    assert kernels[0].startswith("k_head5") and any(k.startswith("k_gemm_dma") for k in kernels) and any("k_tail_fused" in k or "k_conv7_tz" in k for k in kernels)
    assert all(l[5] >= 0 for l in launches) and any(l[5] > 0 for l in launches if l[4].startswith("k_gemm"))


def test_split_k_and_grouping_knobs_do_not_change_results():
    """the split-K path (GViT-3 ffn2 / head2 at <= 128 tokens) and its plain counterpart agree to fp16 rounding, both deterministic"""
    from cfen_vit_dehazing_amd import ops
    cfg = NetConfig(24, 4, patch_size=32, load_size=256)
    x = synthetic_input(1, cfg).to("cuda:0")
    outs = {}
    try:
        for k in (1, 0):
            ops.tune("gemm.splitk", k)
            net = make_net(cfg, "fp16")
            a = [o.clone() for o in net(x)]
            b = [o.clone() for o in net(x)]
            assert all(torch.equal(p, q) for p, q in zip(a, b))
            outs[k] = a
    finally:
        ops.tune("gemm.splitk", 0)          # the shipped default (round 4)
    for p, q in zip(outs[0], outs[1]):
        assert float((p - q).abs().max()) < 5e-3


# ---- the benchmarked configurations themselves (BASELINE.json configs 2, 4, 5), fp16, replayed from the hipGraph ----------------

def _psnr_ssim_on_device(a, b):
    """cfen_oracle.psnr / ssim (the metric definitions of the fixtures: peak 2, 11 x 11 Gaussian window, sigma 1.5, float64) evaluated on the GPU for whole
    512 x 512 / 1024 x 1024 planes of whole batches -- on the host these comparisons were a quarter of the GPU suite's wall time"""
    import torch.nn.functional as F
    a, b = a.double(), b.double()
    mse = torch.mean((a - b) ** 2)
    psnr = float(10 * torch.log10(4.0 / mse)) if float(mse) > 0 else float("inf")
    a01, b01 = (a + 1) / 2, (b + 1) / 2
    g = torch.exp(-((torch.arange(11, dtype=torch.float64, device=a.device) - 5) ** 2) / (2 * 1.5 * 1.5))
    g = g / g.sum()
    C = a.shape[1]
    w = (g[:, None] * g[None, :]).expand(C, 1, 11, 11).contiguous()
    mu1, mu2 = F.conv2d(a01, w, padding=5, groups=C), F.conv2d(b01, w, padding=5, groups=C)
    s11 = F.conv2d(a01 * a01, w, padding=5, groups=C) - mu1 * mu1
    s22 = F.conv2d(b01 * b01, w, padding=5, groups=C) - mu2 * mu2
    s12 = F.conv2d(a01 * b01, w, padding=5, groups=C) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))
    return psnr, float(m.mean())


def test_device_side_psnr_ssim_equal_the_oracle_definitions():
    g = torch.Generator().manual_seed(3)
    a = torch.rand(1, 3, 96, 80, generator=g) * 2 - 1
    b = (a + 0.05 * torch.randn(a.shape, generator=g)).clamp(-1, 1)
    p, s = _psnr_ssim_on_device(a.to("cuda:0"), b.to("cuda:0"))
    assert abs(p - cfen_oracle.psnr(a, b)) < 1e-9 and abs(s - cfen_oracle.ssim(a, b)) < 1e-12


def _crop(t):
    n = t.shape[-1]
    c0 = n // 2 - 32
    return t[:, :, c0:c0 + 64, c0:c0 + 64]


def _fp16_vs_fixture(z, outs, x, max_abs_bar):
    """image 0 of `outs` against the reference vectors: max-abs on crop + strided samples, and PSNR / SSIM of the 64x64 centre crop
    against a common target (the input crop) must move by <= 0.01 dB / 1e-4 relative to the reference's own outputs"""
    first = [o[0:1].float().cpu() for o in outs]
    worst = check_outputs(z, first, max_abs_bar)
    tgt = _crop(x[0:1].float().cpu())
    for nm, o in zip(("xr", "xs", "xd"), first):
        ref = torch.from_numpy(z["crop/" + nm])
        got = _crop(o)
        t = tgt[:, :ref.shape[1]]
        dp = abs(cfen_oracle.psnr(ref, t) - cfen_oracle.psnr(got, t))
        ds = abs(cfen_oracle.ssim(ref, t) - cfen_oracle.ssim(got, t))
        assert dp <= 0.01 and ds <= 1e-4, "%s: dPSNR %.4f dB dSSIM %.2e" % (nm, dp, ds)
    return worst


@pytest.mark.parametrize("name,batch", [("full512_nf24_hdr4", 8), ("full512_nf24_hdr2", 16), ("full1024_nf24_hdr4", 4)])
def test_fp16_benchmark_configs_graph_replay_vs_reference_vectors(name, batch):
    """exactly what bench.py times: batch B, fp16, hipGraph replay.  Image 0 is the fixture's input (seed 0); the other images
    must equal their own batch-1 eager forward up to fp16 rounding (the few-token GViT GEMMs pick other kernels at other batch sizes)"""
    cfg, _, z = load_net_fixture(name)
    net = make_net(cfg, "fp16")
    x = synthetic_input(batch, cfg).to("cuda:0")
    eager = [o.clone() for o in net(x)]
    gid, outs = net.capture(x)
    for o in outs:
        o.zero_()
    net.replay(gid)
    net.replay(gid)
    torch.cuda.synchronize()
    for a, b in zip(eager, outs):
        assert torch.equal(a, b)                                   # replay is bitwise the eager plan
    worst = _fp16_vs_fixture(z, outs, x, FP16_BAR)
    print("%s B=%d fp16 graph: image 0 max-abs vs reference vectors %.2e" % (name, batch, worst))
    if name == "full512_nf24_hdr4":
        # EVERY image of the benchmarked batch against the reference's own batch-8 forward (fixture full512b8: seeds 0 .. 7): max-abs on the crop and
        # strided samples of all eight, whole-image PSNR / SSIM of all 24 output planes against the input within 0.01 dB / 1e-4 of the reference's,
        # and the 256 x 256 centre crop of image 0
        _, b8, z8 = load_net_fixture("full512b8_nf24_hdr4")
        assert b8 == batch == 8
        w8 = check_outputs(z8, outs, FP16_BAR)
        worst_dp = worst_ds = 0.0
        for nm, o in zip(("xr", "xs", "xd"), outs):
            oc = o.float().cpu()
            for b in range(batch):
                pp, ss = _psnr_ssim_on_device(o[b:b + 1], x[b:b + 1, :o.shape[1]])
                worst_dp = max(worst_dp, abs(pp - float(z8["full_psnr/" + nm][b])))
                worst_ds = max(worst_ds, abs(ss - float(z8["full_ssim/" + nm][b])))
            n = oc.shape[-1]
            d = float((oc[0:1, :, n // 2 - 128:n // 2 + 128, n // 2 - 128:n // 2 + 128] - torch.from_numpy(z8["crop256/" + nm])).abs().max())
            assert d <= FP16_BAR, "%s: 256 x 256 crop of image 0 differs by %.3e" % (nm, d)
        print("   images 0..7 vs the reference's batch-8 vectors: max-abs %.2e; whole-image dPSNR %.4f dB, dSSIM %.2e" % (w8, worst_dp, worst_ds))
        assert worst_dp <= 0.01 and worst_ds <= 1e-4
    else:
        # configs 5 and 4: every image of the batch against the reference's own forward of that batch (fixtures full512b16_nf24_hdr2 / full1024b4_nf24_hdr4:
        # strided samples of all images + whole-image PSNR / SSIM against the input)
        _, bb, zb = load_net_fixture(name.replace("full512", "full512b16").replace("full1024", "full1024b4"))
        assert bb == batch
        wb = check_outputs(zb, outs, FP16_BAR)
        worst_dp = worst_ds = 0.0
        for nm, o in zip(("xr", "xs", "xd"), outs):
            for b in range(batch):
                pp, ss = _psnr_ssim_on_device(o[b:b + 1], x[b:b + 1, :o.shape[1]])
                worst_dp = max(worst_dp, abs(pp - float(zb["full_psnr/" + nm][b])))
                worst_ds = max(worst_ds, abs(ss - float(zb["full_ssim/" + nm][b])))
        print("   all %d images vs the reference's batch vectors: max-abs %.2e; whole-image dPSNR %.4f dB, dSSIM %.2e" % (batch, wb, worst_dp, worst_ds))
        assert worst_dp <= 0.01 and worst_ds <= 1e-4
    for i in (batch - 1,):           # (every image is checked against the reference's own batch forward above; one batch-1 forward keeps the batch-invariance check)
        one = net(x[i:i + 1].clone())
        for a, b in zip(outs, one):
            d = float((a[i:i + 1] - b).abs().max())
            assert d <= FP16_BAR, "image %d differs from its batch-1 forward by %.3e" % (i, d)
    del net
    torch.cuda.empty_cache()


def test_fp32_batch2_vs_reference_vectors_of_both_images():
    """exact-fp32 path, B = 2, 512x512: outputs AND all 58 stages of both images against the reference's batch-2 forward"""
    cfg, batch, z = load_net_fixture("full512b2_nf24_hdr4")
    net = make_net(cfg, "fp32")
    outs = net(synthetic_input(batch, cfg).to("cuda:0"))
    st = gpu_stages(net, z)
    for nm, o in zip(("tail_R", "tail_S", "tail_D"), outs):
        st[nm] = o
    check_stages(z, st, 3e-4, rel_sum=2e-4)
    worst = check_outputs(z, outs, 1e-4)
    print("fp32 B=2 max-abs vs reference vectors %.2e" % worst)
    del net
    torch.cuda.empty_cache()


def test_half_precision_guard_keeps_safe_weights_and_falls_back_on_unsafe_ones(tmp_path):
    """--precision half through the model wrapper: the first batch also runs in fp32; seeded weights stay on fp16, a checkpoint whose
    activations leave the fp16 range (an FFN scaled by 3e4) is caught and the model continues in single precision with fp32-exact outputs"""
    from cfen_vit_dehazing_amd.models import create_model
    from cfen_vit_dehazing_amd.options.test_options import TestOptions
    cfg = NetConfig(24, 4, patch_size=8, load_size=64)
    sd = generate_state_dict(cfg, seed=0)
    x = synthetic_input(2, cfg)
    for tag, scale in (("safe", 1.0), ("unsafe", 3e4)):
        ck = tmp_path / tag / "half_guard"
        ck.mkdir(parents=True)
        sd2 = dict(sd)
        sd2["localvit_encoder_01.encoder.layers.0.linear1.weight"] = sd["localvit_encoder_01.encoder.layers.0.linear1.weight"] * scale
        torch.save(sd2, ck / "latest_net_G.pth")
        opt = TestOptions().parse(['--dataroot', str(tmp_path), '--checkpoints_dir', str(tmp_path / tag), '--name', 'half_guard', '--n_feats', '24',
                                   '--hidden_dim_ratio', '4', '--patch_size', '8', '--loadSize', '64', '--sb', '--precision', 'half'])
        model = create_model(opt)
        model.setup(opt)
        model.set_input({'B': x, 'B_paths': ['a.png', 'b.png']})
        model.test(opt)
        fa = model.get_current_visuals()['fake_A'].clone()
        assert torch.isfinite(fa).all()
        if tag == "safe":
            assert model.half_guard_max_abs <= model.HALF_GUARD_BAR and model.netG.compute_dtype == torch.float16
        else:
            assert not model.half_guard_max_abs <= model.HALF_GUARD_BAR and model.netG.compute_dtype == torch.float32
            ref = make_net(cfg, "fp32")
            ref.load_state_dict(sd2)
            ref.to("cuda:0")
            assert torch.equal(fa, ref(x.to("cuda:0"))[2])
        model.test(opt)            # the second batch runs on whatever the guard settled on, without the double forward
        assert torch.isfinite(model.get_current_visuals()['fake_A']).all()


@pytest.mark.parametrize("size", [(8, 64), (32, 256)])
def test_fused_head_equals_the_three_convolutions_bitwise(size):
    """k_head_fused (conv5x5 + ResBlock in one launch, both intermediates as fp16 tiles in LDS) against the three k_conv_tile launches it
    replaces: the `head` stage and the outputs, bit for bit (same MFMA order, same fp16 rounding of the intermediates), at a size with many
    border tiles and at 512x512"""
    from cfen_vit_dehazing_amd import ops
    ps, ls = size
    cfg = NetConfig(24, 4, patch_size=ps, load_size=ls)
    x = synthetic_input(2, cfg).to("cuda:0")
    res = {}
    try:
        ops.tune("net.head5", 0)                 # (round 4's k_head5 sums its products in another order: the three round-3 launches are the twin here)
        ops.tune("net.resblock_fused", 0)
        for k in (1, 0):
            ops.tune("net.head_fused", k)
            net = make_net(cfg, "fp16")
            outs = [o.clone() for o in net(x)]
            res[k] = (net.stage("head").clone(), outs)
            del net
    finally:
        ops.tune("net.head_fused", 0)
        ops.tune("net.head5", 1)
        ops.tune("net.resblock_fused", 1)
    assert torch.equal(res[1][0], res[0][0])
    for a, b in zip(res[1][1], res[0][1]):
        assert torch.equal(a, b)
    torch.cuda.empty_cache()


@pytest.mark.parametrize("size", [(8, 64), (32, 256)])
def test_head_from_input_and_fused_resblock_against_the_unfused_launches(size):
    """round 4: k_head5 (input layout pass + conv5x5 in one launch, 8-byte pixels) and k_resblock_fused (hidden map in LDS) against the launches they
    replace (k_nchw_to_nhwc + three k_conv_tile): the `head` stage and the outputs; every combination of the two knobs, fp32 NCHW and uint8 HWC input"""
    from cfen_vit_dehazing_amd import ops
    ps, ls = size
    cfg = NetConfig(24, 4, patch_size=ps, load_size=ls)
    x = synthetic_input(2, cfg).to("cuda:0")
    n = cfg.image_size
    img = torch.randint(0, 256, (2, n, n, 3), generator=torch.Generator().manual_seed(5), dtype=torch.uint8).to("cuda:0")
    res = {}
    try:
        for h5, rb in ((0, 0), (1, 0), (0, 1), (1, 1)):
            ops.tune("net.head5", h5)
            ops.tune("net.resblock_fused", rb)
            net = make_net(cfg, "fp16")
            outs = [o.clone() for o in net(x)]
            head = net.stage("head").clone()
            outs8 = [o.clone() for o in net(img)]
            res[(h5, rb)] = (head, outs, outs8)
            del net
    finally:
        ops.tune("net.head5", 1)
        ops.tune("net.resblock_fused", 1)
    # the fused ResBlock is bitwise the two launches (same accumulation order, same fp16 rounding of the hidden map) ...
    for h5 in (0, 1):
        a, b = res[(h5, 1)], res[(h5, 0)]
        assert torch.equal(a[0], b[0]), h5
        for u, v in zip(a[1] + a[2], b[1] + b[2]):
            assert torch.equal(u, v), h5
    # ... k_head5 has the same products per output as k_nchw_to_nhwc + k_conv_tile<16, 5> but sums them in chunks of 8 taps instead of 4 + 1:
    # fp32 reassociation, at most an fp16 ulp on the stored map
    a, b = res[(1, 1)], res[(0, 0)]
    assert float((a[0] - b[0]).abs().max()) <= 2e-3 * max(1.0, float(b[0].abs().max()))
    for u, v in zip(a[1] + a[2], b[1] + b[2]):
        assert float((u - v).abs().max()) <= 5e-3
    torch.cuda.empty_cache()


@pytest.mark.parametrize("size,batch", [((8, 64), 2), ((32, 256), 2), ((64, 512), 1)])
def test_fused_tail_equals_the_separate_launches_bitwise(size, batch):
    """round 4: k_up_conv3_fused (us_conv_d01* ConvTranspose + ActNorm + ReLU and the tail's 3x3 in one grouped launch, the 12-channel map between
    them in LDS with its halo recomputed) against k_convT_tile + k_conv_tile: the three outputs bit for bit, and -- with "net.keep_stages" -- the
    us_conv_d01* stage maps bit for bit; without that knob the stage is refused instead of returning a stale buffer.
    round 5: k_tail_fused ("net.tail_fused" = 2: the reflect-pad 7x7 + tanh too, a workgroup walking down a 64-column strip) -- float and uint8
    outputs bit for bit those of the three launches, for every number of vertical segments a strip can be cut into"""
    from cfen_vit_dehazing_amd import ops
    from cfen_vit_dehazing_amd._lib import CfenError
    ps, ls = size
    cfg = NetConfig(24, 4, patch_size=ps, load_size=ls)
    x = synthetic_input(batch, cfg).to("cuda:0")
    res = {}
    try:
        configs = ((0, 0, 4), (1, 1, 4), (1, 0, 4), (2, 0, 4), (2, 0, 1), (2, 0, 2), (2, 0, 8))
        if ls >= 256:       # 512 x 512 and 1024 x 1024: the separate launches, the round-4 pair with its stage maps, the shipped shape and one other segmentation
            configs = ((0, 0, 4), (1, 1, 4), (2, 0, 1), (2, 0, 4)) if ls == 256 else ((0, 0, 4), (1, 1, 4), (2, 0, 1))
        for fused, keep, seg in configs:
            ops.tune("net.tail_fused", fused)
            ops.tune("net.keep_stages", keep)
            ops.tune("tail.segments", seg)
            net = make_net(cfg, "fp16")
            outs = [o.clone() for o in net(x)]
            if fused and not keep:
                with pytest.raises(CfenError):
                    net.stage("us_conv_d01r")
                st = None
            else:
                st = [net.stage("us_conv_d01" + t).clone() for t in "rsd"]
            gid, gouts = net.capture(x)
            net.replay(gid)
            torch.cuda.synchronize()
            for a, b in zip(gouts, outs):
                assert torch.equal(a, b)
            net.output_u8 = True
            u8 = [o.clone() for o in net(x)]
            net.output_u8 = False
            res[(fused, keep, seg)] = (outs, st, u8)
            del net
    finally:
        ops.tune("net.tail_fused", TAIL_FUSED_DEFAULT)
        ops.tune("net.keep_stages", 0)
        ops.tune("tail.segments", 1)
    ref = res[(0, 0, 4)]
    for key, got in res.items():
        for a, b in zip(got[0], ref[0]):
            assert torch.equal(a, b), key
        for a, b in zip(got[2], ref[2]):
            assert torch.equal(a, b), key
    for a, b in zip(res[(1, 1, 4)][1], ref[1]):
        assert torch.equal(a, b)
    torch.cuda.empty_cache()


@pytest.mark.parametrize("size,batch", [(64, 2), (256, 8)])
def test_gvit_upsampling_inside_the_fuse_conv_equals_the_upsample_launch(size, batch):
    """"net.up_fused" (default 0: measured slower): GViT's upsam(upsam(x)) (v3:1323) runs inside the level's 1x1 fuse conv (k_conv UP: the workgroup's pixels are
    interpolated from the low-resolution map in LDS with k_upsample4's arithmetic) instead of k_upsample4 writing a full-resolution copy -- the
    three outputs and every lgcat stage against the plan with the launch (fp16-rounding close); graph replay = eager, bitwise; the GViT stages, which are then
    never stored, are refused by net.stage; the profile has no upsample4 entry left"""
    from cfen_vit_dehazing_amd import ops
    from cfen_vit_dehazing_amd._lib import CfenError
    cfg = NetConfig(24, 4, patch_size=size // 8, load_size=size)
    x = synthetic_input(batch, cfg).to("cuda:0")
    names = ["lgcat_conv_e01", "lgcat_conv_e02", "lgcat_conv_e03", "lgcat_conv_d03r", "lgcat_conv_d02s", "lgcat_conv_d01d"]
    res = {}
    try:
        for fused in (0, 1):
            ops.tune("net.up_fused", fused)
            net = make_net(cfg, "fp16")
            net(x)                                   # first forward initialises nothing here (trained ActNorm), second is the measured plan
            outs = [o.clone() for o in net(x)]
            st = {k: net.stage(k).clone() for k in names}
            if fused:
                with pytest.raises(CfenError):
                    net.stage("globalvit_encoder_01")
                ups = [l[0] for l in net.profile(x)["launches"] if "upsample4" in l[0]]
                # (the 16 x 16 level-3 maps of the 128 x 128 case do not tile into the kernel's pixel runs: that level keeps its launch)
                assert not ups if size == 256 else all("_03" in u for u in ups) and len(ups) == 2
            else:
                net.stage("globalvit_encoder_01")
            gid, gouts = net.capture(x)
            net.replay(gid)
            torch.cuda.synchronize()
            for a, b in zip(gouts, outs):
                assert torch.equal(a, b)
            res[fused] = (outs, st)
            del net
    finally:
        ops.tune("net.up_fused", 0)
    # the same interpolation in two kernels: fp32 sums the compiler contracts differently, then one rounding to fp16 -- last-bit differences of the GViT half of
    # the fuse conv's input (forcing separate multiplies and adds in both made the plans bitwise equal and k_upsample4 20 % slower: not worth it for a variant)
    for k in names:
        a, b = res[0][1][k], res[1][1][k]
        assert float((a - b).abs().max()) <= 4e-3 * max(1.0, float(a.abs().max())), k
    for a, b in zip(res[0][0], res[1][0]):
        assert float((a - b).abs().max()) <= 2e-3
    torch.cuda.empty_cache()


def test_gvit_persistent_chain_plan_agrees_with_the_launch_per_gemm_plan():
    """csrc/k_gvit.hip inside the net (CFEN_GVIT_CHAIN=1: fragment-stream GViT weights, two persistent launches per block instead of eight GEMM
    launches): every GViT stage and the outputs against the default plan at B = 8, 512x512, fp16 -- same math, different summation order
    (128-row units, split-K slices), so fp16-rounding close, not bitwise; the teams' error words stay zero; graph replay = eager, bitwise"""
    import os
    from cfen_vit_dehazing_amd import ops
    cfg = NetConfig(24, 4, patch_size=32, load_size=256)
    x = synthetic_input(8, cfg).to("cuda:0")
    stages = ["globalvit_encoder_01", "globalvit_encoder_02", "globalvit_encoder_03", "globalvit_decoder_03r", "globalvit_decoder_02s", "globalvit_decoder_01d"]
    ops.tune("net.keep_stages", 1)          # the GViT maps are stored at full resolution (default: x4 bilinear inside the fuse convs)
    try:
        base = make_net(cfg, "fp16")
        want = [o.clone() for o in base(x)]
        wst = {k: base.stage(k).clone() for k in stages}
        del base
        os.environ["CFEN_GVIT_CHAIN"] = "1"
        try:
            net = make_net(cfg, "fp16")
        finally:
            del os.environ["CFEN_GVIT_CHAIN"]
        assert net.gvit_chain
        got = [o.clone() for o in net(x)]
        assert net.chain_errors() == [0, 0, 0]
        for k in stages:
            d = float((net.stage(k) - wst[k]).abs().max())
            assert d <= 3e-2 * max(1.0, float(wst[k].abs().max())), (k, d)
    finally:
        ops.tune("net.keep_stages", 0)
    for a, b in zip(got, want):
        assert float((a - b).abs().max()) <= 2e-2
    gid, gouts = net.capture(x)
    for _ in range(3):
        net.replay(gid)
    torch.cuda.synchronize()
    for a, b in zip(gouts, got):
        assert torch.equal(a, b)
    assert net.chain_errors() == [0, 0, 0]
    torch.cuda.empty_cache()


@pytest.mark.parametrize("splitk", [0, 1, 2])
def test_three_forwards_in_flight_on_replica_plans_match_single_forwards_bitwise(splitk):
    """what bench.py times by default (round 4): consecutive steps rotate over three launch plans -- one serial chain of launches each, own
    workspace and output slab, shared packed weights -- on three streams, so three forwards are in flight at once.  Every replica gets a DIFFERENT
    input batch here; after 12 overlapping steps each slab must equal, bit for bit, the eager one-at-a-time forward of its own batch on the same plan (no cross-talk
    through a shared scratch buffer, counter or stage map), B = 8, 512x512, fp16.
    splitk = 1: the same with "gemm.splitk" on -- the few-token GViT GEMMs reduce their K slices inside the launch through arrival counters, which the plan
    zeroes at the head of every forward.  As a MEMSET NODE of the recorded plan that zeroing was not ordered against the kernel nodes around it once several
    graphs were in flight (outputs off by 0.17; round 4 hunted this in the seam's memory ordering); it is a kernel node now ("net.zero_memset" = 1 is the old form).
    splitk = 2: the persistent GViT chains (CFEN_GVIT_CHAIN=1: grid barriers + split-K seams inside one launch), which failed the same way for
    the same reason; their error words must stay zero."""
    import os
    from cfen_vit_dehazing_amd import ops
    from cfen_vit_dehazing_amd.parallel import split_slab
    cfg = NetConfig(24, 4, patch_size=32, load_size=256)
    ops.tune("gemm.splitk", int(splitk == 1))
    if splitk == 2:
        os.environ["CFEN_GVIT_CHAIN"] = "1"
    try:
        net = make_net(cfg, "fp16")
        assert bool(net.gvit_chain) == (splitk == 2)
        if splitk == 2:
            # the chains' grid barriers need every team of every forward in flight resident at once: without the cap a second replica is refused
            net.replica = 1
            with pytest.raises(Exception, match="gvit.max_concurrent"):
                net(synthetic_input(8, cfg).to("cuda:0"))
            net.replica = 0
            ops.tune("gvit.max_concurrent", 3)
        _three_in_flight(net, cfg, split_slab)
        if splitk == 2:
            assert net.chain_errors() == [0, 0, 0]
    finally:
        os.environ.pop("CFEN_GVIT_CHAIN", None)
        ops.tune("gemm.splitk", 0)
        ops.tune("gvit.max_concurrent", 1)


def _three_in_flight(net, cfg, split_slab):
    n, B = cfg.image_size, 8
    xs = [synthetic_input(B, cfg, seed0=8 * k).to("cuda:0") for k in range(3)]
    two_lane = [[o.clone() for o in net(x)] for x in xs]             # two-lane plan, one forward at a time
    assert net.plan_info()["lanes_per_forward"] == 2
    net.serial_plan = True
    assert net.plan_info()["lanes_per_forward"] == 1 and "caller" in net.plan_info()["why"]
    want = [[o.clone() for o in net(x)] for x in xs]                 # the serial plan, eagerly, one forward at a time
    # the lane plan changes when kernels run, not which kernels run ("net.gvit_stream" = 2 since round 5): bit for bit the two-lane results
    for a, b in zip(sum(want, []), sum(two_lane, [])):
        assert torch.equal(a, b)
    slabs = [torch.empty(7 * B * n * n, dtype=torch.float32, device="cuda:0") for _ in range(3)]
    gids = []
    for k in range(3):
        net.replica = k
        gids.append(net.capture(xs[k], out=slabs[k])[0])
    net.replica = 0
    for s in slabs:
        s.fill_(float("nan"))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream("cuda:0") for _ in range(3)]
    for i in range(12):
        with torch.cuda.stream(streams[i % 3]):
            net.replay(gids[i % 3])
    torch.cuda.synchronize()
    for k in range(3):
        for a, b in zip(split_slab(slabs[k], B, n), want[k]):
            assert torch.equal(a, b), k
    torch.cuda.empty_cache()


def test_two_lane_plan_equals_serial_plan_bitwise_full_size():
    """GViT beside LViT on a second lane must not change a single bit (512x512, B=2, eager, graph and PROFILED forwards), with GViT level 1 on the
    launch-per-GEMM chain ("net.gvit_stream" = 0) and on the stream kernels (2, the default on every plan since round 5); the two kernel families
    are the same math in another summation order and agree to fp16 rounding"""
    from cfen_vit_dehazing_amd import ops
    cfg = NetConfig(24, 4, patch_size=32, load_size=256)
    x = synthetic_input(2, cfg).to("cuda:0")
    res = {}
    try:
        for mode in (0, 2):
            ops.tune("net.gvit_stream", mode)
            ops.tune("net.keep_stages", 1)
            net = make_net(cfg, "fp16")
            net.serial_plan = False
            two = [o.clone() for o in net(x)]
            gid, gouts = net.capture(x)
            net.replay(gid)
            torch.cuda.synchronize()
            prof = net.profile(x)                                  # the profiled forward replays the same launches serially: same kernels, same bits
            names = [l[4] for l in prof["launches"] if l[0].startswith("globalvit_encoder_01")]
            assert any("k_mlp3" in k for k in names) == (mode == 2), names
            net.serial_plan = True
            one = [o.clone() for o in net(x)]
            st = net.stage("globalvit_encoder_01").clone()
            for a, b, c in zip(two, one, gouts):
                assert torch.equal(a, b) and torch.equal(a, c)
            res[mode] = (two, st)
            del net
    finally:
        ops.tune("net.gvit_stream", 2)
        ops.tune("net.keep_stages", 0)
    st0, st2 = res[0][1], res[2][1]
    assert float((st0 - st2).abs().max()) <= 3e-2 * max(1.0, float(st0.abs().max()))
    for a, b in zip(res[0][0], res[2][0]):
        assert float((a - b).abs().max()) <= FP16_BAR


def test_repeated_forwards_are_bit_reproducible_at_benchmark_size():
    """tools/stress_determinism.py as a test: B=8 512x512 fp16, eager and graph alternating, with unrelated GEMM noise on a third stream"""
    cfg = NetConfig(24, 4, patch_size=32, load_size=256)
    net = make_net(cfg, "fp16")
    x = synthetic_input(8, cfg).to("cuda:0")
    ref = [o.clone() for o in net(x)]
    gid, gout = net.capture(x)
    big = torch.randn(4096, 4096, device="cuda:0").half()
    side = torch.cuda.Stream()
    bad = []
    for it in range(24):
        if it % 3 == 2:
            with torch.cuda.stream(side):
                for _ in range(4):
                    torch.mm(big, big)
        if it % 2:
            for o in gout:
                o.zero_()
            net.replay(gid)
            outs = gout
        else:
            outs = net(x)
        torch.cuda.synchronize()
        if not all(torch.equal(a, b) for a, b in zip(ref, outs)):
            bad.append(it)
    assert not bad, "runs %s differ bitwise from the first" % bad


# ---- weights as the reference's define_G leaves them + ActNorm2d first-call initialisation on the device (A9) -----------------

@pytest.mark.parametrize("name", ["refinit_tiny_nf24_hdr4", "refinit_full512_nf24_hdr4"])
def test_reference_init_weights_and_device_actnorm_init_fp32(name):
    """define_G + forward without a checkpoint: the first forward fills the 24 ActNorm2d layers from its batch
    (models/actnorm.py:25-37) on the device; parameters and outputs equal those of the reference's first forward"""
    cfg, batch, z = load_net_fixture(name)
    net = make_net(cfg, "fp32", mode="reference_init")
    assert int(net.state_dict()["lgcat_conv_e01.1.initialized"]) == 0
    x = synthetic_input(batch, cfg).to("cuda:0")
    outs = [o.clone() for o in net(x)]
    sd = net.state_dict()
    for k in [str(v) for v in z["actnorm_names"]]:
        assert int(sd[k + ".initialized"]) == 1
        dw = float(np.abs(sd[k + ".weight"].cpu().numpy() - z["actnorm_w/" + k]).max())
        db = float(np.abs(sd[k + ".bias"].cpu().numpy() - z["actnorm_b/" + k]).max())
        assert dw <= 2e-4 and db <= 2e-3 * max(1.0, float(np.abs(z["actnorm_b/" + k]).max())), "%s: dweight %.2e dbias %.2e" % (k, dw, db)
    worst = check_outputs(z, outs, 1e-3)                                       # the north_star fp32 bar
    again = net(x)                                                              # now initialised: same tables, same bits
    for a, b in zip(outs, again):
        assert torch.equal(a, b)
    print("%s fp32 with device ActNorm init: outputs max-abs vs reference %.2e" % (name, worst))


# fp16 path on the reference's own init distribution, measured in round 6 (profiles/r06_fp16_stage_errors_refinit.txt): outputs 6.4e-3 / 6.8e-3 max-abs against the reference
# (device ActNorm init / reference ActNorm parameters) -- torch's CPU fp16 autocast measures 8.3e-3 on the same weights; every stage within 1.3e-2 of its own rms.
# The bars sit 2x above that (the output bar was 6e-2 through round 5, ten times what the path delivers).
REFINIT_FP16_BAR = 1.5e-2
REFINIT_FP16_STAGE_REL_BAR = 3e-2      # sampled max-abs difference of a stage / its mean |value| (the fixture's abs-sum / numel)


def test_reference_init_weights_fp16_psnr_ssim_full512():
    """fp16 path on the reference's own init distribution (kaiming residual branches, N(0,1) position table): both with the
    reference's ActNorm parameters loaded and with ActNorm initialised on the device from fp16 activations; with the reference's
    parameters also EVERY stage against the reference's samples of it, relative to the stage's scale (where fp16 loses bits shows here first)"""
    name = "refinit_full512_nf24_hdr4"
    cfg, batch, z = load_net_fixture(name)
    sd = generate_state_dict(cfg, seed=0, mode="reference_init")
    x = synthetic_input(batch, cfg).to("cuda:0")
    net = make_net(cfg, "fp16", sd=sd)                                          # device init
    w1 = _fp16_vs_fixture(z, net(x), x, REFINIT_FP16_BAR)
    for k in [str(v) for v in z["actnorm_names"]]:
        sd[k + ".weight"], sd[k + ".bias"] = torch.from_numpy(z["actnorm_w/" + k]), torch.from_numpy(z["actnorm_b/" + k])
        sd[k + ".initialized"] = torch.tensor(1)
    net2 = make_net(cfg, "fp16", sd=sd)
    ops.tune("net.keep_stages", 1)
    try:
        w2 = _fp16_vs_fixture(z, net2(x), x, REFINIT_FP16_BAR)
        st = gpu_stages(net2, z)
    finally:
        ops.tune("net.keep_stages", 0)
    rows = []
    for n, t in st.items():
        smp = t.float().cpu().flatten()[sample_idx(n, t.numel())].numpy()
        scale = float(z["stage_abs/" + n]) / t.numel()
        rows.append((float(np.abs(smp - z["stage_smp/" + n]).max()) / scale, n))
    rows.sort(reverse=True)
    print("refinit full512 fp16: max-abs vs reference %.2e (device ActNorm init) / %.2e (reference ActNorm parameters); worst stages (sampled max-abs / mean |value|): %s"
          % (w1, w2, ", ".join("%s %.2e" % (n, r) for r, n in rows[:4])))
    assert rows[0][0] <= REFINIT_FP16_STAGE_REL_BAR, "stage %s: fp16 differs from the reference by %.2e of the stage's mean |value|" % (rows[0][1], rows[0][0])


def test_fp16_outputs_written_by_the_fused_tail_equal_the_rounded_float_outputs():
    """dec_ipt.output_f16 / cfen_net_set_output_f16 (round 6): the sharded run's wire type straight from the fused tail launch -- bit for bit the fp32 outputs rounded to
    nearest even, eager and replayed; refused where the fused tail does not run (fp32 nets, small images)"""
    cfg = NetConfig(24, 4, patch_size=32, load_size=256)
    net = make_net(cfg, "fp16")
    x = synthetic_input(2, cfg).to("cuda:0")
    want = [o.half() for o in net(x)]
    net.output_f16 = True
    got = net(x)
    assert all(g.dtype == torch.float16 and torch.equal(g, w) for g, w in zip(got, want))
    slab = torch.zeros(7 * 2 * 512 * 512, dtype=torch.float16, device="cuda:0")
    gid, outs = net.capture(x, out=slab)
    slab.zero_()
    net.replay(gid)
    torch.cuda.synchronize()
    assert all(torch.equal(g, w) for g, w in zip(outs, want))
    with pytest.raises(ValueError):
        net(x, out=torch.zeros(7 * 2 * 512 * 512, dtype=torch.float32, device="cuda:0"))
    net32 = make_net(cfg, "fp32")
    net32.output_f16 = True
    with pytest.raises(Exception):
        net32(x)
    # ... and where the plan falls back to the separate tail launches
    net2 = make_net(cfg, "fp16")
    net2.output_f16 = True
    ops.tune("net.tail_fused", 1)
    try:
        with pytest.raises(Exception):
            net2(x)
    finally:
        ops.tune("net.tail_fused", TAIL_FUSED_DEFAULT)


def test_uint8_input_equals_host_normalised_input():
    """--u8_input: (B,H,W,3) uint8 straight into the plan == ToTensor + Normalize(0.5, 0.5) on the host (data/base_dataset.py:44-46)"""
    cfg = NetConfig(24, 4, patch_size=8, load_size=64)
    net = make_net(cfg, "fp32")
    g = torch.Generator(); g.manual_seed(3)
    u8 = torch.randint(0, 256, (2, 128, 128, 3), generator=g, dtype=torch.uint8)
    xf = ((u8.permute(0, 3, 1, 2).float() / 255.0) - 0.5) / 0.5
    a = [o.clone() for o in net(xf.to("cuda:0"))]
    b = net(u8.to("cuda:0"))
    for p, q in zip(a, b):
        assert torch.equal(p, q)


@pytest.mark.parametrize("load_size,batch,dtype", [(256, 2, "fp16"), (64, 3, "fp16"), (64, 2, "fp32")])
def test_uint8_outputs_written_by_the_tail_equal_tensor2im_of_the_float_outputs(load_size, batch, dtype):
    """dec_ipt.output_u8 / cfen_net_set_output_u8: the three outputs as (B,H,W,3) uint8 images, byte for byte util.tensor2im of the float outputs
    (util/util.py:12-24 -- (x + 1) / 2 * 255, truncating cast, xs tiled to 3 channels; tests/golden/harness.npz pins that arithmetic).  fp16 nets
    write them from the 7x7 tails' epilogue (k_conv7_tz), fp32 nets take the cfen_tensor2im_u8 pass; eager = graph replay."""
    from cfen_vit_dehazing_amd import ops
    from cfen_vit_dehazing_amd.util import util
    cfg = NetConfig(24, 4, patch_size=load_size // 8, load_size=load_size)
    net = make_net(cfg, dtype)
    x = synthetic_input(batch, cfg).to("cuda:0")
    want = [o.clone() for o in net(x)]
    net.output_u8 = True
    got = [o.clone() for o in net(x)]
    assert all(net._native_u8[k] == (dtype == "fp16") for k in net._native_u8 if k[4])
    n = cfg.image_size
    for g, w in zip(got, want):
        assert g.dtype == torch.uint8 and tuple(g.shape) == (batch, n, n, 3)
        for b in range(batch):
            assert torch.equal(g[b], ops.tensor2im_u8(w[b].contiguous()))
    assert np.array_equal(util.tensor2im(got[2][0]), util.tensor2im(want[2][0]))       # what test.py saves
    if dtype == "fp16":
        gid, gouts = net.capture(x)
        net.replay(gid)
        torch.cuda.synchronize()
        for a, b in zip(gouts, got):
            assert torch.equal(a, b)
    with pytest.raises(ValueError):
        net(x, out=torch.empty(7 * batch * n * n, device="cuda:0"))
    net.output_u8 = False
    for a, b in zip(net(x), want):
        assert torch.equal(a, b)


# ---- sibling generators --model_G iid_hlgvit_crs_gd4_cfs / iid_hlgvit_crs_gd4 / iid_hlgvit_crs_gd4_cfs_v5 (models/networks_iid_hlgvit_crs_gd4_cfs.py,
# ..._crs_gd4.py, ..._cfs_v5.py): same kernels, other launch plans ---------

def _cfs_stages(net, z):
    st = {}
    xf = net.stage("head" if net.cfg.full_res else "ds_conv_e01")
    for n in [str(s) for s in z["stage_names"]]:
        if n.startswith("tail_"):
            continue
        t = net.stage(n)
        if n.startswith("lgcat_conv_d01"):
            t = t - xf                      # `+ xf` (cfs:669,823,977) rides on this stage's epilogue
        st[n] = t
    return st


@pytest.mark.parametrize("name", ["cfs_tiny_nf24_hdr4", "cfs_full256_nf24_hdr4", "crs_tiny_nf24_hdr4", "crs_full256_nf24_hdr4",
                                  "v5_tiny_nf24_hdr4", "v5_full512_nf24_hdr4"])
def test_sibling_variants_fp32_all_stages_and_fp16(name):
    cfg, batch, z = load_net_fixture(name)
    assert cfg.variant == name.split("_")[0] and cfg.image_size == (cfg.load_size if cfg.full_res else 2 * cfg.load_size)
    net = make_net(cfg, "fp32")
    x = synthetic_input(batch, cfg).to("cuda:0")
    outs = net(x)
    st = _cfs_stages(net, z)
    for nm, o in zip(("tail_R", "tail_S", "tail_D"), outs):
        st[nm] = o
    check_stages(z, st, 3e-4, rel_sum=2e-4)
    wo = check_outputs(z, outs, 1e-4)
    net16 = make_net(cfg, "fp16")
    w16 = check_outputs(z, net16(x), FP16_BAR)
    gid, gouts = net16.capture(x)
    net16.replay(gid)
    torch.cuda.synchronize()
    for a, b in zip(net16(x), gouts):
        assert torch.equal(a, b)
    print("%s: fp32 outputs max-abs vs reference %.2e, fp16 %.2e" % (name, wo, w16))


def test_v5_reference_init_actnorm_first_window_on_the_device():
    """v5's 48 ActNorm2d layers: the 24 inside the LViT modules are initialised from the first (top-left) window only, because the
    reference calls an LViT module once per window (networks_iid_hlgvit_crs_gd4_cfs_v5.py:403-440, 1139, 1190; models/actnorm.py:25-37)"""
    name = "refinit_v5_tiny_nf24_hdr4"
    cfg, batch, z = load_net_fixture(name)
    net = make_net(cfg, "fp32", mode="reference_init")
    x = synthetic_input(batch, cfg).to("cuda:0")
    outs = [o.clone() for o in net(x)]
    sd = net.state_dict()
    names = [str(v) for v in z["actnorm_names"]]
    assert len(names) == 48 and sum(".conv_shrink." in k or ".conv_extend." in k for k in names) == 24
    for k in names:
        assert int(sd[k + ".initialized"]) == 1
        dw = float(np.abs(sd[k + ".weight"].cpu().numpy() - z["actnorm_w/" + k]).max())
        db = float(np.abs(sd[k + ".bias"].cpu().numpy() - z["actnorm_b/" + k]).max())
        assert dw <= 2e-4 and db <= 2e-3 * max(1.0, float(np.abs(z["actnorm_b/" + k]).max())), "%s: dweight %.2e dbias %.2e" % (k, dw, db)
    worst = check_outputs(z, outs, 1e-3)
    for a, b in zip(outs, net(x)):
        assert torch.equal(a, b)
    print("%s fp32 with device ActNorm init (windowed for the in-LViT layers): outputs max-abs vs reference %.2e" % (name, worst))


def test_tile_major_gvit_weights_equal_row_major_to_rounding():
    """packing.pack_wtile (default, CFEN_WTILE): one K-step of a 96-feature GViT weight tile is one contiguous run; the <= 128-token GEMMs then
    run on k_gemm_dma instead of k_gemm_skinny (another order of the K sum), so the two layouts agree to rounding, not bitwise"""
    cfg, batch, z = load_net_fixture("small_nf24_hdr4")
    sd = generate_state_dict(cfg, seed=0)
    x = synthetic_input(batch, cfg).to("cuda:0")
    outs = {}
    for wt in (True, False):
        net = dec_ipt(cfg, compute_dtype="fp32")
        net.wtile = wt
        net.load_state_dict(sd, strict=True)
        outs[wt] = [o.clone() for o in net.to("cuda:0")(x)]
        check_outputs(z, outs[wt], 1e-4)
    assert max(float((a - b).abs().max()) for a, b in zip(outs[True], outs[False])) <= 2e-5


# ---- round 5: the pipelined inference driver (test.py --in_flight K) and the periodic --precision half checks -----------------------------------

def _cli_fixture(tmp_path, load_size, nimg, name="iid_hlgvit_crs_gd4_cfs_v3_pipe"):
    import os
    from PIL import Image
    cfg = NetConfig(24, 4, patch_size=load_size // 8, load_size=load_size)
    os.makedirs(tmp_path / "ckpt" / name)
    torch.save(cached_state_dict(cfg), tmp_path / "ckpt" / name / "32_net_G.pth")
    os.makedirs(tmp_path / "data" / "hazy")
    rs = np.random.RandomState(5)
    n = 2 * load_size
    yy, xx = np.mgrid[0:n, 0:n].astype(np.float32) / n
    for i in range(nimg):
        # smooth colour fields + a little noise: compressible like a photograph (pure noise makes the PNG encoder the whole test)
        base = np.stack([np.sin(6.0 * (xx * (i % 5 + 1) + yy)) * 90 + 128, np.cos(5.0 * (yy * (i % 3 + 1) - xx)) * 80 + 120, (xx + yy) * 100 + 20 + 4 * i], -1)
        a = np.clip(base + rs.randn(n, n, 3) * 3.0, 0, 255).astype(np.uint8)
        Image.fromarray(a).save(tmp_path / "data" / "hazy" / ("syn_%04d.png" % (i + 1)))
    return cfg, name


def _run_cli(tmp_path, name, load_size, res, extra):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "test.py"), "--dataroot", str(tmp_path / "data"), "--name", name, "--n_feats", "24",
           "--hidden_dim_ratio", "4", "--sb", "--which_epoch", "32", "--loadSize", str(load_size), "--patch_size", str(load_size // 8),
           "--checkpoints_dir", str(tmp_path / "ckpt"), "--results_dir", str(tmp_path / res)] + list(extra)
    env = dict(os.environ)
    env.pop("GPU_MAX_HW_QUEUES", None)            # the harness chooses it
    out = subprocess.run(cmd, cwd=str(tmp_path), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert out.returncode == 0, out.stdout[-3000:]
    return out.stdout, tmp_path / res / name / "test_32" / "images"


@pytest.mark.parametrize("load_size,nimg,flags", [
    (256, 26, ["--precision", "half", "--u8_input", "--out_all", "--batchSize", "4", "--half_guard_every", "4", "--nThreads", "2"]),   # the benchmarked geometry: uint8 in,
    # tensor2im bytes out of the tails' last launch, a ragged last batch, --precision half checks on batches 0 and 4
    (64, 24, ["--precision", "single", "--batchSize", "5"]),                                                                           # fp32 plans, float input, all four visuals
])
def test_pipelined_cli_writes_the_same_pngs(tmp_path, load_size, nimg, flags):
    """`test.py --in_flight 3` (replica launch plans replayed from hipGraphs on three streams, pinned asynchronous copies, writer threads) against the
    sequential loop of the same command line: every PNG byte for byte, on >= 24 images"""
    import os
    cfg, name = _cli_fixture(tmp_path, load_size, nimg)
    seq_log, seq_dir = _run_cli(tmp_path, name, load_size, "res_seq", flags)
    pipe_log, pipe_dir = _run_cli(tmp_path, name, load_size, "res_pipe", flags + ["--in_flight", "3", "--writers", "4"])
    assert "pipelined driver" in pipe_log and "'lanes_per_forward': 1" in pipe_log, pipe_log[-2000:]
    labels = ["fake_A"] if "--out_all" in flags else ["fake_A", "fake_R", "fake_S", "real_B"]
    want = sorted("syn_%04d_%s.png" % (i + 1, lab) for i in range(nimg) for lab in labels)
    assert sorted(os.listdir(seq_dir)) == want and sorted(os.listdir(pipe_dir)) == want
    for f in want:
        assert open(seq_dir / f, "rb").read() == open(pipe_dir / f, "rb").read(), f
    # round 6: the same with PNG encode in writer PROCESSES (forked before the model exists, images through a shared-memory ring of slots)
    proc_log, proc_dir = _run_cli(tmp_path, name, load_size, "res_proc", flags + ["--in_flight", "3", "--writer_procs", "3"])
    assert "pipelined driver" in proc_log and sorted(os.listdir(proc_dir)) == want
    for f in want:
        assert open(seq_dir / f, "rb").read() == open(proc_dir / f, "rb").read(), f
    if "half" in flags:
        assert "precision: half" in open(tmp_path / "res_pipe" / name / "test_32" / "precision.txt").read()
        assert "checked_batches: 0:" in open(tmp_path / "res_pipe" / name / "test_32" / "precision.txt").read()


def test_half_guard_catches_an_overflow_that_starts_on_a_later_batch(tmp_path):
    """--precision half with --half_guard_every 3: batches 0 .. 2 are fine, then the checkpoint's activations leave the fp16 range (stand-in for 'image 57
    overflows': an FFN scaled by 3e4 behind the model's back).  With -fno-honor-nans kernels that is finite garbage, not NaN -- the check of batch 3 sees
    it, the model continues in fp32, and the two unchecked batches before it are queued for redoing"""
    from cfen_vit_dehazing_amd.models import create_model
    from cfen_vit_dehazing_amd.options.test_options import TestOptions
    cfg = NetConfig(24, 4, patch_size=8, load_size=64)
    sd = cached_state_dict(cfg)
    ck = tmp_path / "ck" / "late_overflow"
    ck.mkdir(parents=True)
    torch.save(sd, ck / "latest_net_G.pth")
    opt = TestOptions().parse(['--dataroot', str(tmp_path), '--checkpoints_dir', str(tmp_path / "ck"), '--name', 'late_overflow', '--n_feats', '24',
                               '--hidden_dim_ratio', '4', '--patch_size', '8', '--loadSize', '64', '--sb', '--precision', 'half', '--half_guard_every', '3',
                               '--results_dir', str(tmp_path / "res")])
    model = create_model(opt)
    model.setup(opt)
    model.plan_half_guard(6)
    key = "localvit_encoder_01.encoder.layers.0.linear1.weight"
    for j in range(5):
        if j == 3:
            with torch.no_grad():
                dict(model.netG.named_parameters())[key].mul_(3e4)
            model.netG.invalidate()
        x = synthetic_input(2, cfg, seed0=2 * j)
        model.set_input({'B': x, 'B_paths': ['b%d_0.png' % j, 'b%d_1.png' % j]})
        model.test(opt)
        fa = model.get_current_visuals()['fake_A']
        assert torch.isfinite(fa.float()).all()
        if j < 3:
            assert model.netG.compute_dtype == torch.float16 and model.redo_paths == []
        else:
            assert model.netG.compute_dtype == torch.float32
        if j == 3:
            batch3_fake_A = fa.float().clone()
    assert [b for b, _ in model.half_guard_log] == [0, 3] and model.half_guard_log[0][1] <= model.HALF_GUARD_BAR and not model.half_guard_log[1][1] <= model.HALF_GUARD_BAR
    assert model.redo_paths == ['b1_0.png', 'b1_1.png', 'b2_0.png', 'b2_1.png']
    # batch 3 itself came out of the fp32 path: equal to a plain fp32 net on the same (scaled) weights
    sd2 = dict(sd)
    sd2[key] = sd[key] * 3e4
    ref = make_net(cfg, "fp32", sd=sd2)
    assert torch.equal(batch3_fake_A, ref(synthetic_input(2, cfg, seed0=6).to("cuda:0"))[2])
    txt = open(tmp_path / "res" / "late_overflow" / "test_latest" / "precision.txt").read()
    assert "precision: single" in txt and "fell_back_at_batch: 3" in txt
