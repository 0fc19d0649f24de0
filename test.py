#!/usr/bin/env python3
"""Inference driver with the reference's CLI (test.py:19-63 there):

    python test.py --dataroot <dir with hazy/> --name iid_hlgvit_crs_gd4_cfs_v3_reside --n_feats 24 \
                   --hidden_dim_ratio 4 --sb --out_all --which_epoch 32

Loads checkpoints/<name>/<which_epoch>_net_G.pth (the reference's own file format), runs the HIP generator on
every image of <dataroot>/hazy and writes results/<name>/<phase>_<which_epoch>/images/<stem>_fake_A.png.
"""
import logging
import os
import sys
import time

_T_START = time.perf_counter()
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch

from cfen_vit_dehazing_amd.data import CreateDataLoader
from cfen_vit_dehazing_amd.models import create_model
from cfen_vit_dehazing_amd.options.test_options import TestOptions
from cfen_vit_dehazing_amd.util import html
from cfen_vit_dehazing_amd.util.visualizer import save_images

def _rerun_in_fp32(opt, model, image_dir, paths):
    """images whose files came from fp16 forwards later found unsafe (--precision half, a periodic check failed): the model has switched to fp32, run them again"""
    import torch.utils.data
    from cfen_vit_dehazing_amd import data as cdata
    ds = cdata.DECVITDATA()
    ds.initialize(opt)
    ds.B_paths, ds.B_size = list(paths), len(paths)
    for data in torch.utils.data.DataLoader(ds, batch_size=opt.batchSize, shuffle=False, num_workers=0):
        model.set_input(data)
        model.test(opt)
        visuals = model.get_current_visuals()
        if opt.out_all:
            for item in [k for k in visuals if 'fake_A' not in k]:
                del visuals[item]
        save_images(image_dir, visuals, model.get_image_paths(), aspect_ratio=opt.aspect_ratio, width=opt.display_winsize)


if __name__ == '__main__':
    _startup = {"imports": round(time.perf_counter() - _T_START, 2)}      # where the seconds in front of the first batch go (printed below; tools/cli_throughput.py)
    _t = time.perf_counter()
    opt = TestOptions().parse()   # --in_flight > 1 also exports GPU_MAX_HW_QUEUES=8 there, in front of its torch.cuda.set_device (the process's first HIP call: the forks
                                  # below come after it, but before the process holds any GPU memory -- that, not the initialised runtime, is what made late forks cost 10-16 s)
    opt.serial_batches = True   # no shuffle
    opt.no_flip = True          # no flip
    opt.display_id = -1         # no visdom display
    if opt.dist_world > 1:
        # one process per GPU under torch.distributed.run: every rank writes its own files, so the data path needs no collective; the ranks only
        # agree on the --precision half checks (models/model_iid_dehazing.py), over a CPU (gloo) group
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=opt.dist_rank, world_size=opt.dist_world)
    data_loader = CreateDataLoader(opt)
    data_loader.start_workers()          # fork the --nThreads decode workers BEFORE the process holds its GPU working set (data/__init__.py)
    if opt.in_flight > 1 and getattr(opt, 'writer_procs', 0) > 0:
        # ... and the PNG writer processes (pipeline.start_writer_processes: a shared-memory ring of image slots made before the fork)
        from cfen_vit_dehazing_amd.config import VARIANTS, FULL_RES_VARIANTS
        from cfen_vit_dehazing_amd import pipeline as _pipeline
        _edge = opt.loadSize if VARIANTS.get(opt.model_G, 'v3') in FULL_RES_VARIANTS else 2 * opt.loadSize
        _pipeline.start_writer_processes(opt.writer_procs, _edge, batch=opt.batchSize, labels=1 if opt.out_all else 4, in_flight=opt.in_flight)
    dataset = data_loader.load_data()
    _startup["options_loader_and_forks"] = round(time.perf_counter() - _t, 2)
    _t = time.perf_counter()
    model = create_model(opt)
    model.setup(opt)
    _startup["create_model_and_checkpoint_load"] = round(time.perf_counter() - _t, 2)
    _t = time.perf_counter()
    web_dir = os.path.join(opt.results_dir, opt.name, '%s_%s' % (opt.phase, opt.which_epoch))
    webpage = html.HTML(web_dir, 'Experiment = %s, Phase = %s, Epoch = %s' % (opt.name, opt.phase, opt.which_epoch))
    n_images = min(len(data_loader), len(data_loader.dataset))
    n_batches = min(opt.how_many, -(-n_images // opt.batchSize))
    model.plan_half_guard(n_batches)
    runner = None
    if opt.in_flight > 1:
        from cfen_vit_dehazing_amd.pipeline import PipelinedRunner
        runner = PipelinedRunner(model, opt, webpage.get_image_dir())          # switches the generator to the serial launch plan: before anything is built
    if n_images > 0 and hasattr(model, 'warm_up'):
        model.warm_up(min(opt.batchSize, n_images), getattr(opt, 'u8_input', False))      # every allocation of the run before the DataLoader forks its workers
    if opt.in_flight > 1:
        # the pipelined driver: --in_flight batches on as many launch-plan replicas / streams / hardware queues, hipGraph replay, pinned asynchronous
        # copies, PNG encode in --writers threads -- the loop bench.py's throughput presupposes, for real files; byte-identical files
        print('pipelined driver: %d batches in flight, plan %s' % (opt.in_flight, model.netG.plan_info()))
        if n_images > 0 and model.actnorm_ready():
            n = model.netG.cfg.image_size
            sizes = [min(opt.batchSize, n_images)] + ([n_images % opt.batchSize] if n_images > opt.batchSize and n_images % opt.batchSize and n_batches * opt.batchSize >= n_images else [])
            u8 = getattr(opt, 'u8_input', False)
            runner.warm_up([(b, n, n, 3) if u8 else (b, 3, n, n) for b in sizes], torch.uint8 if u8 else torch.float32)
        _startup["weight_packing_plans_graphs_and_warm_up"] = round(time.perf_counter() - _t, 2)
        print('startup seconds %s' % _startup)
        stats = runner.run(dataset, opt.how_many)
        runner.close()
        print('pipelined driver: %d images in %.2f s = %.1f images/s file to file (%d batches replayed from graphs, %d through the sequential path)'
              % (stats['images'], stats['seconds'], stats['images'] / max(stats['seconds'], 1e-9), stats['graph_batches'], stats['sequential_batches']))
        print('pipelined driver: main-thread seconds %s' % stats['main_thread_seconds'])
        if 'writer_ring' in stats:
            print('pipelined driver: writer processes read from %s' % stats['writer_ring'])
        print('pipelined driver: sequential-path seconds %s' % stats.get('sequential_seconds'))
    else:
        _startup["weight_packing_plans_and_warm_up"] = round(time.perf_counter() - _t, 2)
        print('startup seconds %s' % _startup)
        t_loop, n_img = time.perf_counter(), 0
        for i, data in enumerate(dataset):
            if i >= opt.how_many:
                break
            n_img += len(data['B_paths'])
            model.set_input(data)
            model.test(opt)
            visuals = model.get_current_visuals()
            if opt.out_all:                       # keep only the dehazed image
                for item in [k for k in visuals if 'fake_A' not in k]:
                    del visuals[item]
            img_path = model.get_image_paths()
            if i % 5 == 0:
                logging.info('processing (%04d)-th image...' % (i * opt.batchSize))
            save_images(webpage.get_image_dir(), visuals, img_path, aspect_ratio=opt.aspect_ratio, width=opt.display_winsize)
        t_loop = time.perf_counter() - t_loop
        print('sequential loop: %d images in %.2f s = %.1f images/s file to file' % (n_img, t_loop, n_img / max(t_loop, 1e-9)))
    model.finish_half_guard()
    if model.redo_paths:
        print('redoing %d images in fp32 (a --precision half check failed after they were written)' % len(model.redo_paths))
        _rerun_in_fp32(opt, model, webpage.get_image_dir(), model.redo_paths)
    if opt.in_flight > 1 and getattr(opt, 'writer_procs', 0) > 0:
        _pipeline.stop_writer_processes()
    if opt.dist_world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
