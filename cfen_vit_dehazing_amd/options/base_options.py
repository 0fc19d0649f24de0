"""Command-line flags of the inference harness, same names / types / defaults as the reference's
options/base_options.py:8-250 for every flag the v3 path reads (SURVEY 5 "Config / flags"), plus the
harness flags of test.py.  Differences, all stated here:
  * --model / --model_G / --dataset_mode default to the values that actually reach the v3 generator
    (`dec_vit`, `iid_hlgvit_crs_gd4_cfs_v3`, `dec_vit`); the reference's defaults (`vit`,
    `iid_hlgvit_crs_gd4`, `vit`) select modules that do not import, so its README commands
    only work once these three are given.  Passing them explicitly works as in the reference.
  * --hidden_dim_ratio / --n_feats keep the reference defaults (6 / 32); the released checkpoints need
    `--n_feats 24 --hidden_dim_ratio 4|2` exactly as in the README.
  * --precision single|half (reference flag, base_options.py:114, unused there) selects the HIP compute
    type: fp32 MFMA or fp16 storage with fp32 accumulation.
  * flags of the training / IPT leftovers are accepted and ignored (parse_known_args), with a note.
"""
import argparse
import os

import torch

from ..util import util


class BaseOptions():
    def __init__(self):
        self.parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
        self.initialized = False

    def initialize(self):
        p = self.parser
        p.add_argument('--dataroot', required=True, help='path to images (should have subfolder hazy)')
        p.add_argument('--batchSize', type=int, default=1, help='input batch size')
        p.add_argument('--loadSize', type=int, default=256, help='edge of the half-resolution feature map (image edge / 2)')
        p.add_argument('--fineSize', type=int, default=128)
        p.add_argument('--input_nc', type=int, default=3)
        p.add_argument('--output_nc', type=int, default=3)
        p.add_argument('--model_G', type=str, default='iid_hlgvit_crs_gd4_cfs_v3', help='selects model to use for netG')
        p.add_argument('--gpu_ids', type=str, default='0', help='gpu ids: e.g. 0  0,1,2. use -1 for CPU (unsupported by the HIP path)')
        p.add_argument('--name', type=str, default='experiment_name', help='checkpoint directory name under --checkpoints_dir')
        p.add_argument('--dataset_mode', type=str, default='dec_vit')
        p.add_argument('--model', type=str, default='dec_vit')
        p.add_argument('--which_direction', type=str, default='AtoB')
        p.add_argument('--nThreads', default=0, type=int, help='# threads for loading data')
        p.add_argument('--checkpoints_dir', type=str, default='./checkpoints', help='models are saved here')
        p.add_argument('--sb', action='store_true', help='take images in order (otherwise randomly, as the reference does)')
        p.add_argument('--display_winsize', type=int, default=256)
        p.add_argument('--max_dataset_size', type=int, default=float("inf"))
        p.add_argument('--resize_or_crop', type=str, default='resize',
                       help="the reference default 'resize' matches no branch of get_transform: images are fed at native size")
        p.add_argument('--init_type', type=str, default='kaiming', help='network initialization [normal|xavier|kaiming|orthogonal]')
        p.add_argument('--verbose', action='store_true')
        p.add_argument('--suffix', default='', type=str)
        p.add_argument('--out_all', action='store_true', help='keep only the dehazed image (fake_A) among the outputs')
        p.add_argument('--seed', type=int, default=1)
        # transformer / generator geometry (base_options.py:96-110,191-201)
        p.add_argument('--patch_size', type=int, default=32, help='LViT window edge in feature-map pixels')
        p.add_argument('--rgb_range', type=int, default=255)
        p.add_argument('--n_colors', type=int, default=3)
        p.add_argument('--hidden_dim_ratio', type=int, default=6)
        p.add_argument('--n_feats', type=int, default=32)
        p.add_argument('--precision', type=str, default='single', choices=('single', 'half'),
                       help='HIP compute type: single = fp32 MFMA, half = fp16 storage / fp32 accumulate')
        p.add_argument('--no_half_guard', action='store_true',
                       help='(extension) with --precision half the first batch also runs in fp32 once and the model falls back to single when the '
                            'fp16 outputs differ by more than 1.5e-2 (range safety of a real checkpoint); this flag skips that check')
        p.add_argument('--half_guard_every', type=int, default=32,
                       help='(extension) with --precision half, repeat that fp32 comparison on every N-th batch of the run (0 = first batch only); all '
                            'ranks of a sharded run agree on the outcome, and the batches since the last passed check are redone in fp32 after a failure')
        p.add_argument('--in_flight', type=int, default=1,
                       help='(extension) batches kept in flight by test.py: 1 = the reference loop (set_input / test / save, one at a time); K > 1 = '
                            'the pipelined driver (cfen_vit_dehazing_amd/pipeline.py): K launch-plan replicas replayed from hipGraphs on K streams, pinned '
                            'asynchronous copies both ways, PNG decode in the DataLoader workers (--nThreads) and encode in --writers threads; the files '
                            'written are byte-identical to the sequential loop')
        p.add_argument('--writers', type=int, default=8, help='(extension) PNG encoder threads of the pipelined driver')
        p.add_argument('--png_compress_level', type=int, default=-1,
                       help='(extension) zlib level 0..9 of the result PNGs; -1 (default) = PIL\'s own default, the reference\'s files byte for byte. 1 encodes ~3x faster '
                            '(same pixels, larger files)')
        p.add_argument('--writer_procs', type=int, default=0,
                       help='(extension) PNG encoder PROCESSES of the pipelined driver instead of --writers threads (0 = threads): forked right after option parsing, images '
                            'handed over through shared memory; encode scales with the host cores (threads contend for the GIL around the compressor)')
        p.add_argument('--u8_input', action='store_true',
                       help='(extension) the dataset hands over uint8 HWC images and ToTensor + Normalize(0.5, 0.5) run on the device '
                            'inside the generator launch plan (12x fewer bytes over PCIe); results are identical')
        p.add_argument('--patch_dim', type=int, default=2)
        p.add_argument('--num_heads', type=int, default=4)
        p.add_argument('--num_layers', type=int, default=1)
        p.add_argument('--dropout_rate', type=float, default=0)
        p.add_argument('--no_norm', action='store_true')
        p.add_argument('--no_mlp', action='store_true')
        p.add_argument('--pos_every', action='store_true')
        p.add_argument('--no_pos', action='store_true')
        p.add_argument('--num_queries', type=int, default=1)
        self.initialized = True

    def parse(self, argv=None):
        if not self.initialized:
            self.initialize()
        opt, unknown = self.parser.parse_known_args(argv)
        if unknown:
            print('note: ignoring flags outside the inference path: %s' % ' '.join(unknown))
        opt.isTrain = self.isTrain
        str_ids = opt.gpu_ids.split(',')
        opt.gpu_ids = []
        for str_id in str_ids:
            id = int(str_id)
            if id >= 0:
                opt.gpu_ids.append(id)
        # one process per GPU: under `python -m torch.distributed.run --nproc-per-node N test.py ...` every rank takes the GPU of its
        # LOCAL_RANK (whatever --gpu_ids says) and its own slice of the dataset; the reference's counterpart is nn.DataParallel over
        # --gpu_ids (networks_iid_hlgvit_crs_gd4_cfs_v3.py:77-83)
        from ..parallel import dist_env
        opt.dist_rank, opt.dist_world, local = dist_env()
        if opt.dist_world > 1:
            if opt.gpu_ids != [local]:
                print('[rank %d] torch.distributed.run: --gpu_ids %s overridden by LOCAL_RANK -> GPU %d' % (opt.dist_rank, opt.gpu_ids, local))
            opt.gpu_ids = [local]
        if opt.in_flight < 1:
            raise ValueError('--in_flight must be >= 1')
        if not -1 <= opt.png_compress_level <= 9:
            raise ValueError('--png_compress_level must be -1 (PIL default) or 0..9')
        from ..util import util as _util
        _util.PNG_COMPRESS_LEVEL = None if opt.png_compress_level < 0 else opt.png_compress_level      # set before any writer process is forked
        if opt.in_flight > 1:
            # the pipelined driver keeps K forwards on K streams; the HIP runtime deals a process's streams onto GPU_MAX_HW_QUEUES hardware queues
            # (default 4) and two busy streams on one queue run one behind the other (4 in flight: 2.51 ms / step on 4 queues, 2.10 on 8,
            # profiles/r04_ab_hw_queues.txt).  The harness, not the user, sets it -- here, before the first HIP call of the process (set_device below)
            os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
        if len(opt.gpu_ids) > 0 and torch.cuda.is_available():
            torch.cuda.set_device(opt.gpu_ids[0])
        args = vars(opt)
        if opt.dist_rank == 0:
            print('------------ Options -------------')
            for k, v in sorted(args.items()):
                print('%s: %s' % (str(k), str(v)))
            print('-------------- End ----------------')
        if opt.suffix:
            suffix = ('_' + opt.suffix.format(**vars(opt))) if opt.suffix != '' else ''
            opt.name = opt.name + suffix
        expr_dir = os.path.join(opt.checkpoints_dir, opt.name)
        util.mkdirs(expr_dir)
        if opt.dist_rank == 0:
            with open(os.path.join(expr_dir, 'opt.txt'), 'wt') as opt_file:
                opt_file.write('------------ Options -------------\n')
                for k, v in sorted(args.items()):
                    opt_file.write('%s: %s\n' % (str(k), str(v)))
                opt_file.write('-------------- End ----------------\n')
        self.opt = opt
        return self.opt
