"""DECHLGVIT of the reference (models/model_iid_dehazing.py:14-156), inference subset.  `netG` is the
HIP-backed generator; `forward` is `[fake_R, fake_S, fake_A] = netG(real_B)` (model_iid_dehazing.py:140-143)."""
import torch

from .base_model import BaseModel


class DECHLGVIT(BaseModel):
    def name(self):
        return 'DECHLGVIT'

    def initialize(self, opt):
        BaseModel.initialize(self, opt)
        if self.isTrain:
            raise NotImplementedError("training is outside the MI355X inference path")
        self.visual_names = ['fake_A', 'real_B', 'fake_R', 'fake_S']
        self.model_names = ['G']
        if opt.model_G == 'iid_hlgvit_crs_gd4_cfs_v3':
            from . import networks_iid_hlgvit_crs_gd4_cfs_v3
            self.netG = networks_iid_hlgvit_crs_gd4_cfs_v3.define_G(opt, None)
        elif opt.model_G == 'iid_hlgvit_crs_gd4_cfs':             # models/model_iid_dehazing.py:84-86
            from . import networks_iid_hlgvit_crs_gd4_cfs
            self.netG = networks_iid_hlgvit_crs_gd4_cfs.define_G(opt, None)
        elif opt.model_G in ('iid_hlgvit_crs_gd4', 'iid_hlgvit_crs_gd4_cfs_v5'):     # models/model_iid_dehazing.py:50-53, 93-95
            from .. import hipnet                             # same module class, the variant comes from opt.model_G (config.VARIANTS)
            self.netG = hipnet.define_G(opt, None)
        # any other --model_G leaves netG undefined, as the reference's if/elif chain does (-> AttributeError)

    def set_input(self, input):
        B = input['B'].to(self.device)                      # hazy image, H2D copy
        self.image_paths = input['B_paths']
        if B.dtype == torch.uint8:
            # --u8_input: (B,H,W,3) uint8 goes to the generator as it is (normalised by the plan's first launch);
            # `real_B` of get_current_visuals stays what the reference shows: the normalised float image
            self._net_in = B
            self.real_B = (B.permute(0, 3, 1, 2).float() / 255.0 - 0.5) / 0.5
        else:
            self._net_in = self.real_B = B

    HALF_GUARD_BAR = 3e-2     # max-abs difference of the fp16 outputs (tanh values in (-1, 1)) from the fp32 path on the first batch

    def setup(self, opt):
        BaseModel.setup(self, opt)
        # fp16 range safety of a REAL checkpoint (ActNorm scales, K = 6144 FFN sums) is unknown until its weights are here: with --precision
        # half the first batch runs through the exact-fp32 path once as well (reference loader: models/base_model.py:114-131)
        self._half_guard = getattr(opt, 'precision', 'single') == 'half' and not getattr(opt, 'no_half_guard', False)
        # test.py only turns the outputs into PNGs (util.tensor2im, util/util.py:12-24): the generator writes those bytes itself -- (B,H,W,3) uint8
        # from the tails' last launch where the geometry allows, a device pass elsewhere (hipnet.dec_ipt.output_u8) -- instead of fp32 planes
        self._u8_out = not getattr(opt, 'isTrain', False) and getattr(opt, 'phase', 'test') == 'test' and hasattr(self.netG, 'output_u8')

    def _guard_dir(self):
        import os
        o = self.opt
        d = os.path.join(getattr(o, 'results_dir', './results/'), getattr(o, 'name', 'experiment'), '%s_%s' % (getattr(o, 'phase', 'test'), getattr(o, 'which_epoch', 'latest')))
        os.makedirs(d, exist_ok=True)
        return d

    def _agree_on_worst(self, worst, timeout=120.0):
        import os
        import time
        world, rank = getattr(self.opt, 'dist_world', 1), getattr(self.opt, 'dist_rank', 0)
        if world <= 1:
            return worst
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            t = torch.tensor([worst if worst == worst else float('inf')], dtype=torch.float64, device=self.device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            return float(t.item())
        d = self._guard_dir()
        run = os.environ.get('TORCHELASTIC_RUN_ID', 'run')
        with open(os.path.join(d, '.half_guard_%s_rank%d' % (run, rank)), 'w') as f:
            f.write(repr(float(worst)))
        vals, t_end = {}, time.time() + timeout
        while len(vals) < world and time.time() < t_end:
            for r in range(world):
                p = os.path.join(d, '.half_guard_%s_rank%d' % (run, r))
                if r not in vals and os.path.exists(p):
                    txt = open(p).read().strip()
                    if txt:
                        vals[r] = float(txt)
            if len(vals) < world:
                time.sleep(0.05)
        if len(vals) < world:      # a rank never reported: be safe, everyone who notices falls back
            return float('inf')
        return max(vals.values())

    def _record_precision(self, chosen, worst):
        import os
        if getattr(self.opt, 'dist_rank', 0) == 0:
            try:
                with open(os.path.join(self._guard_dir(), 'precision.txt'), 'w') as f:
                    f.write('precision: %s\nhalf_guard_max_abs: %r\nbar: %g\n' % (chosen, worst, self.HALF_GUARD_BAR))
            except OSError as e:          # a read-only results directory must not cost the run
                print('note: could not record the chosen precision (%s)' % e)

    def forward(self):
        if getattr(self, '_half_guard', False):
            self._half_guard = False
            self.netG.output_u8 = False                   # the guard compares the float outputs
            self.netG.set_compute_dtype('fp32')
            ref = [t.clone() for t in self.netG(self._net_in)]
            self.netG.set_compute_dtype('fp16')
            out = self.netG(self._net_in)
            worst = max(float((a - b).abs().max()) if bool(torch.isfinite(a).all()) else float('inf') for a, b in zip(out, ref))
            # one process per GPU (test.py under torch.distributed.run): every rank looks at its OWN first image -- the ranks agree on the worst
            # of them, so a result set is never silently mixed-precision.  Without a process group (the launcher is used for its environment
            # only: every rank writes its own files) the decision goes through a file in the results directory.
            worst = self._agree_on_worst(worst)
            self.half_guard_max_abs = worst
            self._record_precision('single' if not worst <= self.HALF_GUARD_BAR else 'half', worst)
            if not worst <= self.HALF_GUARD_BAR:
                print('warning: --precision half differs from the fp32 path by %.3g max-abs on the first batch (bar %.0e): this checkpoint is not '
                      'fp16-safe, continuing with --precision single' % (worst, self.HALF_GUARD_BAR))
                self.netG.set_compute_dtype('fp32')
                [self.fake_R, self.fake_S, self.fake_A] = ref
                return
        if getattr(self, '_u8_out', False):
            self.netG.output_u8 = True
        [self.fake_R, self.fake_S, self.fake_A] = self.netG(self._net_in)
