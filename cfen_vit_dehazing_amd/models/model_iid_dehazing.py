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
        B = input['B']
        if not B.is_cuda and self.device.type == 'cuda':
            # H2D through a pinned staging buffer the model keeps per batch shape (allocated by warm_up, before the DataLoader forks its workers).
            # Measured on MI355X / ROCm 7.2: once forked workers exist, a copy from PAGEABLE host memory (the worker's shared-memory batch, or a plain
            # clone of it) to the device took ~8 s per 6 MB batch -- 6.5 images/s for the whole sequential loop with --nThreads 4 against 35 with
            # --nThreads 0 (profiles/r05_cli_throughput.json); from pinned memory it is the PCIe time
            B = self._staged(B).to(self.device, non_blocking=True)
        else:
            B = B.to(self.device)
        self.image_paths = input['B_paths']
        if B.dtype == torch.uint8:
            # --u8_input: (B,H,W,3) uint8 goes to the generator as it is (normalised by the plan's first launch);
            # `real_B` of get_current_visuals stays what the reference shows: the normalised float image
            self._net_in = B
            self.real_B = (B.permute(0, 3, 1, 2).float() / 255.0 - 0.5) / 0.5
        else:
            self._net_in = self.real_B = B

    def _staged(self, B):
        pins = self.__dict__.setdefault('_pins', {})
        key = (tuple(B.shape), B.dtype)
        if key not in pins:
            if len(pins) >= 4:
                pins.clear()
            pins[key] = torch.empty(B.shape, dtype=B.dtype).pin_memory()
        if torch.cuda.is_available():
            torch.cuda.current_stream().synchronize()       # the previous batch's copy out of this buffer has completed
        pins[key].copy_(B)
        return pins[key]

    # max-abs difference of the fp16 outputs (tanh values in (-1, 1)) from the fp32 path on a checked batch.  Measured (round 6): 9e-4 on the seeded "trained-like" weights,
    # 6.4e-3 .. 8.0e-3 on the reference's own init distribution (torch's CPU fp16 autocast: 8.3e-3); the bar is the test bar of that distribution
    # (tests/test_hip_net.py REFINIT_FP16_BAR) -- it was 3e-2, four times looser than anything the path produces when it is healthy
    HALF_GUARD_BAR = 1.5e-2

    def setup(self, opt):
        BaseModel.setup(self, opt)
        # fp16 range safety of a REAL checkpoint (ActNorm scales, K = 6144 FFN sums) is unknown until its weights are here, and it depends on the
        # IMAGE: with --precision half the first batch and then every --half_guard_every-th batch of the run also go through the exact-fp32 path
        # (reference loader: models/base_model.py:114-131).  The kernels are built with -fno-honor-nans (build.py), so an fp16 overflow inside the net
        # comes out as finite garbage, not NaN: comparing with the fp32 path is the only check that sees it.
        self._half_guard = getattr(opt, 'precision', 'single') == 'half' and not getattr(opt, 'no_half_guard', False)
        self._guard_every = max(0, int(getattr(opt, 'half_guard_every', 32)))
        self._batch_index = 0            # index of the batch the next forward() works on (the pipelined driver sets it: not every batch passes here)
        self._since_check = []           # image paths of the fp16 batches since the last passed check (pipeline.py adds the ones it runs itself)
        self.redo_paths = []             # images whose files were written by fp16 forwards later found unsafe: test.py runs them again in fp32
        self.half_guard_log = []         # (batch index, agreed max-abs) of every check
        self._checks_planned = None      # checks EVERY rank takes part in (plan_half_guard), None = only the first
        self._checks_done = 0
        # test.py only turns the outputs into PNGs (util.tensor2im, util/util.py:12-24): the generator writes those bytes itself -- (B,H,W,3) uint8
        # from the tails' last launch where the geometry allows, a device pass elsewhere (hipnet.dec_ipt.output_u8) -- instead of fp32 planes
        self._u8_out = not getattr(opt, 'isTrain', False) and getattr(opt, 'phase', 'test') == 'test' and hasattr(self.netG, 'output_u8')

    def _guard_dir(self):
        import os
        o = self.opt
        d = os.path.join(getattr(o, 'results_dir', './results/'), getattr(o, 'name', 'experiment'), '%s_%s' % (getattr(o, 'phase', 'test'), getattr(o, 'which_epoch', 'latest')))
        os.makedirs(d, exist_ok=True)
        return d

    # ---- which batches are checked, and how the ranks of a sharded run stay in step --------------------------------------------------------------
    def guard_due(self, batch_index):
        """does the batch with this index (0-based, per rank) go through the fp32 comparison?"""
        if not getattr(self, '_half_guard', False):
            return False
        if getattr(self.opt, 'dist_world', 1) > 1 and self._checks_planned is None:
            return batch_index == 0          # no plan_half_guard(): the ranks' later checks could not be paired up -- only the first one is common to all
        return batch_index == 0 or (self._guard_every > 0 and batch_index % self._guard_every == 0)

    def checks_for(self, n_batches):
        if n_batches <= 0:
            return 0
        return 1 + ((n_batches - 1) // self._guard_every if self._guard_every > 0 else 0)

    def plan_half_guard(self, n_batches):
        """Call once before the first batch of a sharded run: ranks may hold slices that differ by a batch, so they agree on the number of checks the
        LONGEST slice makes; a rank with fewer takes part in the remaining ones from finish_half_guard() (every check is one collective on all ranks)."""
        if not getattr(self, '_half_guard', False):
            return
        mine = self.checks_for(n_batches)
        self._checks_planned = int(self._all_reduce_max(float(mine))) if getattr(self.opt, 'dist_world', 1) > 1 else mine

    def finish_half_guard(self):
        """Call after the last batch: joins the checks other ranks still make; a failure agreed on there marks this rank's unchecked batches as well."""
        if not getattr(self, '_half_guard', False) or self._checks_planned is None:
            return
        while self._checks_done < self._checks_planned:
            worst = self._agree_on_worst(0.0)
            self._checks_done += 1
            if not worst <= self.HALF_GUARD_BAR:
                self._fall_back(worst, batch_index=None)
                break

    def _process_group(self):
        """the harness's process group (test.py creates a gloo group under torch.distributed.run; the launcher's environment is enough for env://)"""
        import torch.distributed as dist
        if not dist.is_available():
            raise RuntimeError('--precision half on %d ranks needs torch.distributed for the ranks to agree on the guard' % getattr(self.opt, 'dist_world', 1))
        if not dist.is_initialized():
            import os
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            dist.init_process_group('gloo', rank=getattr(self.opt, 'dist_rank', 0), world_size=getattr(self.opt, 'dist_world', 1))
        return dist

    def _all_reduce_max(self, value):
        dist = self._process_group()
        dev = self.device if dist.get_backend() == 'nccl' else 'cpu'
        t = torch.tensor([value if value == value else float('inf')], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def _agree_on_worst(self, worst):
        """one process per GPU (test.py under torch.distributed.run): every rank looks at its OWN batch -- the ranks agree on the worst of them, so a
        result set is never silently mixed-precision.  An all-reduce on the harness's process group (round 4 fell back to files in the results
        directory: stale files of an earlier run could make ranks decide differently -- ADVICE r04)."""
        if getattr(self.opt, 'dist_world', 1) <= 1:
            return worst
        return self._all_reduce_max(worst)

    def _record_precision(self, chosen, worst, batch_index=0):
        import os
        if getattr(self.opt, 'dist_rank', 0) == 0:
            try:
                with open(os.path.join(self._guard_dir(), 'precision.txt'), 'w') as f:
                    f.write('precision: %s\nhalf_guard_max_abs: %r\nbar: %g\nchecked_batches: %s\n' % (
                        chosen, worst, self.HALF_GUARD_BAR, ' '.join('%d:%.3g' % (b if b is not None else -1, w) for b, w in self.half_guard_log)))
                    if chosen == 'single' and batch_index:
                        f.write('fell_back_at_batch: %s\nredone_in_fp32: %d images\n' % (batch_index, len(self.redo_paths)))
            except OSError as e:          # a read-only results directory must not cost the run
                print('note: could not record the chosen precision (%s)' % e)

    def note_unchecked(self, paths):
        """the pipelined driver ran this batch in fp16 without passing through forward(): it counts as unchecked until the next check passes"""
        if getattr(self, '_half_guard', False):
            self._since_check.append(list(paths))

    def _fall_back(self, worst, batch_index):
        print('warning: --precision half differs from the fp32 path by %.3g max-abs on batch %s (bar %.0e): this checkpoint is not fp16-safe on '
              'these images, continuing with --precision single%s' % (worst, 'of another rank' if batch_index is None else batch_index, self.HALF_GUARD_BAR,
                                                                      '; %d images since the last passed check will be redone in fp32' % sum(len(p) for p in self._since_check)
                                                                      if self._since_check else ''))
        self._half_guard = False
        self.redo_paths += [p for batch in self._since_check for p in batch]
        self._since_check = []
        self.netG.set_compute_dtype('fp32')
        self._record_precision('single', worst, batch_index if batch_index is not None else -1)

    def warm_up(self, batch_size, u8_input=False):
        """Build everything the run will need -- packed weights, launch plans and workspaces of every compute type in play -- on a dummy batch, BEFORE the
        DataLoader forks its workers: device allocations made while forked children hold the process's GPU mappings took seconds each on MI355X / ROCm 7.2
        (the first --precision half check of a pipelined run with 16 workers: 18 s against < 1 s; profiles/r05_cli_throughput.json).  No guard state changes."""
        if not self.actnorm_ready():
            return False           # the first REAL batch must initialise those layers (models/actnorm.py:25-37), not a dummy
        n = self.netG.cfg.image_size
        shape = (batch_size, n, n, 3) if u8_input else (batch_size, 3, n, n)
        x = self._staged(torch.zeros(shape, dtype=torch.uint8 if u8_input else torch.float32)).to(self.device)
        with torch.no_grad():
            if getattr(self, '_half_guard', False):
                self.netG.output_u8 = False
                self.netG.set_compute_dtype('fp32')
                self.netG(x)
                self.netG.set_compute_dtype('fp16')
                self.netG(x)
            if getattr(self, '_u8_out', False):
                self.netG.output_u8 = True
            self.netG(x)
        torch.cuda.synchronize()
        return True

    def actnorm_ready(self):
        return all(int(b) != 0 for k, b in self.netG.named_buffers() if k.endswith('initialized'))

    def _settle_check(self, j, worst):
        """one check of batch j: the ranks agree on the worst difference; passed -> the batches since the previous check are cleared, failed -> fp32 from here on"""
        worst = self._agree_on_worst(worst)
        self._checks_done += 1
        self.half_guard_max_abs = worst
        self.half_guard_log.append((j, worst))
        if not worst <= self.HALF_GUARD_BAR:
            self._fall_back(worst, j)
            return False
        self._since_check = []
        self._record_precision('half', worst, j)
        return True

    def forward(self):
        j = self._batch_index
        self._batch_index = j + 1
        if self.guard_due(j):
            self.netG.output_u8 = False                   # the guard compares the float outputs
            self.netG.set_compute_dtype('fp32')
            ref = [t.clone() for t in self.netG(self._net_in)]
            self.netG.set_compute_dtype('fp16')
            out = self.netG(self._net_in)
            worst = max(float((a - b).abs().max()) if bool(torch.isfinite(a).all()) else float('inf') for a, b in zip(out, ref))
            if not self._settle_check(j, worst):
                [self.fake_R, self.fake_S, self.fake_A] = ref
                return
        elif getattr(self, '_half_guard', False):
            self._since_check.append(list(self.image_paths))
        if getattr(self, '_u8_out', False):
            self.netG.output_u8 = True
        [self.fake_R, self.fake_S, self.fake_A] = self.netG(self._net_in)
