"""The model wrapper test.py talks to (reference models/base_model.py:9-154, inference surface only).

What the harness uses and this class keeps, with the reference's names and file conventions: `initialize(opt)` (device from
`opt.gpu_ids`, `save_dir = <checkpoints_dir>/<name>`), `setup(opt)` (loads `<which_epoch>_net_<X>.pth` for every X in `model_names`,
strict key matching, keys without a `module.` prefix), `set_input`, `test` (forward under no_grad), `eval`, `get_current_visuals`
(`visual_names` -> attributes, in order), `get_image_paths`, `save_networks`.  Optimisers, schedulers, losses and
`update_learning_rate` belong to training and are not here."""
import os
from collections import OrderedDict

import torch

_STALE_NORM_KEYS = ('running_mean', 'running_var', 'num_batches_tracked')   # InstanceNorm buffers of pre-0.4 checkpoints (base_model.py:103-111)


class BaseModel():
    def name(self):
        return 'BaseModel'

    def initialize(self, opt):
        self.opt, self.isTrain, self.gpu_ids = opt, opt.isTrain, opt.gpu_ids
        self.device = torch.device('cuda', self.gpu_ids[0]) if self.gpu_ids else torch.device('cpu')
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)
        self.model_names, self.visual_names, self.loss_names, self.image_paths = [], [], [], []

    def _nets(self):
        return [(n, getattr(self, 'net' + n)) for n in self.model_names if isinstance(n, str)]

    def _checkpoint(self, which_epoch, name):
        return os.path.join(self.save_dir, '%s_net_%s.pth' % (which_epoch, name))

    # ---- what test.py calls, in its order ----
    def setup(self, opt):
        if self.isTrain:
            raise NotImplementedError("training is outside the MI355X inference path")
        self.load_networks(opt.which_epoch)
        for name, net in self._nets():
            if opt.verbose:
                print(net)
            print('[Network %s] %.3f M parameters' % (name, sum(p.numel() for p in net.parameters()) / 1e6))

    def set_input(self, input):
        self.input = input

    def forward(self):
        pass

    def test(self, opt=None):
        with torch.no_grad():
            self.forward()

    def eval(self):
        for _, net in self._nets():
            net.eval()

    def get_current_visuals(self):
        return OrderedDict((n, getattr(self, n)) for n in self.visual_names if isinstance(n, str))

    def get_image_paths(self):
        return self.image_paths

    # ---- checkpoints: the reference's own files load and come back out unchanged ----
    def load_networks(self, which_epoch):
        for name, net in self._nets():
            path = self._checkpoint(which_epoch, name)
            print('loading the model from %s' % path)
            state = torch.load(path, map_location=str(self.device))
            known = net.state_dict()
            for key in [k for k in state if k.endswith(_STALE_NORM_KEYS) and k not in known]:
                del state[key]
            net.load_state_dict(state)               # strict, as the reference

    def save_networks(self, which_epoch):
        for name, net in self._nets():
            torch.save({k: v.cpu() for k, v in net.state_dict().items()}, self._checkpoint(which_epoch, name))
