"""BaseModel of the reference (models/base_model.py:9-154), inference subset: device pick, checkpoint
load/save with the reference's file naming and strict key matching, set_input/test/get_current_visuals."""
import os
from collections import OrderedDict

import torch


class BaseModel():
    def name(self):
        return 'BaseModel'

    def initialize(self, opt):
        self.opt = opt
        self.gpu_ids = opt.gpu_ids
        self.isTrain = opt.isTrain
        self.device = torch.device('cuda:{}'.format(self.gpu_ids[0])) if self.gpu_ids else torch.device('cpu')
        self.save_dir = os.path.join(opt.checkpoints_dir, opt.name)
        self.loss_names = []
        self.model_names = []
        self.visual_names = []
        self.image_paths = []

    def set_input(self, input):
        self.input = input

    def forward(self):
        pass

    def setup(self, opt):
        if self.isTrain:
            raise NotImplementedError("training is outside the MI355X inference path")
        self.load_networks(opt.which_epoch)
        self.print_networks(opt.verbose)

    def eval(self):
        for name in self.model_names:
            if isinstance(name, str):
                getattr(self, 'net' + name).eval()

    def test(self, opt=None):
        with torch.no_grad():
            self.forward()

    def get_image_paths(self):
        return self.image_paths

    def get_current_visuals(self):
        visual_ret = OrderedDict()
        for name in self.visual_names:
            if isinstance(name, str):
                visual_ret[name] = getattr(self, name)
        return visual_ret

    def save_networks(self, which_epoch):
        for name in self.model_names:
            if isinstance(name, str):
                save_path = os.path.join(self.save_dir, '%s_net_%s.pth' % (which_epoch, name))
                net = getattr(self, 'net' + name)
                torch.save({k: v.cpu() for k, v in net.state_dict().items()}, save_path)   # keys without `module.` (base_model.py:98)

    def load_networks(self, which_epoch):
        for name in self.model_names:
            if isinstance(name, str):
                load_path = os.path.join(self.save_dir, '%s_net_%s.pth' % (which_epoch, name))
                net = getattr(self, 'net' + name)
                print('loading the model from %s' % load_path)
                state_dict = torch.load(load_path, map_location=str(self.device))
                for key in list(state_dict.keys()):          # pre-0.4 InstanceNorm checkpoints (base_model.py:103-111)
                    if key.endswith(('running_mean', 'running_var', 'num_batches_tracked')) and key not in net.state_dict():
                        state_dict.pop(key)
                net.load_state_dict(state_dict)               # strict

    def print_networks(self, verbose):
        print('---------- Networks initialized -------------')
        for name in self.model_names:
            if isinstance(name, str):
                net = getattr(self, 'net' + name)
                num_params = sum(p.numel() for p in net.parameters())
                if verbose:
                    print(net)
                print('[Network %s] Total number of parameters : %.3f M' % (name, num_params / 1e6))
        print('-----------------------------------------------')
