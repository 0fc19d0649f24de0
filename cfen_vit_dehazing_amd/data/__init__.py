"""Folder dataset + loader of the inference harness: data/__init__.py:10-60, data/dec_vit_data.py:11-120,
data/base_dataset.py:20-47, data/image_folder.py:37-47 of the reference (test-time subset: `dataroot/hazy/*`).
torchvision is not a dependency: ToTensor + Normalize(0.5, 0.5) are written out (HWC uint8 -> CHW float in
[-1, 1]); images are fed at native size because the reference default --resize_or_crop 'resize' matches no
transform branch (base_dataset.py:20-47)."""
import os
import random

import numpy as np
import torch
import torch.utils.data
from PIL import Image

IMG_EXTENSIONS = ['.jpg', '.JPG', '.jpeg', '.JPEG', '.png', '.PNG', '.ppm', '.PPM', '.bmp', '.BMP']


def is_image_file(filename):
    return any(filename.endswith(extension) for extension in IMG_EXTENSIONS)


def make_dataset(dir):
    images = []
    assert os.path.isdir(dir), '%s is not a valid directory' % dir
    for root, _, fnames in sorted(os.walk(dir)):
        for fname in fnames:
            if is_image_file(fname):
                images.append(os.path.join(root, fname))
    return list(set(images))


def to_normalized_tensor(img):
    """transforms.ToTensor() + Normalize((.5,.5,.5),(.5,.5,.5)) (base_dataset.py:44-46)."""
    a = np.asarray(img, dtype=np.uint8)
    t = torch.from_numpy(a.copy()).permute(2, 0, 1).float().div(255.0)
    return (t - 0.5) / 0.5


def to_u8_hwc(img):
    """--u8_input: the decoded image as it is, (H,W,3) uint8; ToTensor + Normalize happen on the device (csrc/k_tokens.hip: k_u8hwc_to_nhwc)."""
    return torch.from_numpy(np.asarray(img, dtype=np.uint8).copy())


class _ResizeShortSide:
    """--resize_or_crop resize_only / scale_width: the short side becomes --loadSize (a picklable callable: the loader's workers are forked by
    default, early -- CustomDatasetDataLoader.start_workers -- but CFEN_LOADER_CONTEXT=forkserver|spawn must be able to pickle the transform)"""

    def __init__(self, load_size):
        self.load_size = load_size

    def __call__(self, img):
        w, h = img.size
        s = self.load_size / min(w, h)
        return to_normalized_tensor(img.resize((max(1, round(w * s)), max(1, round(h * s))), Image.BICUBIC))


def get_transform(opt):
    mode = opt.resize_or_crop
    if mode in ('resize', 'none'):
        return to_u8_hwc if getattr(opt, 'u8_input', False) else to_normalized_tensor
    if mode in ('resize_only', 'scale_width'):
        return _ResizeShortSide(opt.loadSize)
    raise NotImplementedError("--resize_or_crop %s uses random crops (training only)" % mode)


class DECVITDATA(torch.utils.data.Dataset):
    def initialize(self, opt):
        self.opt = opt
        self.root = opt.dataroot
        self.dir_B = os.path.join(opt.dataroot, 'hazy')
        self.B_paths = sorted(make_dataset(self.dir_B))
        world, rank = getattr(opt, 'dist_world', 1), getattr(opt, 'dist_rank', 0)
        if world > 1:
            # one process per GPU (test.py under torch.distributed.run): this rank's contiguous slice of the images the run covers
            from ..parallel import shard_items
            if not opt.sb:
                raise ValueError("multi-process inference needs --sb (the reference samples images randomly without it, dec_vit_data.py:51-58)")
            # (both bounds may be inf -- the default max_dataset_size and options without how_many: int(inf) would raise OverflowError)
            import math
            bound = min(opt.max_dataset_size, getattr(opt, 'how_many', float('inf')) * opt.batchSize)
            limit = len(self.B_paths) if not math.isfinite(bound) else min(len(self.B_paths), int(bound))
            self.B_paths = shard_items(self.B_paths[:limit], world, rank)
        self.B_size = len(self.B_paths)
        self.transform = get_transform(opt)

    def __getitem__(self, index):
        if self.opt.sb:
            B_path = self.B_paths[index % self.B_size]
        else:                                   # the reference samples randomly unless --sb (dec_vit_data.py:51-58)
            B_path = self.B_paths[random.randint(0, self.B_size - 1)]
        B = self.transform(Image.open(B_path).convert('RGB'))
        gray = (self.opt.output_nc == 1 and self.opt.which_direction != 'BtoA') or (self.opt.input_nc == 1 and self.opt.which_direction == 'BtoA')
        if gray and B.dtype == torch.uint8:
            raise ValueError("--u8_input hands over the decoded RGB image; single-channel input (--input_nc / --output_nc 1) needs the float path")
        if gray:
            B = (B[0, ...] * 0.299 + B[1, ...] * 0.587 + B[2, ...] * 0.114).unsqueeze(0)
        return {'B': B, 'B_paths': B_path}

    def __len__(self):
        return self.B_size

    def name(self):
        return 'DEC_ViT'


def CreateDataset(opt):
    if opt.dataset_mode == 'dec_vit':
        dataset = DECVITDATA()
    else:
        raise ValueError("Dataset [%s] not recognized." % opt.dataset_mode)
    print("dataset [%s] was created" % (dataset.name()))
    dataset.initialize(opt)
    return dataset


class CustomDatasetDataLoader():
    def name(self):
        return 'CustomDatasetDataLoader'

    def initialize(self, opt):
        self.opt = opt
        self.dataset = CreateDataset(opt)
        workers = int(opt.nThreads)
        # How the worker processes come to be matters on MI355X / ROCm 7.2: forking a process that already holds its GPU working set (packed weights,
        # workspaces, pinned buffers) stalled the GPU for 10-16 s at the first launch afterwards (the fork write-protects the parent's pinned /
        # registered pages and the driver re-validates the process's buffers): test.py --nThreads 4 ran at 6.5 images/s against 35 with --nThreads 0,
        # and the first batch of a pipelined run waited 14 s (profiles/r05_cli_throughput.json).  So the harness forks EARLY: test.py calls
        # start_workers() right after the options are parsed, before the model exists (the reference creates the loader first too, test.py:24-26, but
        # its workers only start at the first iteration).  CFEN_LOADER_CONTEXT=forkserver|spawn starts fresh worker processes instead (the main
        # script must then be importable: `if __name__ == '__main__'`).
        ctx = os.environ.get('CFEN_LOADER_CONTEXT', 'fork') if workers > 0 else None
        self.dataloader = torch.utils.data.DataLoader(self.dataset, batch_size=opt.batchSize, shuffle=not opt.sb,
                                                      num_workers=workers, multiprocessing_context=ctx,
                                                      pin_memory=bool(workers > 0 and getattr(opt, 'in_flight', 1) > 1 and torch.cuda.is_available()))
        # (pin_memory: the pipelined driver copies every batch H2D asynchronously; torch's pin thread stages it off the main thread)
        self._started = None

    def start_workers(self):
        """fork the worker processes NOW (they begin decoding the first batches); the next iteration over the loader uses them"""
        if int(self.opt.nThreads) > 0 and self._started is None:
            self._started = iter(self.dataloader)
        return self

    def load_data(self):
        return self

    def __len__(self):
        return min(len(self.dataset), self.opt.max_dataset_size)

    def __iter__(self):
        it, self._started = (self._started, None) if getattr(self, '_started', None) is not None else (iter(self.dataloader), None)
        for i, data in enumerate(it):
            if i * self.opt.batchSize >= self.opt.max_dataset_size:
                break
            yield data


def CreateDataLoader(opt):
    data_loader = CustomDatasetDataLoader()
    print(data_loader.name())
    data_loader.initialize(opt)
    return data_loader
