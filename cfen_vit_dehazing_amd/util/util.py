"""util/util.py of the reference (tensor2im :12-24, save_image :51-53, mkdirs :66-76)."""
import os

import numpy as np
import torch
from PIL import Image


def tensor2im(input_image, imtype=np.uint8):
    """(C,H,W) tensor in [-1,1] -> HWC uint8: (x+1)/2*255 then astype = truncation toward zero, no clamp,
    no rounding; 1-channel tensors are tiled to 3 (util/util.py:12-24)."""
    if isinstance(input_image, torch.Tensor):
        image_tensor = input_image.data
    else:
        return input_image
    if image_tensor.is_cuda and image_tensor.dim() == 3 and imtype == np.uint8:
        # same arithmetic on the device (csrc/k_tokens.hip: k_tensor2im_u8): only H*W*3 bytes cross PCIe instead of fp32 planes
        from .. import ops
        return ops.tensor2im_u8(image_tensor.float().contiguous()).cpu().numpy()
    image_numpy = image_tensor.cpu().float().numpy()
    if image_numpy.shape[0] == 1:
        image_numpy = np.tile(image_numpy, (3, 1, 1))
    image_numpy = (np.transpose(image_numpy, (1, 2, 0)) + 1) / 2.0 * 255.0
    return image_numpy.astype(imtype)


def save_image(image_numpy, image_path):
    Image.fromarray(image_numpy).save(image_path)


def mkdirs(paths):
    if isinstance(paths, list) and not isinstance(paths, str):
        for path in paths:
            mkdir(path)
    else:
        mkdir(paths)


def mkdir(path):
    if not os.path.exists(path):
        os.makedirs(path)
