"""Image helpers of the harness (reference util/util.py: tensor2im :12-24, save_image :51-53, mkdirs :66-76)."""
import os

import numpy as np
import torch
from PIL import Image


def tensor2im(input_image, imtype=np.uint8):
    """(C,H,W) tensor in [-1,1] -> (H,W,3) uint8 the way the reference does it: (x + 1) / 2 * 255, then astype -- truncation toward
    zero, no clamp, no rounding -- and a 1-channel tensor tiled to 3 (util/util.py:12-24; pinned by tests/golden/harness.npz).
    Anything that is not a tensor is handed back as it is."""
    if not isinstance(input_image, torch.Tensor):
        return input_image
    t = input_image.data
    if t.dtype == torch.uint8 and t.dim() == 3 and t.shape[-1] == 3 and imtype == np.uint8:
        return t.cpu().numpy()      # already an image: the generator wrote tensor2im's bytes itself (hipnet.dec_ipt.output_u8)
    if t.is_cuda and t.dim() == 3 and imtype == np.uint8:
        # the same arithmetic on the device (csrc/k_tokens.hip: k_tensor2im_u8): H*W*3 bytes cross PCIe instead of fp32 planes
        from .. import ops
        return ops.tensor2im_u8(t.float().contiguous()).cpu().numpy()
    a = t.cpu().float().numpy()
    if a.shape[0] == 1:
        a = np.tile(a, (3, 1, 1))
    return ((np.transpose(a, (1, 2, 0)) + 1) / 2.0 * 255.0).astype(imtype)


# zlib level of the PNG writer (`--png_compress_level`, an extension: the reference writes PIL's default, 6).  The default leaves every file byte for byte as the
# reference's util.save_image (util/util.py:47-49 there) writes it; level 1 encodes ~3x faster into ~25 % larger files of the SAME pixels (file -> file inference on a
# host whose CPU share is small is bound by this encode: profiles/r06_cli_throughput.json).  None = PIL's default.
PNG_COMPRESS_LEVEL = None


def save_image(image_numpy, image_path):
    if PNG_COMPRESS_LEVEL is None:
        Image.fromarray(image_numpy).save(image_path)
    else:
        Image.fromarray(image_numpy).save(image_path, compress_level=int(PNG_COMPRESS_LEVEL))


def mkdirs(paths):
    for path in ([paths] if isinstance(paths, str) else paths):
        os.makedirs(path, exist_ok=True)


def mkdir(path):
    os.makedirs(path, exist_ok=True)
