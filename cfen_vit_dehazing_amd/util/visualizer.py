"""save_images of the reference (util/visualizer.py:10-27): one PNG per (image, visual) named
`<stem>_<label>.png` under the web page's image directory."""
import ntpath
import os

from . import util


def save_images(image_dir, visuals, image_path, aspect_ratio=1.0, width=256, multi_flag=False):
    for i in range(len(image_path)):
        short_path = ntpath.basename(image_path[i])
        name = os.path.splitext(short_path)[0]
        for label, im_data in visuals.items():
            im = util.tensor2im(im_data[i, :, :, :])
            save_path = os.path.join(image_dir, '%s_%s.png' % (name, label))
            util.save_image(im, save_path)
