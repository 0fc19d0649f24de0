"""Module-level names of the reference's models/networks_iid_hlgvit_crs_gd4_cfs.py (--model_G iid_hlgvit_crs_gd4_cfs): the sibling
generator with a full-resolution head and no ds_conv_e01 / us_conv_d01* stage.  Same HIP kernels, another launch plan
(csrc/cfen_net.cpp, variant 1); the variant is taken from opt.model_G by config.config_from_opt."""
from ..hipnet import dec_ipt, define_G, init_weights  # noqa: F401
