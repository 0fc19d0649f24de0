"""Module-level names of the reference's models/networks_iid_hlgvit_crs_gd4_cfs_v3.py that callers use."""
from ..hipnet import dec_ipt, define_G, init_weights  # noqa: F401
