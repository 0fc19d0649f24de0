"""Network geometry of the CFEN-ViT v3 generator, derived from the reference's `opt` flags.

Reference: models/networks_iid_hlgvit_crs_gd4_cfs_v3.py:104-388 (`dec_ipt.__init__`) reads
n_feats, hidden_dim_ratio, patch_size, patch_dim, num_heads, loadSize (options/base_options.py:16,96,
104,110,191-193).  Everything the kernels need (window size, token counts, embedding dims, head
counts, hidden dims per level) is computed once here so host code and tests agree.
"""
from dataclasses import dataclass
from types import SimpleNamespace

BRANCHES = ("r", "s", "d")
# opt.model_G -> variant of the generator family (reference models/model_iid_dehazing.py:84-95)
VARIANTS = {"iid_hlgvit_crs_gd4_cfs_v3": "v3", "iid_hlgvit_crs_gd4_cfs": "cfs", "iid_hlgvit_crs_gd4": "crs", "iid_hlgvit_crs_gd4_cfs_v5": "v5"}
# variants whose three levels run at the image's own resolution (no ds_conv_e01 / us_conv_d01*, n_feats channels in head and tails,
# tail_color shared by R and D): networks_iid_hlgvit_crs_gd4_cfs.py:368 and networks_iid_hlgvit_crs_gd4.py:368 (xf = head(input))
FULL_RES_VARIANTS = ("cfs", "crs")


@dataclass(frozen=True)
class VitGeom:
    """One LViT/GViT instance (reference v3:1062-1135 / 1197-1270)."""
    name: str
    kind: str          # "lvit" | "gvit"
    level: int         # 1..3
    channels: int      # C of the map its tokens are cut from (v5 LViT: the conv_shrink output, a quarter of the level's channels)
    patch: int         # 2 (LViT) or 4 (GViT, after 4x avg-pool)
    seq: int           # tokens per window / per pooled image
    dim: int           # embedding dim D = C * patch^2
    heads: int
    hidden: int        # FFN / mlp_head hidden width
    shrink: int = 1    # v5 LViT (networks_iid_hlgvit_crs_gd4_cfs_v5.py:1101-1106,1139,1190): 1x1 conv_shrink to C/4 channels in front of
                       # the block and conv_extend back to C behind it; `channels`, `dim`, `hidden` are the shrunk sizes

    @property
    def map_channels(self):
        """channels of the level's feature map the instance reads and writes"""
        return self.channels * self.shrink


@dataclass(frozen=True)
class NetConfig:
    n_feats: int = 24
    hidden_dim_ratio: int = 4
    patch_size: int = 32      # LViT window edge in feature-map pixels (opt.patch_size)
    patch_dim: int = 2
    num_heads: int = 4
    load_size: int = 256      # edge of the level-1 feature map xf (opt.loadSize)
    n_colors: int = 3
    variant: str = "v3"       # which sibling generator (opt.model_G, models/model_iid_dehazing.py:84-95), see VARIANTS

    @property
    def image_size(self):
        """v3 runs its three levels on a stride-2 down-sampled map (ds_conv_e01, v3:297-298, 396); `cfs` keeps full resolution
        (networks_iid_hlgvit_crs_gd4_cfs.py:368: xf = head(input)), so there the image IS the level-1 map."""
        return self.load_size if self.variant in FULL_RES_VARIANTS else 2 * self.load_size

    @property
    def head_channels(self):
        """channels of the head CNN / tails: n_feats/2 at twice the resolution (v3:123-127), n_feats at level-1 resolution (cfs :117-121)"""
        return self.n_feats if self.variant in FULL_RES_VARIANTS else self.n_feats // 2

    @property
    def full_res(self):
        return self.variant in FULL_RES_VARIANTS

    def level_channels(self, level):
        return self.n_feats << (level - 1)

    def level_size(self, level):
        return self.load_size >> (level - 1)

    def validate(self):
        # the reference's nested Crop2x2 (v3:403-428, 493-500, 526-529) yields windows of
        # loadSize/8 at every level; LViT.img_dim == opt.patch_size must match (v3:1186 fold).
        if self.load_size != 8 * self.patch_size:
            raise ValueError("loadSize (%d) must equal 8*patch_size (%d): the reference's fixed "
                             "crop nesting requires it" % (self.load_size, 8 * self.patch_size))
        if self.patch_dim != 2:
            raise ValueError("only patch_dim=2 is supported (reference default)")
        if self.n_feats % 8 != 0:
            raise ValueError("n_feats must be a multiple of 8")
        if self.variant == "v5" and (self.n_feats // 4 * self.patch_dim ** 2) % self.num_heads:
            raise ValueError("v5: the shrunk embedding dim (n_feats) must be divisible by num_heads")
        if (self.load_size // 4) % 16 != 0 and self.load_size // 16 < 1:
            raise ValueError("loadSize too small")

    def vit_instances(self):
        """All 24 transformer instances in reference construction order (v3:136-246)."""
        out = []
        p = self.patch_dim
        seq_l = (self.patch_size // p) ** 2

        def lv(name, level):
            c = self.level_channels(level)
            shrink = 4 if self.variant == "v5" else 1        # v5:1086-1097: embedding_dim // 4, hidden_dim // 4, num_channels // 4
            c //= shrink
            d = c * p * p
            return VitGeom(name, "lvit", level, c, p, seq_l, d, self.num_heads << (level - 1),
                           d * self.hidden_dim_ratio, shrink)

        def gv(name, level):
            c = self.level_channels(level)
            gp = 2 * p
            d = c * gp * gp
            img = self.level_size(level) // gp          # v3:196-246 img_dim
            seq = (img // gp) ** 2
            hidden = d * self.hidden_dim_ratio
            if name == "globalvit_encoder_02":
                # v3:200 uses patch_dim (not patch_dim*2) in hidden_dim: reference quirk kept.
                hidden = c * p * p * self.hidden_dim_ratio
            return VitGeom(name, "gvit", level, c, gp, seq, d, self.num_heads << (level - 1), hidden)

        out += [lv("localvit_encoder_0%d" % l, l) for l in (1, 2, 3)]
        for b in BRANCHES:
            out += [lv("localvit_decoder_0%d%s" % (l, b), l) for l in (3, 2, 1)]
        out += [gv("globalvit_encoder_0%d" % l, l) for l in (1, 2, 3)]
        for b in BRANCHES:
            out += [gv("globalvit_decoder_0%d%s" % (l, b), l) for l in (3, 2, 1)]
        return out

    def vit(self, name):
        for g in self.vit_instances():
            if g.name == name:
                return g
        raise KeyError(name)


def config_from_opt(opt):
    """Build a NetConfig from a reference-style `opt` namespace (options/base_options.py)."""
    for flag in ("no_mlp", "pos_every", "no_pos", "no_norm"):
        if getattr(opt, flag, False):
            raise NotImplementedError("--%s is not supported by the HIP path (reference default is off)" % flag)
    if getattr(opt, "num_layers", 1) != 1:
        raise NotImplementedError("num_layers != 1 is not supported")
    if getattr(opt, "dropout_rate", 0) != 0:
        raise NotImplementedError("dropout_rate != 0 is not supported (inference path)")
    model_g = getattr(opt, "model_G", "iid_hlgvit_crs_gd4_cfs_v3")
    if model_g not in VARIANTS:
        raise NotImplementedError("--model_G %s is not built on the HIP path (available: %s)" % (model_g, ", ".join(sorted(VARIANTS))))
    cfg = NetConfig(n_feats=int(opt.n_feats), hidden_dim_ratio=int(opt.hidden_dim_ratio),
                    patch_size=int(opt.patch_size), patch_dim=int(getattr(opt, "patch_dim", 2)),
                    num_heads=int(getattr(opt, "num_heads", 4)), load_size=int(opt.loadSize),
                    n_colors=int(getattr(opt, "n_colors", 3)), variant=VARIANTS[model_g])
    cfg.validate()
    return cfg


def default_opt(**overrides):
    """The subset of reference flags the hot path reads, with BASELINE.json's values."""
    o = dict(n_feats=24, hidden_dim_ratio=4, patch_size=32, patch_dim=2, num_heads=4, num_layers=1,
             num_queries=1, dropout_rate=0, no_mlp=False, pos_every=False, no_pos=False, no_norm=False,
             loadSize=256, rgb_range=255, n_colors=3, init_type="kaiming", gpu_ids=[], model_G="iid_hlgvit_crs_gd4_cfs_v3")
    o.update(overrides)
    return SimpleNamespace(**o)
