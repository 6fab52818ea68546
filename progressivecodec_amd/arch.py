"""Architecture description of the reference's canonical codec.

This is the single place that knows the layer shapes of
``ChannelProgresssiveWACNN`` (reference: src/compress/models/CHProg_cnn.py:30-274,
base class WACNN src/compress/models/cnn.py:23-134) in the configuration the
authors ran (SURVEY.md section 8 preamble):

    N=192, M=640, division_dimension=[320,640], dim_chunk=32, multiple_decoder=True,
    multiple_encoder=False, multiple_hyperprior=True, delta_encode=True,
    joiner_policy="res", support_progressive_slices=5, max_support_slices=5

It yields the reference ``state_dict`` key layout (1019 tensors) so that a reference
checkpoint can be loaded unchanged, and the layer *programs* the native runtime
executes.  No tensors are created here.
"""
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import List, Tuple


@dataclass(frozen=True)
class CodecConfig:
    N: int = 192
    M: int = 640
    division_dimension: Tuple[int, int] = (320, 640)
    dim_chunk: int = 32
    multiple_decoder: bool = True
    multiple_encoder: bool = False
    multiple_hyperprior: bool = True
    delta_encode: bool = True
    joiner_policy: str = "res"
    support_progressive_slices: int = 5
    max_support_slices: int = 5
    mask_policy: str = "two-levels"
    num_heads: int = 8
    scales_min: float = 0.11
    scales_max: float = 256.0
    scales_levels: int = 64

    @property
    def ns0(self):
        return self.division_dimension[0] // self.dim_chunk

    @property
    def ns1(self):
        return self.division_dimension[1] // self.dim_chunk

    def check_supported(self):
        """The native runtime implements the canonical topology, with one encoder (WACNN's 3 -> M g_a, cnn.py:34-44) or with
        multiple_encoder=True (two 3 -> 320 encoders whose outputs are concatenated, CHProg_cnn.py:131-144,691-697)."""
        ok = (self.multiple_decoder and self.multiple_hyperprior
              and self.delta_encode and self.joiner_policy == "res"
              and self.support_progressive_slices == 5 and self.max_support_slices == 5
              and self.dim_chunk == 32 and tuple(self.division_dimension) == (320, 640)
              and self.N == 192 and self.M == 640)
        if not ok:
            raise NotImplementedError(
                "progressivecodec_amd implements the canonical ProgressiveCodec configuration only "
                "(SURVEY.md section 8); got %r" % (self,))


CC_WIDTHS = (224, 176, 128, 64, 32)  # CHProg_cnn.py:167-175


def _gdn(spec, p, C):
    spec[p + ".beta"] = ((C,), "float32", "gdn_beta")
    spec[p + ".gamma"] = ((C, C), "float32", "gdn_gamma")
    spec[p + ".beta_reparam.pedestal"] = ((1,), "float32", "pedestal")
    spec[p + ".beta_reparam.lower_bound.bound"] = ((1,), "float32", "beta_bound")
    spec[p + ".gamma_reparam.pedestal"] = ((1,), "float32", "pedestal")
    spec[p + ".gamma_reparam.lower_bound.bound"] = ((1,), "float32", "gamma_bound")


def _conv(spec, p, cin, cout, k):
    spec[p + ".weight"] = ((cout, cin, k, k), "float32", "conv_w")
    spec[p + ".bias"] = ((cout,), "float32", "conv_b")


def _deconv(spec, p, cin, cout, k):
    spec[p + ".weight"] = ((cin, cout, k, k), "float32", "deconv_w")
    spec[p + ".bias"] = ((cout,), "float32", "conv_b")


def _ru(spec, p, C):  # layers.py:38-57
    _conv(spec, p + ".conv.0", C, C // 2, 1)
    _conv(spec, p + ".conv.2", C // 2, C // 2, 3)
    _conv(spec, p + ".conv.4", C // 2, C, 1)


def _wam(spec, p, C, ws, heads):  # layers.py:31-75, win_attention.py:52-81
    for i in range(3):
        _ru(spec, f"{p}.conv_a.{i}", C)
    a = p + ".conv_b.0.attn"
    spec[a + ".relative_position_bias_table"] = (((2 * ws - 1) ** 2, heads), "float32", "relpos_table")
    spec[a + ".relative_position_index"] = ((ws * ws, ws * ws), "int64", "relpos_index")
    spec[a + ".qkv.weight"] = ((3 * C, C), "float32", "linear_w")
    spec[a + ".qkv.bias"] = ((3 * C,), "float32", "conv_b")
    spec[a + ".proj.weight"] = ((C, C), "float32", "linear_w")
    spec[a + ".proj.bias"] = ((C,), "float32", "conv_b")
    for i in range(1, 4):
        _ru(spec, f"{p}.conv_b.{i}", C)
    _conv(spec, p + ".conv_b.4", C, C, 1)


def _stack5(spec, p, cin):
    c = cin
    for j, w in enumerate(CC_WIDTHS):
        _conv(spec, f"{p}.{2 * j}", c, w, 3)
        c = w


def cc_in_channels(cfg: CodecConfig, family: str, i: int) -> int:
    """Input width of the i-th 5-conv stack of a family (CHProg_cnn.py:167,193,237,264)."""
    d0 = cfg.division_dimension[0]
    delta = cfg.division_dimension[1] - cfg.division_dimension[0]
    est = cfg.support_progressive_slices + 1
    if family in ("cc_mean_transforms", "cc_scale_transforms"):
        return d0 + 32 * min(i, 5)
    if family == "lrp_transforms":
        return d0 + 32 * min(i + 1, 6)
    if family in ("cc_mean_transforms_prog", "cc_scale_transforms_prog"):
        return delta + 32 * min(i + 1, est)
    if family == "lrp_transforms_prog":
        return delta + 32 * min(i + 2, est + 1)
    raise KeyError(family)


def param_spec(cfg: CodecConfig = CodecConfig()) -> "OrderedDict[str, tuple]":
    """name -> (shape, dtype, kind), in the reference's state_dict order."""
    cfg.check_supported()
    N, M, d0, H = cfg.N, cfg.M, cfg.division_dimension[0], cfg.num_heads
    s = OrderedDict()
    # g_a (cnn.py:34-44): conv GDN conv GDN WAM(8,4) conv GDN conv(->M) WAM(4,2); multiple_encoder (CHProg_cnn.py:131-144): a
    # ModuleList of two such nets ending in d0 = 320 channels each (keys g_a.<k>.<layer>...)
    for p, cout in ((("g_a.0", d0), ("g_a.1", d0)) if cfg.multiple_encoder else (("g_a", M),)):
        _conv(s, p + ".0", 3, N, 5); _gdn(s, p + ".1", N)
        _conv(s, p + ".2", N, N, 5); _gdn(s, p + ".3", N)
        _wam(s, p + ".4", N, 8, H)
        _conv(s, p + ".5", N, N, 5); _gdn(s, p + ".6", N)
        _conv(s, p + ".7", N, cout, 5)
        _wam(s, p + ".8", cout, 4, H)
    # g_s[0..1] (CHProg_cnn.py:149-161)
    for k in range(2):
        p = f"g_s.{k}"
        _wam(s, p + ".0", d0, 4, H)
        _deconv(s, p + ".1", d0, N, 5); _gdn(s, p + ".2", N)
        _deconv(s, p + ".3", N, N, 5); _gdn(s, p + ".4", N)
        _wam(s, p + ".5", N, 8, H)
        _deconv(s, p + ".6", N, N, 5); _gdn(s, p + ".7", N)
        _deconv(s, p + ".8", N, 3, 5)
    # h_a (cnn.py:57-67)
    for j, (ci, co) in enumerate(((M, 320), (320, 288), (288, 256), (256, 224), (224, N))):
        _conv(s, f"h_a.{2 * j}", ci, co, 3)
    # h_mean_s / h_scale_s [0..1] (CHProg_cnn.py:208-232)
    def _hs(p):
        _conv(s, p + ".0", N, 192, 3)
        _conv(s, p + ".2.0", 192, 224 * 4, 3)
        _conv(s, p + ".4", 224, 256, 3)
        _conv(s, p + ".6.0", 256, 288 * 4, 3)
        _conv(s, p + ".8", 288, d0, 3)
    # module registration order in the reference: WACNN.__init__ registers h_a, h_mean_s, h_scale_s,
    # cc_mean, cc_scale, lrp, entropy_bottleneck, gaussian_conditional; the subclass re-assigns
    # g_s, cc_*, lrp, h_*_s in place and appends the *_prog families.
    for k in range(2):
        _hs(f"h_mean_s.{k}")
    for k in range(2):
        _hs(f"h_scale_s.{k}")
    for fam in ("cc_mean_transforms", "cc_scale_transforms", "lrp_transforms"):
        for i in range(cfg.ns0):
            _stack5(s, f"{fam}.{i}", cc_in_channels(cfg, fam, i))
    eb = "entropy_bottleneck"
    filters = (1, 3, 3, 3, 3, 1)
    for i in range(5):
        s[f"{eb}._matrix{i}"] = ((N, filters[i + 1], filters[i]), "float32", "eb_matrix")
        s[f"{eb}._bias{i}"] = ((N, filters[i + 1], 1), "float32", "eb_bias")
        if i < 4:
            s[f"{eb}._factor{i}"] = ((N, filters[i + 1], 1), "float32", "eb_factor")
    s[f"{eb}.quantiles"] = ((N, 1, 3), "float32", "eb_quantiles")
    s[f"{eb}._offset"] = ((N,), "int32", "table")
    s[f"{eb}._quantized_cdf"] = ((N, 0), "int32", "table")
    s[f"{eb}._cdf_length"] = ((N,), "int32", "table")
    s[f"{eb}.target"] = ((3,), "float32", "eb_target")
    s[f"{eb}.likelihood_lower_bound.bound"] = ((1,), "float32", "likelihood_bound")
    gc = "gaussian_conditional"
    s[f"{gc}._offset"] = ((cfg.scales_levels,), "int32", "table")
    s[f"{gc}._quantized_cdf"] = ((cfg.scales_levels, 0), "int32", "table")
    s[f"{gc}._cdf_length"] = ((cfg.scales_levels,), "int32", "table")
    s[f"{gc}.scale_table"] = ((cfg.scales_levels,), "float32", "scale_table")
    s[f"{gc}.scale_bound"] = ((1,), "float32", "scale_bound")
    s[f"{gc}.likelihood_lower_bound.bound"] = ((1,), "float32", "likelihood_bound")
    s[f"{gc}.lower_bound_scale.bound"] = ((1,), "float32", "scale_bound")
    for fam in ("cc_mean_transforms_prog", "cc_scale_transforms_prog", "lrp_transforms_prog"):
        for i in range(cfg.ns0):
            _stack5(s, f"{fam}.{i}", cc_in_channels(cfg, fam, i))
    return s


# ----------------------------------------------------------------------------- REM (rate-enhancement module)
REM_SUBNETS = ("enc_base_entropy_params", "enc_enh_entropy_params", "enc_base_rep", "enc")


def rem_subnet_blocks(dimension="big", N=32, mu_std=False):
    """(in, out) channels of the ResidualBlocks of one LatentRateReduction (reference models/CHProgREM.py:12-72).  mu_std=True: the
    enhancement-parameter branch takes cat(mu, scale) (2N channels, :30,46) and the last block of `enc` returns 2N channels (:42,64)."""
    extra = [(N, N)] if dimension == "big" else []
    e_in, e_out = (2 * N, 2 * N) if mu_std else (N, N)
    return {
        "enc_base_entropy_params": [(2 * N, N), (N, N)] + extra,
        "enc_enh_entropy_params": [(e_in, N), (N, N)] + extra,
        "enc_base_rep": [(N, N), (N, N)] + extra,
        "enc": [(3 * N, 2 * N), (2 * N, 2 * N)] + ([(2 * N, 2 * N)] if dimension == "big" else []) + [(2 * N, e_out)],
    }


def rem_param_spec(check_multiple=3, dimension="big", N=32, mu_std=False) -> "OrderedDict[str, tuple]":
    """State-dict layout of PostRateProcessedNetwork.post_latent (CHProgREM.py:227-234): [check level][slice] LatentRateReduction,
    each ResidualBlock = conv1 3x3, conv2 3x3 and, when in != out, a 1x1 skip (models/utils.py:59-87).  name -> (shape, dtype, kind)."""
    s = OrderedDict()
    blocks = rem_subnet_blocks(dimension, N, mu_std)
    for k in range(check_multiple):
        for i in range(10):
            for sub in REM_SUBNETS:
                for j, (ci, co) in enumerate(blocks[sub]):
                    p = f"{k}.{i}.{sub}.{j}"
                    _conv(s, p + ".conv1", ci, co, 3)
                    _conv(s, p + ".conv2", co, co, 3)
                    if ci != co:
                        _conv(s, p + ".skip", ci, co, 1)
    return s
