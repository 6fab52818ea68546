"""Deterministic synthetic weights in the reference's state_dict layout.

No trained checkpoint ships with the reference (its only checkpoint is a dangling
symlink) and there is no network, so tests and bench.py use
weights drawn from a per-tensor seeded numpy generator (PCG64; identical on every
machine).  Plain fan-in-scaled initialisation yields degenerate latents (every
predicted scale clamps to the 0.11 floor), so a few tensors are
amplified to give the entropy-coding stage realistic work: symbols of several
units, predicted scales spread over the 64-entry table, ragged per-channel
hyper-prior CDFs, non-trivial GDN and window-attention parameters.

The CDF tables (``*._quantized_cdf/_cdf_length/_offset``) are *not* produced here:
they are built by ``ChannelProgresssiveWACNN.update()`` exactly as the reference
does (cnn.py:137-142) or taken from a checkpoint.
"""
import math
import zlib
from collections import OrderedDict

import numpy as np

from .arch import CodecConfig, param_spec

_PED = float((2.0 ** -18) ** 2)  # parametrizers.py:27-30
# amplification recipe (tuned through the real reference: index histogram 0..27, bypass rate ~0.3 %)
AMP_Y, AMP_Z, AMP_S, BIAS_S = 0.8, 4.0, 2.0, 0.6
AMP_HS, AMP_M = 0.15, 1.0


def _rng(name: str, seed: int):
    return np.random.default_rng([zlib.crc32(name.encode()), seed])


def relative_position_index(ws: int) -> np.ndarray:
    """win_attention.py:64-74: pair-wise relative position index inside a ws x ws window."""
    ch, cw = np.meshgrid(np.arange(ws), np.arange(ws), indexing="ij")
    coords = np.stack([ch.ravel(), cw.ravel()])            # 2, T
    rel = coords[:, :, None] - coords[:, None, :]          # 2, T, T
    rel = rel.transpose(1, 2, 0).copy()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1).astype(np.int64)


def scale_table(cfg: CodecConfig = CodecConfig()) -> np.ndarray:
    """cnn.py:14-20 (exp(linspace(log min, log max, levels))), evaluated in float32."""
    import torch
    return torch.exp(torch.linspace(math.log(cfg.scales_min), math.log(cfg.scales_max),
                                    cfg.scales_levels)).numpy().copy()


def synthetic_state_dict(cfg: CodecConfig = CodecConfig(), seed: int = 0, as_torch: bool = True):
    spec = param_spec(cfg)
    out = OrderedDict()
    eb_scale = 10.0 ** (1.0 / 5.0)  # entropy_models.py:325 (init_scale=10, 4 filters)
    filters = (1, 3, 3, 3, 3, 1)
    for name, (shape, dtype, kind) in spec.items():
        g = _rng(name, seed)
        if kind == "conv_w":
            co, ci, kh, kw = shape
            std = math.sqrt(2.0 / (ci * kh * kw))
            if kh == 5:
                std *= 0.8
            w = g.standard_normal(shape) * std
            if name in ("g_a.7.weight", "g_a.0.7.weight", "g_a.1.7.weight"):
                w *= AMP_Y       # latent amplitude: symbols of a few units
            if name.startswith("h_a.8"):
                w *= AMP_Z       # hyper-latent z spread over a few integers
            if name.endswith(".8.weight") and name.startswith("cc_scale_transforms"):
                w *= AMP_S       # spread the predicted scales over the table
            if name.endswith(".8.weight") and name.startswith("cc_mean_transforms"):
                w *= AMP_M       # keep the predicted means below the latent amplitude
            if name.endswith(".8.weight") and name.startswith(("h_mean_s", "h_scale_s")):
                w *= AMP_HS      # hyper-synthesis output amplitude (interior-pixel limit)
            v = w
        elif kind == "deconv_w":
            ci, co, kh, kw = shape
            v = g.standard_normal(shape) * math.sqrt(2.0 / (ci * kh * kw / 4.0)) * 0.7
        elif kind == "linear_w":
            v = g.standard_normal(shape) * math.sqrt(1.0 / shape[1])
        elif kind == "conv_b":
            v = g.standard_normal(shape) * 0.05
            if name.endswith(".8.bias") and name.startswith("cc_scale_transforms"):
                v = v + BIAS_S
            if name == "g_s.0.8.bias" or name == "g_s.1.8.bias":
                v = v + 0.45
        elif kind == "gdn_beta":
            beta = 1.0 + 0.5 * g.random(shape)
            v = np.sqrt(np.maximum(beta + _PED, _PED))
        elif kind == "gdn_gamma":
            C = shape[0]
            gamma = 0.1 * np.eye(C) + 0.004 * np.abs(g.standard_normal(shape))
            v = np.sqrt(np.maximum(gamma + _PED, _PED))
        elif kind == "pedestal":
            v = np.full(shape, _PED)
        elif kind == "beta_bound":
            v = np.full(shape, (1e-6 + _PED) ** 0.5)
        elif kind == "gamma_bound":
            v = np.full(shape, (0.0 + _PED) ** 0.5)
        elif kind == "relpos_table":
            v = np.clip(g.standard_normal(shape) * 0.4, -1.0, 1.0)
        elif kind == "relpos_index":
            v = relative_position_index(int(round(math.sqrt(shape[0]))))
        elif kind == "eb_matrix":
            i = int(name[-1])
            init = math.log(math.expm1(1.0 / eb_scale / filters[i + 1]))
            v = init + 0.1 * g.standard_normal(shape)
        elif kind == "eb_bias":
            v = g.uniform(-0.5, 0.5, shape)
        elif kind == "eb_factor":
            v = 0.1 * g.standard_normal(shape)
        elif kind == "eb_quantiles":
            med = 0.6 * g.standard_normal((shape[0], 1))
            lo = med - g.uniform(4.0, 11.0, (shape[0], 1))
            hi = med + g.uniform(4.0, 11.0, (shape[0], 1))
            v = np.concatenate([lo, med, hi], axis=1).reshape(shape)
        elif kind == "eb_target":
            t = math.log(2.0 / 1e-9 - 1.0)
            v = np.array([-t, 0.0, t])
        elif kind == "likelihood_bound":
            v = np.full(shape, 1e-9)
        elif kind == "scale_bound":
            v = np.full(shape, cfg.scales_min)
        elif kind == "scale_table":
            v = scale_table(cfg)
        elif kind == "table":
            v = np.zeros([d for d in shape], dtype=np.int32)
        else:
            raise KeyError(kind)
        out[name] = np.ascontiguousarray(np.asarray(v).astype(dtype))
    if as_torch:
        import torch
        return OrderedDict((k, torch.from_numpy(v)) for k, v in out.items())
    return out


def synthetic_post_state_dict(check_multiple: int = 3, dimension: str = "big", seed: int = 0, as_torch: bool = True, mu_std: bool = False):
    """Deterministic weights for PostRateProcessedNetwork.post_latent (reference models/CHProgREM.py:227-234) in its own state-dict
    layout ("<level>.<slice>.<subnet>.<block>.conv1.weight" ...).  Fan-in scaled; the last block of `enc` is damped so that the
    refinement moves the predicted scale by a fraction of itself (an untrained net would otherwise swamp it)."""
    from .arch import rem_param_spec
    out = OrderedDict()
    for name, (shape, dtype, kind) in rem_param_spec(check_multiple, dimension, mu_std=mu_std).items():
        g = _rng("post_latent." + name, seed)
        if kind == "conv_w":
            co, ci, kh, kw = shape
            v = g.standard_normal(shape) * math.sqrt(1.0 / (ci * kh * kw))
            if ".enc.3." in name or (dimension != "big" and ".enc.2." in name):
                v = v * 0.5
        else:
            v = g.standard_normal(shape) * 0.02
        out[name] = np.ascontiguousarray(np.asarray(v).astype(dtype))
    if as_torch:
        import torch
        return OrderedDict((k, torch.from_numpy(v)) for k, v in out.items())
    return out
