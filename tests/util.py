"""Shared helpers for the parity tests (seeded inputs, fixtures, oracle construction)."""
import functools
import json
import os

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def inputs(B, H, W, seed, kind="rand"):
    """Identical to tests/golden/make_golden.py:inputs."""
    g = torch.Generator().manual_seed(seed)
    if kind == "rand":
        return torch.rand(B, 3, H, W, generator=g)
    lo = torch.rand(B, 3, (H + 7) // 8, (W + 7) // 8, generator=g)
    return F.interpolate(lo, size=(H, W), mode="bilinear", align_corners=False).clamp(0, 1)


@functools.lru_cache(maxsize=None)
def tables_npz():
    return dict(np.load(os.path.join(GOLD, "tables.npz")))


@functools.lru_cache(maxsize=None)
def e2e_cases():
    return json.load(open(os.path.join(GOLD, "e2e.json")))


@functools.lru_cache(maxsize=None)
def synth_sd():
    from progressivecodec_amd.synth import synthetic_state_dict
    sd = synthetic_state_dict()
    t = tables_npz()
    # the tables the reference built for these weights (fixture), so that results are comparable with the goldens
    sd["gaussian_conditional._quantized_cdf"] = torch.from_numpy(t["gc_cdf"])
    sd["gaussian_conditional._cdf_length"] = torch.from_numpy(t["gc_len"])
    sd["gaussian_conditional._offset"] = torch.from_numpy(t["gc_off"])
    sd["entropy_bottleneck._quantized_cdf"] = torch.from_numpy(t["eb_cdf"])
    sd["entropy_bottleneck._cdf_length"] = torch.from_numpy(t["eb_len"])
    sd["entropy_bottleneck._offset"] = torch.from_numpy(t["eb_off"])
    return sd


def oracle_codec(backend):
    from oracle.codec_ref import RefCodec
    return RefCodec(synth_sd(), backend)


@functools.lru_cache(maxsize=None)
def gpu_codec():
    from progressivecodec_amd import ChannelProgresssiveWACNN
    net = ChannelProgresssiveWACNN(device="cuda:0")
    net.load_state_dict(synth_sd())
    return net
