"""Host-side mirror of the reference's REM model, ``PostRateProcessedNetwork``
(/root/reference/src/compress/models/CHProgREM.py:205): a frozen ChannelProgresssiveWACNN whose predicted scale of every enhancement
slice is refined by a small CNN (``LatentRateReduction``, :12-86) before the variance mask is taken (``apply_latent_enhancement``,
:375-428) -- three sets of ten CNNs, one set per range of quality between the ``check_levels``.

Same constructor keywords, ``load_state_dict(state_dict_base, state_dict_post)``, ``compress`` / ``decompress`` signatures and return
dictionaries as the reference (``real_compress=True``, ``checkpoint_rep=None``, ``mu_std=False``: what its compress_with_ac-style
evaluation uses).  All arithmetic runs in the native codec (libpcodec.so: ``pc_codec_set_rem``); nothing here computes.
"""
import ctypes as C

import numpy as np

from ._lib import check, lib
from .arch import rem_param_spec
from .model import ChannelProgresssiveWACNN


class PostRateProcessedNetwork:
    def __init__(self, base_net, check_levels=(0.01, 0.25, 1.75), mu_std=False, dimension="big", escalation=False):
        if not isinstance(base_net, ChannelProgresssiveWACNN):
            raise AssertionError("base_net must be a ChannelProgresssiveWACNN")               # CHProgREM.py:224
        if mu_std or escalation:
            raise NotImplementedError("mu_std / escalation variants of the REM are not implemented (SURVEY.md section 8f)")
        if not 1 <= len(check_levels) <= 3:
            raise ValueError("one to three check levels")
        self.base_net = base_net
        self.check_levels = [float(v) for v in check_levels]
        self.check_multiple = len(self.check_levels)
        self.mu_std, self.dimension, self.escalation = mu_std, dimension, escalation
        self._post = None

    def eval(self):
        return self

    def load_state_dict(self, state_dict_base, state_dict_post=None, strict=False):
        """CHProgREM.py:361-369: the base codec's state dict and, optionally, post_latent's."""
        extra = None
        if state_dict_post is not None:
            spec = rem_param_spec(self.check_multiple, self.dimension)
            missing = [k for k in spec if k not in state_dict_post]
            if missing:
                # (the native REM needs every post_latent tensor whatever `strict` says: a CNN with absent weights cannot run; the
                # reference's strict=False would leave them at their random initialisation)
                raise RuntimeError(f"post_latent state dict lacks {missing[:4]}...")
            unexpected = [k for k in state_dict_post if k not in spec]
            if strict and unexpected:
                raise RuntimeError(f"Error(s) in loading state_dict for post_latent: unexpected {unexpected[:4]}...")
            extra = {}
            for k, (shape, dtype, _) in spec.items():
                v = state_dict_post[k]
                a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
                if tuple(a.shape) != tuple(shape):
                    raise RuntimeError(f"size mismatch for post_latent.{k}: {tuple(a.shape)} vs {tuple(shape)}")
                extra["post_latent." + k] = a
            self._post = {k[len("post_latent."):]: v for k, v in extra.items()}
        self.base_net.load_state_dict(state_dict_base, strict=strict, _extra=extra)   # honours `strict` (reference default False, :361)
        return self

    def update(self, *a, **kw):
        return self.base_net.update(*a, **kw)

    def _on(self):
        if self._post is None:
            raise ValueError("load_state_dict(state_dict_base, state_dict_post) first: the REM needs its post_latent weights")
        lv = (C.c_double * self.check_multiple)(*self.check_levels)
        check(lib().pc_codec_set_rem(self.base_net._h, lv, self.check_multiple), "pc_codec_set_rem")

    def _off(self):
        check(lib().pc_codec_set_rem(self.base_net._h, None, 0), "pc_codec_set_rem")

    def compress(self, x, quality=0.0, mask_pol="point-based-std", checkpoint_rep=None, real_compress=True, used_qual=None):
        """CHProgREM.py:673-888 -> {"strings", "shape", "masks", "y_hat"}."""
        if checkpoint_rep is not None or not real_compress:
            raise NotImplementedError("checkpoint_rep / real_compress=False (training-time representations) are out of scope")
        self._on()
        try:
            out = self.base_net.compress(x, quality, mask_pol)
            out["y_hat"] = self.base_net.read_latent("yhat_enh" if quality > 0 else "yhat_base", x.shape[0], x.shape[2] // 16, x.shape[3] // 16)
        finally:
            self._off()
        return out

    def decompress(self, strings, shape, quality, mask_pol=None, checkpoint_rep=None, timing=False, used_qual=None):
        """CHProgREM.py:896-1126 -> {"x_hat", "y_hat", "time"}."""
        import time
        if checkpoint_rep is not None:
            raise NotImplementedError("checkpoint_rep is out of scope")
        self._on()
        try:
            t0 = time.time()
            out = self.base_net.decompress(strings, shape, quality, mask_pol)
            B = len(strings[1])
            out["y_hat"] = self.base_net.read_latent("yhat_enh" if quality != 0 else "yhat_base", B, 4 * int(shape[0]), 4 * int(shape[1]))
            if timing:
                import torch
                torch.cuda.synchronize(self.base_net.device)
            out["time"] = time.time() - t0 if timing else 0
        finally:
            self._off()
        return out
