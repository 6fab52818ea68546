"""Host-side mirror of the reference's REM model, ``PostRateProcessedNetwork``
(/root/reference/src/compress/models/CHProgREM.py:205): a frozen ChannelProgresssiveWACNN whose predicted scale of every enhancement
slice is refined by a small CNN (``LatentRateReduction``, :12-86) before the variance mask is taken (``apply_latent_enhancement``,
:375-428) -- three sets of ten CNNs, one set per range of quality between the ``check_levels``.

Same constructor keywords (``check_levels``, ``mu_std``, ``dimension``, ``escalation``), ``load_state_dict(state_dict_base,
state_dict_post)``, ``compress`` / ``decompress`` signatures and return dictionaries as the reference (``real_compress=True``).
``mu_std=True``: the nets take cat(mu, scale) and refine the predicted mean as well (:30,42,397-416).  ``dimension="middle"``: two
ResidualBlocks per sub-net instead of three (:23-43).  ``checkpoint_rep``: a [B,320,h,w] representation that replaces the decoded base
slices as the nets' x_base input (:773,989); ``extract_chekpoint_representation_from_images`` (sic, :335-373) chains the check levels
through it when ``escalation=True``.  All arithmetic runs in the native codec (libpcodec.so: ``pc_codec_set_rem``,
``pc_codec_set_rem_checkpoint``); nothing here computes.
"""
import ctypes as C
from collections import OrderedDict

import numpy as np

from ._lib import check, lib
from .arch import rem_param_spec
from .model import ChannelProgresssiveWACNN, _module_base


class PostRateProcessedNetwork(_module_base()):
    """A ``torch.nn.Module`` like the reference's (CHProgREM.py:205): ``base_net`` is a registered sub-module, ``state_dict()`` carries
    ``base_net.*`` and ``post_latent.*`` as the reference's checkpoints do; the tensors are host copies of what was loaded."""

    def __init__(self, base_net, check_levels=(0.01, 0.25, 1.75), mu_std=False, dimension="big", escalation=False):
        super().__init__()
        if not isinstance(base_net, ChannelProgresssiveWACNN):
            raise AssertionError("base_net must be a ChannelProgresssiveWACNN")               # CHProgREM.py:224
        if not 1 <= len(check_levels) <= 3:
            raise ValueError("one to three check levels")
        self.base_net = base_net
        self.check_levels = [float(v) for v in check_levels]
        self.check_multiple = len(self.check_levels)
        self.mu_std, self.dimension, self.escalation = mu_std, dimension, escalation
        self._post = None

    # ------------------------------------------------------------------ nn.Module surface
    def to(self, *args, **kwargs):
        return self                                                         # bound to base_net's HIP device

    def forward(self, *args, **kwargs):
        raise NotImplementedError("training forward() (CHProgREM.py:430-670) is out of scope")

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        import torch
        out = OrderedDict() if destination is None else destination
        self.base_net.state_dict(destination=out, prefix=prefix + "base_net.")
        for k, a in (self._post or {}).items():
            out[prefix + "post_latent." + k] = torch.from_numpy(np.array(a, copy=True))
        return out

    def named_parameters(self, prefix="", recurse=True, remove_duplicate=True):
        import torch
        yield from self.base_net.named_parameters(prefix=prefix + "base_net.")
        for k, a in (self._post or {}).items():
            yield prefix + "post_latent." + k, torch.nn.Parameter(torch.from_numpy(np.array(a, copy=True)), requires_grad=False)

    def parameters(self, recurse=True):
        for _, p in self.named_parameters():
            yield p

    def load_state_dict(self, state_dict_base, state_dict_post=None, strict=False):
        """CHProgREM.py:361-369: the base codec's state dict and, optionally, post_latent's."""
        extra = None
        if state_dict_post is not None:
            spec = rem_param_spec(self.check_multiple, self.dimension, mu_std=self.mu_std)
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

    def _on(self, checkpoint_rep, B, h, w):
        """Switch the refinement on for the next base_net call of a [B, ., 16h, 16w] batch.  The checkpoint pointer is ALWAYS (re)set --
        NULL without a checkpoint_rep -- so a pointer left behind by a call that failed before the native side consumed it can never
        reach the next one; its shape is checked against the call's own, as the reference's torch.cat would (CHProgREM.py:773,989):
        the native nets index it with the call's B and h*w."""
        if self._post is None:
            raise ValueError("load_state_dict(state_dict_base, state_dict_post) first: the REM needs its post_latent weights")
        rep = None
        if checkpoint_rep is not None:
            import torch
            if checkpoint_rep.dim() != 4 or tuple(checkpoint_rep.shape) != (B, 320, h, w):
                raise ValueError(f"checkpoint_rep must be [B, 320, H/16, W/16] = {(B, 320, h, w)}, got {tuple(checkpoint_rep.shape)}")
            rep = checkpoint_rep.to(self.base_net.device, torch.float32).contiguous()
        lv = (C.c_double * self.check_multiple)(*self.check_levels)
        check(lib().pc_codec_set_rem(self.base_net._h, lv, self.check_multiple), "pc_codec_set_rem")
        self._rep = rep                                                     # kept alive until the call has consumed it
        check(lib().pc_codec_set_rem_checkpoint(self.base_net._h, C.c_void_p(rep.data_ptr()) if rep is not None else None),
              "pc_codec_set_rem_checkpoint")

    def extract_chekpoint_representation_from_images(self, x, quality, rc=True):
        """CHProgREM.py:335-373 (name as in the reference): the y_hat a coder of `quality` produced -- with escalation=True, chained
        through the check levels below it, each level's nets reading the representation of the level before."""
        if not self.escalation:
            return self.compress(x, quality=quality, mask_pol="point-based-std", real_compress=rc)["y_hat"]
        rep = self.compress(x, quality=self.check_levels[0], mask_pol="point-based-std", real_compress=rc)["y_hat"]
        if quality == self.check_levels[0]:
            return rep
        rep1 = self.compress(x, quality=self.check_levels[1], mask_pol="point-based-std", checkpoint_rep=rep, real_compress=rc)["y_hat"]
        if quality == self.check_levels[1]:
            return rep1
        return self.compress(x, quality=self.check_levels[2], mask_pol="point-based-std", checkpoint_rep=rep1, real_compress=rc)["y_hat"]

    def _off(self):
        check(lib().pc_codec_set_rem_checkpoint(self.base_net._h, None), "pc_codec_set_rem_checkpoint")
        self._rep = None
        check(lib().pc_codec_set_rem(self.base_net._h, None, 0), "pc_codec_set_rem")

    def compress(self, x, quality=0.0, mask_pol="point-based-std", checkpoint_rep=None, real_compress=True, used_qual=None):
        """CHProgREM.py:673-888 -> {"strings", "shape", "masks", "y_hat"}."""
        if not real_compress:
            raise NotImplementedError("real_compress=False (the training-time quantiser without entropy coding) is out of scope")
        if x.dim() != 4:
            raise ValueError("Invalid `inputs` size. Expected a [B,3,H,W] tensor.")
        self._on(checkpoint_rep, x.shape[0], x.shape[2] // 16, x.shape[3] // 16)
        try:
            out = self.base_net.compress(x, quality, mask_pol)
            out["y_hat"] = self.base_net.read_latent("yhat_enh" if quality > 0 else "yhat_base", x.shape[0], x.shape[2] // 16, x.shape[3] // 16)
        finally:
            self._off()
        return out

    def decompress(self, strings, shape, quality, mask_pol=None, checkpoint_rep=None, timing=False, used_qual=None):
        """CHProgREM.py:896-1126 -> {"x_hat", "y_hat", "time"}."""
        import time
        if not isinstance(strings, (tuple, list)) or len(strings) != 2:
            raise ValueError("Invalid `strings` parameter type.")
        self._on(checkpoint_rep, len(strings[1]), 4 * int(shape[0]), 4 * int(shape[1]))
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
