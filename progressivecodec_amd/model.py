"""Host-side mirror of the reference's model class for the compress()/decompress() path.

``ChannelProgresssiveWACNN`` here has the same constructor keywords, method names, argument
meaning, return structure and error behaviour as the reference class
(/root/reference/src/compress/models/CHProg_cnn.py:30, :686, :849; base class models/cnn.py:23,
:137, :195), but owns no torch modules: weights live in HBM inside the native codec object
(libpcodec.so, include/pcodec.h) and both entry points are native HIP launch sequences.
PyTorch is used only to hold the input / output device tensors and to supply the HIP stream.
"""
import ctypes as C
import math
from collections import OrderedDict

import numpy as np

from . import entropy
from ._lib import check, lib
from .arch import CodecConfig, param_spec

_MASK_POL = {"point-based-std": 0, "two-levels": 1, "three-levels-std": 2}
_PARAM_KINDS = {"conv_w", "conv_b", "deconv_w", "linear_w", "gdn_beta", "gdn_gamma", "relpos_table", "eb_matrix", "eb_bias", "eb_factor",
                "eb_quantiles"}          # arch.param_spec kinds that are nn.Parameters in the reference (the rest are buffers)
_DT = {"float32": 0, "int32": 1, "int64": 2}


def get_scale_table(min=0.11, max=256, levels=64):
    """models/cnn.py:14-20."""
    import torch
    return torch.exp(torch.linspace(math.log(min), math.log(max), levels))


def _module_base():
    import torch
    return torch.nn.Module


def _one_call_at_a_time(fn):
    """The C object is not re-entrant (a second caller gets PC_ERR_STATE) and the strings of a compress() are fetched after the call:
    calls into one object from several host threads are serialised here.  Use two objects to run an encoder and a decoder side by side."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *a, **kw):
        with self.__dict__["_call_lock"]:
            return fn(self, *a, **kw)
    return wrapper


class ChannelProgresssiveWACNN(_module_base()):
    """A ``torch.nn.Module`` (isinstance checks, ``state_dict()``, ``parameters()``, ``eval()`` ... behave as the caller of the reference
    expects: SURVEY.md section 8b) whose tensors are host-side copies of what was loaded -- the working weights live packed in
    HBM inside the native codec object."""

    def __init__(self, N=192, M=640, division_dimension=(320, 640), dim_chunk=32, multiple_decoder=True,
                 multiple_encoder=False, multiple_hyperprior=True, mask_policy="two-levels", lmbda_list=(0.0055, 0.04),
                 joiner_policy="res", support_progressive_slices=5, delta_encode=True, device="cuda:0", **kwargs):
        super().__init__()
        self.cfg = CodecConfig(N=N, M=M, division_dimension=tuple(division_dimension), dim_chunk=dim_chunk,
                               multiple_decoder=multiple_decoder, multiple_encoder=multiple_encoder,
                               multiple_hyperprior=multiple_hyperprior, delta_encode=delta_encode,
                               joiner_policy=joiner_policy, support_progressive_slices=support_progressive_slices,
                               mask_policy=mask_policy)
        self.cfg.check_supported()
        self.mask_policy = mask_policy
        self.lmbda_list = list(lmbda_list)
        import torch
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("progressivecodec_amd runs on a HIP device only (no CPU fallback)")
        self.__dict__["_call_lock"] = __import__("threading").RLock()
        self._h = C.c_void_p()
        check(lib().pc_codec_create(C.byref(self._h), self.device.index or 0), "pc_codec_create")
        self._sd = None
        self._gc = None
        self._eb = None
        self._scale_table = None
        self._finalized = False

    def __del__(self):
        h = self.__dict__.get("_h", None)
        if h is not None and h.value:
            lib().pc_codec_destroy(h)
            self.__dict__["_h"] = C.c_void_p()

    # ------------------------------------------------------------------ nn.Module surface
    def to(self, *args, **kwargs):
        """The codec object is bound to the HIP device it was created on; moving it is not supported (and never needed on the
        compress()/decompress() path)."""
        return self

    def forward(self, *args, **kwargs):
        raise NotImplementedError("training forward() (CHProg_cnn.py:478-682) is out of scope; use forward_single_quality() for rate estimation")

    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        """The reference's 1019-key layout (models/cnn.py:195-202), as loaded, with the CDF buffers update() built."""
        import torch
        out = OrderedDict() if destination is None else destination
        if self._sd is None:
            return out
        for k, a in self._sd.items():
            out[prefix + k] = torch.from_numpy(np.array(a, copy=True))
        for which, p in ((self._gc, "gaussian_conditional"), (self._eb, "entropy_bottleneck")):
            if which is not None:
                out[prefix + p + "._quantized_cdf"] = torch.from_numpy(which.cdf.copy())
                out[prefix + p + "._cdf_length"] = torch.from_numpy(which.length.copy())
                out[prefix + p + "._offset"] = torch.from_numpy(which.offset.copy())
        if self._scale_table is not None:
            out[prefix + "gaussian_conditional.scale_table"] = torch.from_numpy(self._scale_table.copy())
        return out

    def named_parameters(self, prefix="", recurse=True, remove_duplicate=True):
        import torch
        if self._sd is None:
            return
        spec = param_spec(self.cfg)
        for k, a in self._sd.items():
            if spec[k][2] in _PARAM_KINDS:
                yield prefix + k, torch.nn.Parameter(torch.from_numpy(np.array(a, copy=True)), requires_grad=False)

    def parameters(self, recurse=True):
        for _, p in self.named_parameters():
            yield p

    def named_buffers(self, prefix="", recurse=True, remove_duplicate=True):
        pnames = {k for k, _ in self.named_parameters()}
        for k, v in self.state_dict().items():
            if k not in pnames:
                yield prefix + k, v

    def buffers(self, recurse=True):
        for _, b in self.named_buffers():
            yield b

    def load_state_dict(self, state_dict, strict=True, _extra=None):
        """models/cnn.py:195-202 / base.py:62-70: accepts the reference's 1019-key state_dict.  (_extra: tensors of a wrapping model --
        the REM's post_latent.* -- handed to the native codec before it is finalised.)"""
        if self._finalized:
            # a second load (the reference's usual REM flow: load the base net, wrap it, then rem.load_state_dict(base, post) --
            # CHProgREM.py:361-369): the native object is immutable once finalised, so it is replaced by a fresh one
            with self._call_lock:
                lib().pc_codec_destroy(self._h)
                self._h = C.c_void_p()
                check(lib().pc_codec_create(C.byref(self._h), self.device.index or 0), "pc_codec_create")
                self._finalized = False
                # the fresh handle holds no tables and the checkpoint's own scale table: forget the old handle's (ADVICE r03 -- update()
                # would otherwise see them set and skip, leaving the native side without CDFs or with CDFs of another scale table)
                self._gc = self._eb = self._scale_table = None
        for k, a in (_extra or {}).items():
            a = np.ascontiguousarray(a, np.float32)
            shp = (C.c_int64 * a.ndim)(*a.shape)
            check(lib().pc_codec_set_tensor(self._h, k.encode(), a.ctypes.data_as(C.c_void_p), _DT["float32"], shp, a.ndim), f"set_tensor({k})")
        spec = param_spec(self.cfg)
        missing = [k for k in spec if k not in state_dict]
        unexpected = [k for k in state_dict if k not in spec]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing[:5]}... unexpected {unexpected[:5]}...")
        sd = OrderedDict()
        for k, (shape, dtype, kind) in spec.items():
            if k not in state_dict:
                continue
            v = state_dict[k]
            a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            a = np.ascontiguousarray(a.astype(dtype, copy=False))
            if kind != "table" and tuple(a.shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {k}: {tuple(a.shape)} vs {tuple(shape)}")
            sd[k] = a
            if kind == "table":
                continue
            shp = (C.c_int64 * a.ndim)(*a.shape)
            check(lib().pc_codec_set_tensor(self._h, k.encode(), a.ctypes.data_as(C.c_void_p), _DT[dtype], shp, a.ndim),
                  f"set_tensor({k})")
        self._sd = sd
        check(lib().pc_codec_finalize(self._h), "pc_codec_finalize")
        self._finalized = True
        for which, p in ((0, "gaussian_conditional"), (1, "entropy_bottleneck")):
            cdf = sd.get(p + "._quantized_cdf")
            if cdf is not None and cdf.size > 0:
                self._set_tables(which, entropy.CdfTables(cdf, sd[p + "._cdf_length"], sd[p + "._offset"]))
        return self

    def _set_tables(self, which, t):
        check(lib().pc_codec_set_tables(self._h, which, t.cdf.ctypes.data_as(C.c_void_p), t.cdf.shape[0], t.cdf.shape[1],
                                        t.length.ctypes.data_as(C.c_void_p), t.offset.ctypes.data_as(C.c_void_p)),
              "pc_codec_set_tables")
        if which == 0:
            self._gc = t
        else:
            self._eb = t

    def update(self, scale_table=None, force=False):
        """models/cnn.py:137-142: ``scale_table`` defaults to get_scale_table(); GaussianConditional.update_scale_table
        (entropy_models.py:588-597) rebuilds the Gaussian tables only if none are loaded yet or ``force``; then
        CompressionModel.update (base.py:41-60) does the same for the EntropyBottleneck.  Returns whether anything was rebuilt."""
        if self._sd is None:
            raise ValueError("load_state_dict() first")
        updated = False
        if self._gc is None or force:
            st = get_scale_table() if scale_table is None else scale_table
            st = np.ascontiguousarray([float(v) for v in (st.tolist() if hasattr(st, "tolist") else st)], np.float32)   # _prepare_scale_table :575-576
            cur = self._scale_table if self._scale_table is not None else self._sd["gaussian_conditional.scale_table"]
            if st.shape != cur.shape or not np.array_equal(st, cur):
                check(lib().pc_codec_set_scale_table(self._h, st.ctypes.data_as(C.c_void_p), st.size), "pc_codec_set_scale_table")
            self._scale_table = st
            self._set_tables(0, entropy.gaussian_conditional_tables(st))
            updated = True
        if self._eb is None or force:
            self._set_tables(1, entropy.entropy_bottleneck_tables(self._sd))
            updated = True
        return updated

    def set_option(self, name, value):
        """pc_codec_set_option: schedule options of this object ("serial_schedule", "lanes_enc", "lanes_dec", "host_threads") -- results never
        depend on them.  Options belong to the native handle: a reload of a finalised codec (load_state_dict) starts from the defaults."""
        check(lib().pc_codec_set_option(self._h, name.encode(), int(value)), f"pc_codec_set_option({name})")
        return self

    # ------------------------------------------------------------------ the hot path
    def _stream(self):
        import torch
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _fetch_strings(self):
        """every byte string of the last compress as a flat list (slot-major, then z), through one bulk copy"""
        tot, n = C.c_size_t(), C.c_int()
        check(lib().pc_codec_strings_size(self._h, C.byref(tot), C.byref(n)), "pc_codec_strings_size")
        buf = C.create_string_buffer(max(1, tot.value))
        lens = (C.c_size_t * n.value)()
        check(lib().pc_codec_copy_strings(self._h, buf, tot.value, lens, n.value), "pc_codec_copy_strings")
        raw = buf.raw
        out, off = [], 0
        for k in lens:
            out.append(raw[off:off + k])
            off += k
        return out

    def _decompress_packed(self, slots, z_strings, B, zh, zw, qualities, mask_pol):
        """slots: list of B-lists of byte strings in slot order; returns x_hat [L, B, 3, H, W]"""
        import torch
        flat = [s for sl in slots for s in sl] + list(z_strings)
        data = b"".join(flat)
        lens = (C.c_size_t * len(flat))(*map(len, flat))
        L = len(qualities)
        qa = (C.c_double * L)(*qualities)
        x_hat = torch.empty((L, B, 3, 64 * zh, 64 * zw), device=self.device, dtype=torch.float32)
        check(lib().pc_codec_decompress_packed(self._h, data, lens, B, zh, zw, qa, L, _MASK_POL[mask_pol], C.c_void_p(x_hat.data_ptr()),
                                               self._stream()), "pc_codec_decompress_packed")
        return x_hat

    def _set_cust_map(self, cust_map, B, h, w):
        """cust_map: [B, 320, H/16, W/16] importance map whose per-slice quantile replaces the scale's (layers/masking.py:171-194).
        Returns the device tensor (keep it alive until the call has been issued)."""
        import torch
        if cust_map is None:
            check(lib().pc_codec_set_cust_map(self._h, None), "pc_codec_set_cust_map")
            return None
        if tuple(cust_map.shape) != (B, 320, h, w):
            raise ValueError(f"cust_map must be [B, 320, H/16, W/16] = {(B, 320, h, w)}, got {tuple(cust_map.shape)}")
        cm = cust_map.to(self.device, torch.float32).contiguous()
        check(lib().pc_codec_set_cust_map(self._h, C.c_void_p(cm.data_ptr())), "pc_codec_set_cust_map")
        return cm

    @_one_call_at_a_time
    def compress(self, x, quality=0.0, mask_pol=None, cust_map=None):
        """CHProg_cnn.py:686-847.  Returns {"strings": [y_strings, z_strings], "shape", "masks"}."""
        import torch
        mask_pol = self.mask_policy if mask_pol is None else mask_pol
        if mask_pol not in _MASK_POL:
            raise NotImplementedError(f"mask policy {mask_pol!r}")
        if self._gc is None or self._eb is None:
            raise ValueError("Uninitialized CDFs. Run update() first")           # entropy_models.py:182-184
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("Invalid `inputs` size. Expected a [B,3,H,W] tensor.")
        B, _, H, W = x.shape
        if H % 64 or W % 64:
            raise ValueError("H and W must be multiples of 64 (pad as training/step.py:318 does)")
        x = x.to(self.device, torch.float32).contiguous()
        h, w = H // 16, W // 16
        n_enh = 0 if quality <= 0 else 10
        masks = torch.empty((10, B, 32, h, w), device=self.device, dtype=torch.float32) if n_enh else None
        cm = self._set_cust_map(cust_map if quality > 0 else None, B, h, w)      # CHProg_cnn.py:721-722
        check(lib().pc_codec_compress(self._h, C.c_void_p(x.data_ptr()), B, H, W, float(quality), _MASK_POL[mask_pol],
                                      C.c_void_p(masks.data_ptr()) if masks is not None else None, self._stream()),
              "pc_codec_compress")
        ns = lib().pc_codec_num_slices(self._h)
        strs = self._fetch_strings()                                             # all slots, then z: one copy
        y_strings = [strs[s * B:(s + 1) * B] for s in range(ns)]
        z_strings = strs[-B:]
        return {"strings": [y_strings, z_strings], "shape": torch.Size([H // 64, W // 64]),
                "masks": [masks[i] for i in range(10)] if masks is not None else []}

    @_one_call_at_a_time
    def decompress(self, strings, shape, quality, mask_pol=None, cust_map=None):
        """CHProg_cnn.py:849-999.  Returns {"x_hat": Tensor[B,3,H,W] in [0,1]}."""
        import torch
        mask_pol = self.mask_policy if mask_pol is None else mask_pol
        if mask_pol not in _MASK_POL:
            raise NotImplementedError(f"mask policy {mask_pol!r}")
        if self._gc is None or self._eb is None:
            raise ValueError("Uninitialized CDFs. Run update() first")
        if not isinstance(strings, (tuple, list)) or len(strings) != 2:
            raise ValueError("Invalid `strings` parameter type.")                # entropy_models.py:250-251
        y_strings, z_strings = strings
        B = len(z_strings)
        ns = len(y_strings)
        if any(len(s) != B for s in y_strings):
            raise ValueError("Invalid strings or indexes parameters")            # entropy_models.py:253-254
        zh, zw = int(shape[0]), int(shape[1])
        if ns < (10 if quality == 0 else 20):
            raise ValueError("Invalid strings or indexes parameters")
        cm = self._set_cust_map(cust_map if quality > 0 else None, B, 4 * zh, 4 * zw)                     # CHProg_cnn.py:850-851
        slots = [list(sl) for sl in y_strings[:20]] + [[b""] * B for _ in range(20 - min(ns, 20))]
        x_hat = self._decompress_packed(slots, z_strings, B, zh, zw, [float(quality)], mask_pol)[0]
        del cm
        return {"x_hat": x_hat}

    # ------------------------------------------------------------------ likelihood (rate estimation) path
    @_one_call_at_a_time
    def forward_single_quality(self, x, quality, mask_pol="point-based-std", force_enhanced=False, training=False):
        """CHProg_cnn.py:1002-1198 in eval mode -- what test_epoch / valid_epoch call (training/step.py:215-267).
        Returns {"x_hat", "likelihoods": {"y", "z"}, "masks"}: y [B, 320 or 640, H/16, W/16], z [B, 192, H/64, W/64]; estimated
        bits = -sum(log2(likelihood)).  The auxiliary entries of the reference's dictionary (y_hat, mu, std ...) are not
        returned.  Training-mode noise is out of scope.  force_enhanced with quality 0: enhancement path with all-zero masks (:1006,1022,1064)."""
        import torch
        if training:
            raise NotImplementedError("only the eval path of forward_single_quality is implemented (SURVEY.md section 8f)")
        mask_pol = self.mask_policy if mask_pol is None else mask_pol
        if mask_pol not in _MASK_POL:
            raise NotImplementedError(f"mask policy {mask_pol!r}")
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("Invalid `inputs` size. Expected a [B,3,H,W] tensor.")
        B, _, H, W = x.shape
        if H % 64 or W % 64:
            raise ValueError("H and W must be multiples of 64 (pad as training/step.py:318 does)")
        x = x.to(self.device, torch.float32).contiguous()
        h, w = H // 16, W // 16
        enh = quality != 0 or bool(force_enhanced)                               # CHProg_cnn.py:1022,1064
        nch = 640 if enh else 320
        x_hat = torch.empty((B, 3, H, W), device=self.device, dtype=torch.float32)
        y_lik = torch.empty((B, nch, h, w), device=self.device, dtype=torch.float32)
        z_lik = torch.empty((B, 192, H // 64, W // 64), device=self.device, dtype=torch.float32)
        masks = torch.empty((10, B, 32, h, w), device=self.device, dtype=torch.float32) if enh else None
        P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        check(lib().pc_codec_forward(self._h, P(x), B, H, W, float(quality), _MASK_POL[mask_pol], P(x_hat), P(y_lik), P(z_lik), P(masks),
                                     1 if force_enhanced else 0, self._stream()), "pc_codec_forward")
        return {"x_hat": x_hat, "likelihoods": {"y": y_lik, "z": z_lik},
                "masks": [masks[i] for i in range(10)] if masks is not None else []}

    # ------------------------------------------------------------------ multi-level (shared base) coding
    @_one_call_at_a_time
    def compress_levels(self, x, qualities, mask_pol=None):
        """compress() for a list of mask levels with the level-independent part (g_a, h_a, z, h_s, the ten base slices;
        CHProg_cnn.py:692-767) computed once -- SURVEY.md section 8(f) rank 1.  Returns one compress()-style dictionary per
        level; the z strings and the ten base y strings are the same objects in every entry, and every entry equals what
        compress(x, q, mask_pol) returns for that level."""
        import torch
        mask_pol = self.mask_policy if mask_pol is None else mask_pol
        if mask_pol not in _MASK_POL:
            raise NotImplementedError(f"mask policy {mask_pol!r}")
        if self._gc is None or self._eb is None:
            raise ValueError("Uninitialized CDFs. Run update() first")
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("Invalid `inputs` size. Expected a [B,3,H,W] tensor.")
        qualities = [float(q) for q in qualities]
        if not qualities:
            raise ValueError("at least one level")
        B, _, H, W = x.shape
        if H % 64 or W % 64:
            raise ValueError("H and W must be multiples of 64 (pad as training/step.py:318 does)")
        x = x.to(self.device, torch.float32).contiguous()
        h, w = H // 16, W // 16
        L = len(qualities)
        masks = [torch.empty((10, B, 32, h, w), device=self.device, dtype=torch.float32) if q > 0 else None for q in qualities]
        mp = (C.c_void_p * L)(*[m.data_ptr() if m is not None else None for m in masks])
        qa = (C.c_double * L)(*qualities)
        check(lib().pc_codec_compress_levels(self._h, C.c_void_p(x.data_ptr()), B, H, W, qa, L, _MASK_POL[mask_pol], mp, self._stream()),
              "pc_codec_compress_levels")
        strs = self._fetch_strings()                                             # slots [10 + 10*L][B], then z
        slot = lambda k: strs[k * B:(k + 1) * B]
        z_strings = strs[-B:]
        base = [slot(s) for s in range(10)]
        out = []
        for lv, q in enumerate(qualities):
            enh = [slot(10 + 10 * lv + s) for s in range(10)] if q > 0 else []
            out.append({"strings": [base + enh, z_strings], "shape": torch.Size([H // 64, W // 64]),
                        "masks": [masks[lv][i] for i in range(10)] if masks[lv] is not None else []})
        return out

    @_one_call_at_a_time
    def decompress_levels(self, strings_per_level, shape, qualities, mask_pol=None):
        """decompress() for a list of levels of the same images: z, h_s and the ten base slices are decoded once
        (CHProg_cnn.py:855-904), each level decodes its enhancement chain and runs its synthesis transform.
        strings_per_level[l] = [y_strings, z_strings] as returned by compress()/compress_levels() for level l (the base and z
        strings are taken from entry 0).  Returns one {"x_hat"} dictionary per level, equal to decompress() of that level."""
        import torch
        mask_pol = self.mask_policy if mask_pol is None else mask_pol
        if mask_pol not in _MASK_POL:
            raise NotImplementedError(f"mask policy {mask_pol!r}")
        if self._gc is None or self._eb is None:
            raise ValueError("Uninitialized CDFs. Run update() first")
        qualities = [float(q) for q in qualities]
        L = len(qualities)
        if L == 0 or len(strings_per_level) != L:
            raise ValueError("one [y_strings, z_strings] entry per level")
        for st in strings_per_level:
            if not isinstance(st, (tuple, list)) or len(st) != 2:
                raise ValueError("Invalid `strings` parameter type.")
        z_strings = list(strings_per_level[0][1])
        B = len(z_strings)
        zh, zw = int(shape[0]), int(shape[1])
        slots = [list(sl) for sl in strings_per_level[0][0][:10]]
        if len(slots) != 10:
            raise ValueError("Invalid strings or indexes parameters")
        for lv, q in enumerate(qualities):
            ys = strings_per_level[lv][0]
            if q != 0:
                if len(ys) < 20:
                    raise ValueError("Invalid strings or indexes parameters")
                slots += [list(sl) for sl in ys[10:20]]
            else:
                slots += [[b""] * B for _ in range(10)]
        if any(len(sl) != B for sl in slots):
            raise ValueError("Invalid strings or indexes parameters")
        x_hat = self._decompress_packed(slots, z_strings, B, zh, zw, qualities, mask_pol)
        return [{"x_hat": x_hat[lv]} for lv in range(L)]

    def read_latent(self, name, B, h, w):
        """the decoded latent of the last call ("yhat_base" / "yhat_enh": NHWC [B*h*w][320] inside the codec) as a host tensor
        [B, 320, h, w] -- the "y_hat" entry of the REM's dictionaries (CHProgREM.py:888,1122)"""
        import torch
        a = self.read_tap(name)[: B * h * w * 320].reshape(B, h, w, 320)
        return torch.from_numpy(np.ascontiguousarray(a.transpose(0, 3, 1, 2)))

    # ------------------------------------------------------------------ test taps
    def read_tap(self, name, dtype=np.float32):
        n = C.c_size_t()
        f = lib().pc_codec_read_tap
        check(f(self._h, name.encode(), None, 0, C.byref(n)), "read_tap")
        out = np.empty(n.value, dtype)
        check(f(self._h, name.encode(), out.ctypes.data_as(C.c_void_p), n.value, C.byref(n)), "read_tap")
        return out
