"""CPU oracle for ChannelProgresssiveWACNN.compress()/decompress()  (TEST INFRASTRUCTURE ONLY).

A functional restatement of the reference's inference path for the canonical
configuration, with two interchangeable float back-ends:

* ``backend="torch"``  -- the same ATen CPU ops the reference calls (F.conv2d, ...).  On one
  machine this reproduces the reference's byte strings exactly; tests/golden pins it
  (tests/test_oracle_vs_golden.py).  It is also bench.py's ``cpu_baseline`` ("port").
* ``backend="cdet"``   -- the float primitives of oracle/pc_oracle.c: every conv / linear / GDN
  contraction is one fmaf chain in the order of the numeric contract and the
  transcendental functions come from include/pc_math.h.  This is what the HIP kernels are
  compared against bit-for-bit.

The integer stages (index, quantile mask, quantise, rANS) are the C restatements of
pc_oracle.c in both back-ends.  Nothing here is imported by progressivecodec_amd/.

Citations are to files under /root/reference/src/compress/.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import liboracle as lo

NS0, NS1 = 10, 20           # base / total slices   (models/CHProg_cnn.py:77-78,109-110)
D0 = 320                    # division_dimension[0]
MAX_SUPPORT = 5             # models/cnn.py:30 ; support_progressive_slices = 5
HEADS = 8


# ----------------------------------------------------------------------------- back-ends
class TorchOps:
    """ATen CPU ops, called as the reference calls them."""
    name = "torch"

    def conv(self, x, w, b, stride, pad):
        return F.conv2d(x, w, b, stride=stride, padding=pad)

    def deconv(self, x, w, b):                      # models/utils.py:196-204
        return F.conv_transpose2d(x, w, b, stride=2, padding=2, output_padding=1)

    def gelu(self, x):
        return F.gelu(x)

    def tanh(self, x):
        return torch.tanh(x)

    def sigmoid(self, x):
        return torch.sigmoid(x)

    def leaky(self, x):                             # nn.LeakyReLU() default slope (models/utils.py:65)
        return F.leaky_relu(x, 0.01)

    def gdn(self, x, beta, gamma, inverse):         # layers/gdn.py:50-63
        C = x.shape[1]
        norm = F.conv2d(x ** 2, gamma.reshape(C, C, 1, 1), beta)
        norm = torch.sqrt(norm) if inverse else torch.rsqrt(norm)
        return x * norm

    def win_attention(self, x, p, ws, shift):       # layers/win_attention.py:153-207, :84-115
        B, C, H, W = x.shape
        T, d = ws * ws, C // HEADS
        t = x.permute(0, 2, 3, 1)
        if shift > 0:
            t = torch.roll(t, shifts=(-shift, -shift), dims=(1, 2))
        win = t.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, T, C)
        qkv = F.linear(win, p["qkv.weight"], p["qkv.bias"]).reshape(-1, T, 3, HEADS, d).permute(2, 0, 3, 1, 4).contiguous()
        q, k, v = qkv[0], qkv[1], qkv[2]
        q = q * (d ** -0.5)
        attn = q @ k.transpose(-2, -1)
        bias = p["relative_position_bias_table"][p["relative_position_index"].view(-1)].view(T, T, -1)
        attn = attn + bias.permute(2, 0, 1).contiguous().unsqueeze(0)
        if shift > 0:
            m = shift_mask(H, W, ws, shift)
            nW = m.shape[0]
            attn = attn.view(-1, nW, HEADS, T, T) + m.unsqueeze(1).unsqueeze(0)
            attn = attn.view(-1, HEADS, T, T)
        attn = torch.softmax(attn, dim=-1)
        o = (attn @ v).transpose(1, 2).reshape(-1, T, C)
        o = F.linear(o, p["proj.weight"], p["proj.bias"])
        o = o.view(B, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).contiguous().view(B, H, W, C)
        if shift > 0:
            o = torch.roll(o, shifts=(shift, shift), dims=(1, 2))
        return x + o.permute(0, 3, 1, 2).contiguous()


def shift_mask(H, W, ws, shift):
    """win_attention.py:157-175: 0 / -100 mask between tokens of different wrap-around regions."""
    img = torch.zeros((1, H, W, 1))
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    mw = img.view(1, H // ws, ws, W // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).contiguous().view(-1, ws * ws)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    return am.masked_fill(am != 0, -100.0).masked_fill(am == 0, 0.0)


def _nhwc(x):
    return np.ascontiguousarray(x.permute(0, 2, 3, 1).numpy())


def _nchw(a):
    return torch.from_numpy(np.ascontiguousarray(a.transpose(0, 3, 1, 2)))


class CDetOps:
    """Numeric-contract primitives (oracle/pc_oracle.c).  Chain order: flattened k = tap * Cin + channel
    (taps in (ky, kx) ascending), aligned groups of 8 visited 0,4,1,5,2,6,3,7 inside a group;
    bias added after the chain."""
    name = "cdet"

    def conv(self, x, w, b, stride, pad):
        co, ci, kh, kw = w.shape
        B, _, H, W = x.shape
        Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
        taps = [(ky - pad, kx - pad) for ky in range(kh) for kx in range(kw)]
        wt = w.permute(2, 3, 1, 0).reshape(kh * kw, ci, co).numpy()
        acc = lo.conv_nhwc(_nhwc(x), wt, taps, stride, Ho, Wo)
        if b is not None:
            acc = acc + b.numpy().reshape(1, 1, 1, -1)
        return _nchw(acc)

    def deconv(self, x, w, b):
        # ConvTranspose2d(k5, s2, p2, op1): out[2i+py, 2j+px] = sum over ky = py (mod 2), kx = px (mod 2) of
        # in[i + (py+2-ky)/2, j + (px+2-kx)/2] * w[:, :, ky, kx]
        ci, co, kh, kw = w.shape
        B, _, H, W = x.shape
        xn = _nhwc(x)
        out = np.zeros((B, 2 * H, 2 * W, co), np.float32)
        for py in range(2):
            for px in range(2):
                kys, kxs = range(py, 5, 2), range(px, 5, 2)
                taps = [((py + 2 - ky) // 2, (px + 2 - kx) // 2) for ky in kys for kx in kxs]
                wt = torch.stack([w[:, :, ky, kx] for ky in kys for kx in kxs]).numpy()
                lo.conv_nhwc(xn, wt, taps, 1, H, W, out=out, ostride=(2, 2), ooff=(py, px))
        if b is not None:
            out = out + b.numpy().reshape(1, 1, 1, -1)
        return _nchw(out)

    def gelu(self, x):
        return torch.from_numpy(lo.unary(x.numpy(), "gelu"))

    def tanh(self, x):
        return torch.from_numpy(lo.unary(x.numpy(), "tanh"))

    def sigmoid(self, x):
        return torch.from_numpy(lo.unary(x.numpy(), "sigmoid"))

    def leaky(self, x):                             # x > 0 ? x : x * float32(0.01): one correctly rounded multiply, same on every device
        a = x.numpy()
        return torch.from_numpy(np.where(a > 0, a, a * np.float32(0.01)).astype(np.float32))

    def gdn(self, x, beta, gamma, inverse):
        B, C, H, W = x.shape
        xn = _nhwc(x)
        wt = gamma.t().contiguous().reshape(1, C, C).numpy()       # [j][i] = gamma[i][j]
        norm = lo.conv_nhwc(xn, wt, [(0, 0)], 1, H, W, square=True) + beta.numpy().reshape(1, 1, 1, -1)
        norm = lo.unary(norm, "sqrt" if inverse else "rsqrt")
        return _nchw(xn * norm)

    def win_attention(self, x, p, ws, shift):
        B, C, H, W = x.shape
        T, d = ws * ws, C // HEADS
        xn = _nhwc(x)
        qkv = lo.conv_nhwc(xn, p["qkv.weight"].t().contiguous().reshape(1, C, 3 * C).numpy(), [(0, 0)], 1, H, W)
        qkv = qkv + p["qkv.bias"].numpy().reshape(1, 1, 1, -1)
        bias = p["relative_position_bias_table"][p["relative_position_index"].view(-1)].view(T, T, -1)
        bias = bias.permute(2, 0, 1).contiguous().numpy()
        o = lo.win_attention(qkv, bias, HEADS, ws, shift, np.float32(d ** -0.5))
        o = lo.conv_nhwc(o, p["proj.weight"].t().contiguous().reshape(1, C, C).numpy(), [(0, 0)], 1, H, W)
        o = o + p["proj.bias"].numpy().reshape(1, 1, 1, -1)
        return _nchw(xn + o)


# ----------------------------------------------------------------------------- tables
def gdn_params(sd, p):
    """NonNegativeParametrizer.forward, ops/parametrizers.py:46-49 (LowerBound then square minus pedestal)."""
    beta = torch.max(sd[p + ".beta"], sd[p + ".beta_reparam.lower_bound.bound"]) ** 2 - sd[p + ".beta_reparam.pedestal"]
    gamma = torch.max(sd[p + ".gamma"], sd[p + ".gamma_reparam.lower_bound.bound"]) ** 2 - sd[p + ".gamma_reparam.pedestal"]
    return beta, gamma


def _pmf_to_cdf(pmf, tail_mass, pmf_length, max_length):
    """EntropyModel._pmf_to_cdf, entropy_models/entropy_models.py:172-180."""
    cdf = np.zeros((len(pmf_length), max_length + 2), np.int32)
    for i in range(len(pmf_length)):
        prob = np.concatenate([pmf[i, : pmf_length[i]], tail_mass[i]])
        c = lo.pmf_to_quantized_cdf(prob, 16)
        cdf[i, : c.size] = c.astype(np.int32)
    return cdf


def gaussian_conditional_tables(scale_table: torch.Tensor, tail_mass=1e-9) -> lo.Tables:
    """GaussianConditional.update, entropy_models.py:599-624."""
    import scipy.stats
    multiplier = -scipy.stats.norm.ppf(tail_mass / 2)
    pmf_center = torch.ceil(scale_table * multiplier).int()
    pmf_length = 2 * pmf_center + 1
    max_length = int(pmf_length.max())
    samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
    s = scale_table.unsqueeze(1).float()
    cum = lambda v: 0.5 * torch.erfc(float(-(2 ** -0.5)) * v)
    upper = cum((0.5 - samples) / s)
    lower = cum((-0.5 - samples) / s)
    pmf = upper - lower
    tail = 2 * lower[:, :1]
    cdf = _pmf_to_cdf(pmf.numpy(), tail.numpy(), pmf_length.numpy(), max_length)
    return lo.Tables(cdf, (pmf_length + 2).numpy(), (-pmf_center).numpy())


def entropy_bottleneck_tables(sd, prefix="entropy_bottleneck") -> lo.Tables:
    """EntropyBottleneck.update, entropy_models.py:354-393 (+ _logits_cumulative :400-419)."""
    q = sd[prefix + ".quantiles"]
    medians = q[:, 0, 1]
    minima = torch.clamp(torch.ceil(medians - q[:, 0, 0]).int(), min=0)
    maxima = torch.clamp(torch.ceil(q[:, 0, 2] - medians).int(), min=0)
    pmf_start = medians - minima
    pmf_length = maxima + minima + 1
    max_length = int(pmf_length.max())
    samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]

    def logits(x):
        for i in range(5):
            x = torch.matmul(F.softplus(sd[f"{prefix}._matrix{i}"]), x)
            x = x + sd[f"{prefix}._bias{i}"]
            if i < 4:
                x = x + torch.tanh(sd[f"{prefix}._factor{i}"]) * torch.tanh(x)
        return x

    lower, upper = logits(samples - 0.5), logits(samples + 0.5)
    sign = -torch.sign(lower + upper)
    pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
    tail = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
    cdf = _pmf_to_cdf(pmf.numpy(), tail.numpy(), pmf_length.numpy(), max_length)
    return lo.Tables(cdf, (pmf_length + 2).numpy(), (-minima).numpy())


# ----------------------------------------------------------------------------- the codec
class RefCodec:
    def __init__(self, state_dict, backend="torch", gc_tables=None, eb_tables=None, num_threads=None):
        self.sd = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))) for k, v in state_dict.items()}
        self.ops = TorchOps() if backend == "torch" else CDetOps()
        self.scale_table = self.sd["gaussian_conditional.scale_table"].float()
        self.scale_bound = float(self.sd["gaussian_conditional.lower_bound_scale.bound"][0])
        self.gc = gc_tables
        self.eb = eb_tables
        if self.gc is None and self.sd["gaussian_conditional._quantized_cdf"].numel() > 0:
            g = "gaussian_conditional"
            self.gc = lo.Tables(self.sd[g + "._quantized_cdf"].numpy(), self.sd[g + "._cdf_length"].numpy(), self.sd[g + "._offset"].numpy())
        if self.eb is None and self.sd["entropy_bottleneck._quantized_cdf"].numel() > 0:
            g = "entropy_bottleneck"
            self.eb = lo.Tables(self.sd[g + "._quantized_cdf"].numpy(), self.sd[g + "._cdf_length"].numpy(), self.sd[g + "._offset"].numpy())
        self.medians = self.sd["entropy_bottleneck.quantiles"][:, 0, 1].contiguous()   # _get_medians :350

    def update(self):                                   # models/cnn.py:137-142
        self.gc = gaussian_conditional_tables(self.scale_table)
        self.eb = entropy_bottleneck_tables(self.sd)
        return True

    # ---- layers
    def _c(self, x, p, stride=1, pad=None):
        w = self.sd[p + ".weight"]
        return self.ops.conv(x, w, self.sd[p + ".bias"], stride, w.shape[-1] // 2 if pad is None else pad)

    def _ru(self, x, p):                                # layers/layers.py:38-57
        o = self.ops.gelu(self._c(x, p + ".conv.0"))
        o = self.ops.gelu(self._c(o, p + ".conv.2"))
        o = self._c(o, p + ".conv.4")
        return self.ops.gelu(o + x)

    def _wam(self, x, p, ws, shift):                    # layers/layers.py:59-75
        a = x
        for i in range(3):
            a = self._ru(a, f"{p}.conv_a.{i}")
        ap = {k: self.sd[f"{p}.conv_b.0.attn.{k}"] for k in
              ("qkv.weight", "qkv.bias", "proj.weight", "proj.bias", "relative_position_bias_table", "relative_position_index")}
        b = self.ops.win_attention(x, ap, ws, shift)
        for i in range(1, 4):
            b = self._ru(b, f"{p}.conv_b.{i}")
        b = self._c(b, p + ".conv_b.4")
        return a * self.ops.sigmoid(b) + x

    def _gdn(self, x, p, inverse=False):
        beta, gamma = gdn_params(self.sd, p)
        return self.ops.gdn(x, beta, gamma, inverse)

    def _g_a_net(self, x, p):                           # models/cnn.py:34-44
        x = self._gdn(self._c(x, p + ".0", 2), p + ".1")
        x = self._gdn(self._c(x, p + ".2", 2), p + ".3")
        x = self._wam(x, p + ".4", 8, 4)
        x = self._gdn(self._c(x, p + ".5", 2), p + ".6")
        x = self._c(x, p + ".7", 2)
        return self._wam(x, p + ".8", 4, 2)

    def g_a(self, x):
        if "g_a.0.0.weight" in self.sd:                 # multiple_encoder: CHProg_cnn.py:131-144, used at :691-697
            return torch.cat([self._g_a_net(x, "g_a.0"), self._g_a_net(x, "g_a.1")], 1)
        return self._g_a_net(x, "g_a")

    def g_s(self, k, y):                                # models/CHProg_cnn.py:149-161
        p = f"g_s.{k}"
        d = lambda x, q: self.ops.deconv(x, self.sd[q + ".weight"], self.sd[q + ".bias"])
        x = self._wam(y, p + ".0", 4, 2)
        x = self._gdn(d(x, p + ".1"), p + ".2", True)
        x = self._gdn(d(x, p + ".3"), p + ".4", True)
        x = self._wam(x, p + ".5", 8, 4)
        x = self._gdn(d(x, p + ".6"), p + ".7", True)
        return d(x, p + ".8")

    def h_a(self, y):                                   # models/cnn.py:57-67
        g = self.ops.gelu
        x = g(self._c(y, "h_a.0"))
        x = g(self._c(x, "h_a.2"))
        x = g(self._c(x, "h_a.4", 2))
        x = g(self._c(x, "h_a.6"))
        return self._c(x, "h_a.8", 2)

    def h_s(self, fam, k, z):                           # models/CHProg_cnn.py:208-232
        p, g = f"{fam}.{k}", self.ops.gelu
        x = g(self._c(z, p + ".0"))
        x = g(F.pixel_shuffle(self._c(x, p + ".2.0"), 2))
        x = g(self._c(x, p + ".4"))
        x = g(F.pixel_shuffle(self._c(x, p + ".6.0"), 2))
        return self._c(x, p + ".8")

    def stack5(self, fam, i, x):                        # models/CHProg_cnn.py:165-203,235-274
        p = f"{fam}.{i}"
        for j in range(4):
            x = self.ops.gelu(self._c(x, f"{p}.{2 * j}"))
        return self._c(x, f"{p}.8")

    # ---- entropy stages
    def _indexes(self, scale):                          # entropy_models.py:661-666
        return torch.from_numpy(lo.build_indexes(scale.numpy(), self.scale_table.numpy(), self.scale_bound))

    def _mask(self, scale, pr, mask_pol, cust_map=None):   # layers/masking.py:163-247
        if cust_map is not None:                        # :171-194: same quantile rule on the caller's map, whatever the policy
            return torch.from_numpy(lo.mask_point_based_std(cust_map.contiguous().numpy(), pr))
        if mask_pol == "point-based-std":
            return torch.from_numpy(lo.mask_point_based_std(scale.numpy(), pr))
        if mask_pol == "two-levels":
            return torch.zeros_like(scale) if pr == 0 else torch.ones_like(scale)
        if mask_pol == "three-levels-std":              # :229-247: quantile 0.8 == the point-based rule at pr = 2
            if pr == 0:
                return torch.zeros_like(scale)
            if pr == 2:
                return torch.ones_like(scale)
            return torch.from_numpy(lo.mask_point_based_std(scale.numpy(), 2.0))
        raise NotImplementedError(mask_pol)

    def _encode(self, sym, idx, tables):                # entropy_models.py:226-235 (one stream per image, C,H,W order)
        s, i = sym.numpy(), idx.numpy()
        return [lo.rans_encode(s[b], i[b], tables) for b in range(s.shape[0])]

    def _decode(self, strings, idx, tables):            # entropy_models.py:276-286
        i = idx.numpy()
        out = np.stack([lo.rans_decode(strings[b], i[b], tables).reshape(i[b].shape) for b in range(len(strings))])
        return torch.from_numpy(out)

    def _eb_indexes(self, B, h, w):                     # entropy_models.py:492-502
        C = self.medians.numel()
        return torch.arange(C, dtype=torch.int32).view(1, C, 1, 1).expand(B, C, h, w).contiguous()

    def _hyper(self, z_hat, quality):                   # CHProg_cnn.py:705-715 / :856-867
        ls = self.h_s("h_scale_s", 0, z_hat)
        lm = self.h_s("h_mean_s", 0, z_hat)
        if quality != 0:
            ls = torch.cat([ls, self.h_s("h_scale_s", 1, z_hat)], 1)
            lm = torch.cat([lm, self.h_s("h_mean_s", 1, z_hat)], 1)
        return lm, ls

    def _lrp(self, fam, i, mean_support, y_hat):        # CHProg_cnn.py:759-762
        lrp = self.stack5(fam, i, torch.cat([mean_support, y_hat], 1))
        return y_hat + 0.5 * self.ops.tanh(lrp)

    def _enh_support(self, base, enh, i):               # determine_support, CHProg_cnn.py:377-383
        return [base[i]] + (enh[i - min(MAX_SUPPORT, i):i] if i > 0 else [])

    # ---- compress / decompress
    def refine_scale(self, i, quality, mask_pol, y_b_hat, mu_base, std_base, mu, scale):
        """hook of the REM model (RemCodec below): the plain codec leaves the predicted parameters alone.  Returns (mu, scale)."""
        return mu, scale

    def compress(self, x, quality=0.0, mask_pol="point-based-std", taps=None, cust_map=None, force_enhanced=False):
        """ChannelProgresssiveWACNN.compress, models/CHProg_cnn.py:686-847.  force_enhanced (forward_single_quality only,
        :1006,1022,1064): at quality 0 still run both hyper-priors and the enhancement chain (all-zero masks)."""
        T = taps if taps is not None else {}
        y = self.g_a(x)                                                     # :692
        z = self.h_a(y)                                                     # :700
        B, _, zh, zw = z.shape
        med = self.medians.view(1, -1, 1, 1)
        z_sym = torch.from_numpy(lo.quantize(z.numpy(), med.expand_as(z).contiguous().numpy()))   # :702 -> entropy_models.py:508-515,212
        z_idx = self._eb_indexes(B, zh, zw)
        z_strings = self._encode(z_sym, z_idx, self.eb)
        z_hat = z_sym.float() + med                                         # :704 -> entropy_models.py:517-522,289
        lm, ls = self._hyper(z_hat, 1.0 if force_enhanced else quality)
        T.update(y=y, z=z, z_sym=z_sym, latent_means=lm, latent_scales=ls)
        y_slices = y.chunk(NS1, 1)
        y_strings, masks, base = [], [], []
        for i in range(NS0):                                                # :729-764
            sup = base[:min(MAX_SUPPORT, i)]
            mean_support = torch.cat([lm[:, :D0]] + sup, 1)
            scale_support = torch.cat([ls[:, :D0]] + sup, 1)
            mu = self.stack5("cc_mean_transforms", i, mean_support)
            scale = self.stack5("cc_scale_transforms", i, scale_support)
            idx = self._indexes(scale)                                      # :751
            sym = torch.from_numpy(lo.quantize(y_slices[i].numpy(), mu.numpy()))   # :752 -> entropy_models.py:212
            y_strings.append(self._encode(sym, idx, self.gc))
            y_hat = sym.float() + mu                                        # :754-755
            y_hat = self._lrp("lrp_transforms", i, mean_support, y_hat)
            base.append(y_hat)
            T[f"b{i}"] = dict(mu=mu, scale=scale, idx=idx, sym=sym, y_hat=y_hat)
        if quality <= 0 and not force_enhanced:                             # :766-767
            return {"strings": [y_strings, z_strings], "shape": torch.Size([zh, zw]), "masks": masks}
        enh = []
        cm = cust_map.chunk(NS0, 1) if cust_map is not None else None       # :721-722
        for i in range(NS0):                                                # :775-845
            y_slice = y_slices[NS0 + i] - y_slices[i]                       # delta_encode :780-781
            sup = self._enh_support(base, enh, i)
            mean_support = torch.cat([lm[:, D0:]] + sup, 1)
            scale_support = torch.cat([ls[:, D0:]] + sup, 1)
            mu = self.stack5("cc_mean_transforms_prog", i, mean_support)
            scale = self.stack5("cc_scale_transforms_prog", i, scale_support)
            mu, scale = self.refine_scale(i, quality, mask_pol, base[i], T[f"b{i}"]["mu"], T[f"b{i}"]["scale"], mu, scale)
            mask = self._mask(scale, quality, mask_pol, cm[i] if cm is not None else None)   # :819-824
            masks.append(mask)
            idx = self._indexes(scale * mask)                               # :828
            sym = torch.from_numpy(lo.quantize(((y_slice - mu) * mask).numpy()))   # :830
            y_strings.append(self._encode(sym, idx, self.gc))
            y_hat = sym.float() + mu                                        # :833-834
            y_hat = self._lrp("lrp_transforms_prog", i, mean_support, y_hat)
            y_hat = y_hat + base[i]                                         # merge "res" :843,385-387
            enh.append(y_hat)
            T[f"e{i}"] = dict(mu=mu, scale=scale, mask=mask, idx=idx, sym=sym, y_hat=y_hat)
        return {"strings": [y_strings, z_strings], "shape": torch.Size([zh, zw]), "masks": masks}

    # ---- likelihood path
    def _gc_likelihood(self, values, scale_eff):
        """GaussianConditional._likelihood (values = outputs - means, or outputs when no means are passed) + likelihood_lower_bound, entropy_models.py:626-659,
        :578-582 (half * erfc(const * x), const = -(2 ** -0.5), all in float32).  Back-end "torch": torch.erfc as the reference;
        "cdet": erfc in double on the float32 argument, rounded to float32 (what the HIP kernel does)."""
        values = values.abs()
        scales = torch.max(scale_eff, torch.tensor(self.scale_bound, dtype=torch.float32))       # lower_bound_scale
        cst = torch.tensor(-(2 ** -0.5), dtype=torch.float32)
        u, l = (0.5 - values) / scales, (-0.5 - values) / scales
        if isinstance(self.ops, TorchOps):
            up, low = 0.5 * torch.erfc(cst * u), 0.5 * torch.erfc(cst * l)
        else:
            from scipy.special import erfc
            e = lambda t: torch.from_numpy(erfc((cst * t).numpy().astype(np.float64)).astype(np.float32))
            up, low = 0.5 * e(u), 0.5 * e(l)
        return torch.max(up - low, torch.tensor(1e-9, dtype=torch.float32))

    def _eb_likelihood(self, z_hat):
        """EntropyBottleneck._likelihood + likelihood_lower_bound on the dequantised hyper-latent, entropy_models.py:400-433,:446-479."""
        B, C, h, w = z_hat.shape
        v = z_hat.permute(1, 0, 2, 3).reshape(C, 1, -1)

        def logits(x):
            for i in range(5):
                x = torch.matmul(F.softplus(self.sd[f"entropy_bottleneck._matrix{i}"]), x)
                x = x + self.sd[f"entropy_bottleneck._bias{i}"]
                if i < 4:
                    x = x + torch.tanh(self.sd[f"entropy_bottleneck._factor{i}"]) * torch.tanh(x)
            return x

        lower, upper = logits(v - 0.5), logits(v + 0.5)
        sign = -torch.sign(lower + upper)
        lik = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))
        lik = torch.max(lik, torch.tensor(1e-9, dtype=torch.float32))
        return lik.reshape(C, B, h, w).permute(1, 0, 2, 3).contiguous()

    def forward_single_quality(self, x, quality, mask_pol="point-based-std", force_enhanced=False):
        """ChannelProgresssiveWACNN.forward_single_quality in eval mode, models/CHProg_cnn.py:1002-1198: the chain of compress()
        (same mu / scale / mask / round / LRP per slice, :1033-1160) with likelihoods instead of entropy coding, then g_s."""
        T = {}
        out = self.compress(x, quality, mask_pol, taps=T, force_enhanced=force_enhanced)
        med = self.medians.view(1, -1, 1, 1)
        z_lik = self._eb_likelihood(T["z_sym"].float() + med)                       # compute_hyperprior :400
        liks, y_hat = [], []
        for i in range(NS0):                                                        # :1050
            t = T[f"b{i}"]
            outputs = t["sym"].float() + t["mu"]                                    # quantize(.., "dequantize", means), :137-139,:159-165
            liks.append(self._gc_likelihood(outputs - t["mu"], t["scale"]))         # values = inputs - means (float32: not exactly sym)
            y_hat.append(t["y_hat"])
        if quality == 0 and not force_enhanced:                                     # :1063-1080
            return {"x_hat": self.g_s(0, torch.cat(y_hat, 1)).clamp_(0, 1), "likelihoods": {"y": torch.cat(liks, 1), "z": z_lik}, "masks": []}
        y_hat = []
        for i in range(NS0):                                                        # :1150 (scale * block_mask, no means)
            t = T[f"e{i}"]
            liks.append(self._gc_likelihood(t["sym"].float(), t["scale"] * t["mask"]))
            y_hat.append(t["y_hat"])
        return {"x_hat": self.g_s(1, torch.cat(y_hat, 1)).clamp_(0, 1), "likelihoods": {"y": torch.cat(liks, 1), "z": z_lik}, "masks": out["masks"]}

    def decompress(self, strings, shape, quality, mask_pol="point-based-std", taps=None, cust_map=None):
        """ChannelProgresssiveWACNN.decompress, models/CHProg_cnn.py:849-999."""
        T = taps if taps is not None else {}
        y_strings, z_strings = strings
        B = len(z_strings)
        zh, zw = int(shape[0]), int(shape[1])
        med = self.medians.view(1, -1, 1, 1)
        z_sym = self._decode(z_strings, self._eb_indexes(B, zh, zw), self.eb)       # :855
        z_hat = z_sym.float() + med
        lm, ls = self._hyper(z_hat, quality)
        base, mu_b, std_b = [], [], []
        for i in range(NS0):                                                # :874-904
            sup = base[:min(MAX_SUPPORT, i)]
            mean_support = torch.cat([lm[:, :D0]] + sup, 1)
            scale_support = torch.cat([ls[:, :D0]] + sup, 1)
            mu = self.stack5("cc_mean_transforms", i, mean_support)
            scale = self.stack5("cc_scale_transforms", i, scale_support)
            mu_b.append(mu); std_b.append(scale)
            idx = self._indexes(scale)
            sym = self._decode(y_strings[i], idx, self.gc)                  # :894
            y_hat = sym.float() + mu                                        # :896
            base.append(self._lrp("lrp_transforms", i, mean_support, y_hat))
        if quality == 0:                                                    # :907-916
            x_hat = self.g_s(0, torch.cat(base, 1)).clamp_(0, 1)
            T.update(y_hat=torch.cat(base, 1))
            return {"x_hat": x_hat}
        enh = []
        cm = cust_map.chunk(NS0, 1) if cust_map is not None else None       # :850-851
        for i in range(NS0):                                                # :921-983
            sup = self._enh_support(base, enh, i)
            mean_support = torch.cat([lm[:, D0:]] + sup, 1)
            scale_support = torch.cat([ls[:, D0:]] + sup, 1)
            mu = self.stack5("cc_mean_transforms_prog", i, mean_support)
            scale = self.stack5("cc_scale_transforms_prog", i, scale_support)
            mu, scale = self.refine_scale(i, quality, mask_pol, base[i], mu_b[i], std_b[i], mu, scale)
            mask = self._mask(scale, quality, mask_pol, cm[i] if cm is not None else None)   # :960-965
            idx = self._indexes(scale * mask)                               # :968
            sym = self._decode(y_strings[NS0 + i], idx, self.gc)
            y_hat = sym.float() + mu
            y_hat = self._lrp("lrp_transforms_prog", i, mean_support, y_hat)
            enh.append(y_hat + base[i])
        y_hat = torch.cat(enh, 1)
        T.update(y_hat=y_hat)
        return {"x_hat": self.g_s(1, y_hat).clamp_(0, 1)}                   # :986-990


# ----------------------------------------------------------------------------- REM (models/CHProgREM.py)
class RemCodec(RefCodec):
    """PostRateProcessedNetwork.compress / .decompress (models/CHProgREM.py:673-888, :896-1126), real_compress=True: the base codec's
    chain, with the predicted scale -- and, with mu_std=True, the predicted mean too (:414-416) -- of every enhancement slice refined by a
    LatentRateReduction CNN (:12-86) before the mask is taken: apply_latent_enhancement (:375-428).  `post_sd`: the state dict of
    `post_latent` ("<level>.<slice>.<subnet>.<block>.conv1.weight" ...); the sub-net depth (dimension "big" / "middle") follows from
    its keys.  `checkpoint_rep` (set_checkpoint_rep): a [B,320,h,w] representation that replaces the decoded base slices as the CNN's
    x_base input (:773,989) -- what the escalation mode chains from level to level (:335-373)."""

    def __init__(self, state_dict, post_sd, backend="torch", check_levels=(0.01, 0.25, 1.75), mu_std=False, **kw):
        super().__init__(state_dict, backend, **kw)
        self.post = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.asarray(v))) for k, v in post_sd.items()}
        self.check_levels = list(check_levels)
        self.mu_std = mu_std
        self.checkpoint_rep = None

    def set_checkpoint_rep(self, rep):
        self.checkpoint_rep = rep

    def _rb(self, x, p):                                # ResidualBlock, models/utils.py:59-87
        w1, b1, w2, b2 = (self.post[f"{p}.{n}"] for n in ("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias"))
        out = self.ops.leaky(self.ops.conv(x, w1, b1, 1, 1))
        out = self.ops.leaky(self.ops.conv(out, w2, b2, 1, 1))
        ident = x
        if f"{p}.skip.weight" in self.post:
            ident = self.ops.conv(x, self.post[f"{p}.skip.weight"], self.post[f"{p}.skip.bias"], 1, 0)
        return out + ident

    def _seq(self, x, p):
        j = 0
        while f"{p}.{j}.conv1.weight" in self.post:
            x = self._rb(x, f"{p}.{j}")
            j += 1
        return x

    def find_check_quality(self, quality):              # :446-466
        c = self.check_levels
        if quality <= c[0]:
            return 0, 0
        if len(c) in (2, 3) and c[0] < quality <= c[1]:
            return c[0], c[1]
        if len(c) == 2 and quality > c[1]:
            return c[1], 10
        if len(c) == 3 and c[1] < quality <= c[2]:
            return c[1], c[-1]
        return c[-1], 10

    def refine_scale(self, i, quality, mask_pol, y_b_hat, mu_base, std_base, mu, scale):
        """apply_latent_enhancement, :375-428 (attention mask = star - bar from the UNREFINED scale, rounded; nothing below the first
        check level; the net of the quality's range), then LatentRateReduction.forward :74-86.  Returns (mu, scale)."""
        c = self.check_levels
        if quality <= c[0]:
            return mu, scale
        q_bar, _ = self.find_check_quality(quality)
        # the reference's call sites (:620,832,1060) do not forward mask_pol: the attention mask is always "point-based-std" (:385)
        att = torch.round(self._mask(scale, quality, "point-based-std") - self._mask(scale, q_bar, "point-based-std"))
        if len(c) == 1:
            k = 0
        elif len(c) == 2:
            k = 0 if c[0] < quality <= c[1] else 1
        else:
            k = 0 if c[0] < quality <= c[1] else (1 if c[1] < quality <= c[2] else 2)
        p = f"{k}.{i}"
        if self.checkpoint_rep is not None:                                 # :773,989
            y_b_hat = self.checkpoint_rep[:, 32 * i:32 * (i + 1)]
        ident = torch.cat([mu, scale], 1) if self.mu_std else scale         # :789 mu_scale_enh
        if self.mu_std:
            att = torch.cat([att, att], 1)                                  # :397-398
        f_ent_prog = self._seq(ident, p + ".enc_enh_entropy_params")
        f_latent = self._seq(y_b_hat, p + ".enc_base_rep")
        f_ent_base = self._seq(torch.cat([mu_base, std_base], 1), p + ".enc_base_entropy_params")
        ret = self._seq(torch.cat([f_latent, f_ent_base, f_ent_prog], 1), p + ".enc")
        out = ret * att + ident
        if self.mu_std:
            m2, s2 = out.chunk(2, 1)                                        # :414-416
            return m2.contiguous(), s2.contiguous()
        return mu, out


# ----------------------------------------------------------------------------- harness (training/step.py:277-404)
def compute_padding(h, w, min_div=64):
    """compressai.ops.compute_padding as used at training/step.py:318 (centre padding)."""
    H2, W2 = (h + min_div - 1) // min_div * min_div, (w + min_div - 1) // min_div * min_div
    l, t = (W2 - w) // 2, (H2 - h) // 2
    r, b = W2 - w - l, H2 - h - t
    return (l, r, t, b), (-l, -r, -t, -b)


def bpp_of(strings, B, h, w):
    """training/step.py:357-365 generalised to B images: 8 * total bytes / (B*h*w)."""
    y_strings, z_strings = strings
    n = sum(len(s) for sl in y_strings for s in sl) + sum(len(s) for s in z_strings)
    return 8.0 * n / (B * h * w)


def psnr_of(a, b):
    """training/step.py:13-18,349."""
    mse = torch.mean((a - b) ** 2).item()
    return -10.0 * math.log10(mse) if mse > 0 else float("inf")
