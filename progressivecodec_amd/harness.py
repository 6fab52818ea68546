"""compress_with_ac -- the reference's end-to-end real-bitstream harness
(/root/reference/src/compress/training/step.py:277-404), restated for in-memory images.

Per image: centre zero-pad to a multiple of 64 (step.py:318-319, compressai.ops.compute_padding),
for every level `p` of `pr_list`: compress -> decompress (only the decode is timed, step.py:332-340),
un-pad and clamp (step.py:342-343), PSNR = -10 log10(mean((x - x_hat)^2)) (step.py:13-18,349),
bpp = 8 * (sum of the byte-string lengths) / (H*W) over the UNPADDED size (step.py:357-365).
File reading, wandb logging and the text dump of step.py are outside the hot path.

Return signature: the reference returns (bpp, psnr, mssim, dec_time) per level (step.py:404); this harness returns
(bpp, psnr, dec_time, rows) -- the MS-SSIM column (step.py:350-353, `pytorch_msssim.ms_ssim`) is NOT produced: that package is
absent from this image and from the GPU box, the reference holds no MS-SSIM fixture, so any restatement would be parity-unpinned.
Callers that need it can compute it from the x_hat of decompress() with their own MS-SSIM.
"""
import math
import time

#: the level list the authors evaluate (train.py:293)
PR_LIST = [0, 0.05, 0.1, 0.25, 0.5, 0.6, 0.75, 1, 1.25, 2, 3, 5, 10]


def config3_images(n=24, first_seed=100):
    """BASELINE.json Config 3 stand-in for the Kodak set (no Kodak files exist offline): `n` tensors [1,3,512,768]
    drawn with torch.rand from seeds first_seed .. first_seed+n-1; every fourth one (index 3, 7, 11, ...: six of 24, as many as Kodak has
    portrait images) is transposed to 768x512.  Shared by tests, tools/configs_bench.py and tests/golden/make_golden_config3.py."""
    import torch
    out = []
    for i in range(n):
        x = torch.rand(1, 3, 512, 768, generator=torch.Generator().manual_seed(first_seed + i))
        out.append(x.transpose(2, 3).contiguous() if i % 4 == 3 else x)
    return out


def compare_with_golden_strings(strings, y_sha, z_sha):
    """What separates byte strings made here from a committed fixture of the reference's (sha256 per string: tests/golden/config2.json,
    config3.json).  strings = [y_strings[slice][image], z_strings[image]] as compress() returns them; y_sha[slice][image], z_sha[image].
    Returns a dict: identical-string counts, per image the first diverging y slice (None: every y string identical; -1: the
    hyper-latent string already differs), the flip-free image list and the histogram of first diverging slices.  An image is
    "flip-free" when all of its strings equal the reference's; float rounding (the reference's oneDNN summation order vs the numeric
    contract's fmaf chain) that flips one round() or compare makes an image differ from that slice on."""
    import hashlib
    sha = lambda b: hashlib.sha256(b).hexdigest()
    ys, zs = strings
    B = len(zs)
    z_same = [sha(zs[b]) == z_sha[b] for b in range(B)]
    same = [[sha(ys[s][b]) == y_sha[s][b] for b in range(B)] for s in range(len(ys))]
    first = []
    for b in range(B):
        if not z_same[b]:
            first.append(-1)
            continue
        f = next((s for s in range(len(ys)) if not same[s][b]), None)
        first.append(f)
    hist = {}
    for f in first:
        if f is not None:
            hist[str(f)] = hist.get(str(f), 0) + 1
    return {"images": B, "z_strings_identical": sum(z_same), "y_strings_identical": sum(sum(r) for r in same), "y_strings": len(ys) * B,
            "first_diverging_slice": first, "flip_free_images": [b for b, f in enumerate(first) if f is None],
            "first_diverging_slice_histogram": dict(sorted(hist.items(), key=lambda kv: int(kv[0])))}


def compute_padding(in_h, in_w, min_div=64):
    out_h = (in_h + min_div - 1) // min_div * min_div
    out_w = (in_w + min_div - 1) // min_div * min_div
    left = (out_w - in_w) // 2
    right = out_w - in_w - left
    top = (out_h - in_h) // 2
    bottom = out_h - in_h - top
    return (left, right, top, bottom), (-left, -right, -top, -bottom)


def _pipeline_of(model):
    """The CodecPipeline behind compress_with_ac(overlap=True): `model` itself when it is one, else one built around `model` (its
    encoder) on first use and kept on it -- the decoder object is a second weights copy in HBM."""
    from .pipeline import CodecPipeline
    if isinstance(model, CodecPipeline):
        return model
    pipe = model.__dict__.get("_pipeline")
    if pipe is None:
        pipe = CodecPipeline.from_model(model)
        model.__dict__["_pipeline"] = pipe
    return pipe


def compress_with_ac(model, images, pr_list=None, mask_pol="point-based-std", device="cuda", shared_base=False, batch_same_size=False,
                     overlap=False, group_size=18):
    """images: iterable of [1,3,H,W] (or [3,H,W]) float tensors in [0,1].
    Returns (bpp[level], psnr[level], dec_time[level]) averaged over the images, as step.py:404 does,
    plus the per-image table.

    batch_same_size=True (implies shared_base; beyond the reference, which loops over single images, step.py:297): images of equal
    size are stacked and coded in one call per group -- an image codes identically alone and inside any batch (every image is its own
    set of rANS streams, entropy_models.py:227; asserted by the GPU tests), so the RD table is the same table, rows in input order; the
    slice chain then runs at M = n_images * H/16 * W/16 rows instead of one image's.  dec_time = group decode time / (images * levels).

    shared_base=True codes all levels of an image with model.compress_levels / decompress_levels: g_a, h_a, z, h_s and
    the ten base slices run once per image instead of once per level.  The RD table is
    identical; dec_time is then the time of the joint decode divided by the number of levels.

    overlap=True (implies batch_same_size and shared_base): the groups of equal-sized images are cut into jobs of at most `group_size`
    images (default 18: with two levels in flight inside every multi-level call -- round 4 -- large jobs are the efficient ones; the
    overlap then hides one job's decode behind the next job's encode) and run through a CodecPipeline (progressivecodec_amd/pipeline.py) -- the decode of job i beside the encode of job i+1, on
    an encoder and a decoder object, two streams, two host threads.  `model` may be a CodecPipeline, or a loaded model around which
    one is built on first use.  Same strings, same x_hat, same RD table; dec_time = the decode stream's time on the job / (images * levels)."""
    import torch
    import torch.nn.functional as F
    pr_list = list(PR_LIST if pr_list is None else pr_list)
    rows = []
    if overlap:
        pipe = _pipeline_of(model)
        imgs = [(x if x.dim() == 4 else x.unsqueeze(0)) for x in images]
        groups = {}
        for i, x in enumerate(imgs):
            groups.setdefault((x.shape[2], x.shape[3]), []).append(i)
        jobs = []
        for (h, w), idxs in groups.items():
            pad, unpad = compute_padding(h, w, 64)
            for j0 in range(0, len(idxs), max(1, int(group_size))):
                part = idxs[j0:j0 + max(1, int(group_size))]
                jobs.append({"idxs": part, "hw": (h, w), "unpad": unpad, "pad": pad, "qualities": pr_list, "mask_pol": mask_pol, "time_decode": True})

        def feed():                                                     # inputs are moved and padded just before their encode is issued
            for job in jobs:
                xb = torch.cat([imgs[i] for i in job["idxs"]], 0).to(pipe.device)
                job["xb"] = xb
                job["x"] = F.pad(xb, job["pad"], mode="constant", value=0)
                yield job
        by_image = {}
        with torch.no_grad():
            for job, datas, outs in pipe.code(feed()):
                h, w = job["hw"]
                xb, idxs = job.pop("xb"), job["idxs"]
                job.pop("x")
                dec_time = 1e-3 * pipe.decode_ms(job) / (len(pr_list) * len(idxs))
                for p, data, out_dec in zip(pr_list, datas, outs):
                    x_hat = F.pad(out_dec["x_hat"], job["unpad"]).clamp_(0, 1)
                    y_strings, z_strings = data["strings"]
                    mses = torch.mean((xb - x_hat) ** 2, dim=(1, 2, 3)).tolist()
                    for b, i in enumerate(idxs):
                        nbytes = sum(len(s[b]) for s in y_strings) + len(z_strings[b])
                        by_image.setdefault(i, []).append({"quality": p, "bpp": 8.0 * nbytes / (h * w),
                                                           "psnr": -10.0 * math.log10(mses[b]) if mses[b] > 0 else float("inf"), "dec_time": dec_time})
        for i in range(len(imgs)):
            rows.extend(by_image[i])
        n_img = max(1, len(imgs))
        avg = lambda key, p: sum(r[key] for r in rows if r["quality"] == p) / n_img
        return ([avg("bpp", p) for p in pr_list], [avg("psnr", p) for p in pr_list], [avg("dec_time", p) for p in pr_list], rows)
    if batch_same_size:
        imgs = [(x if x.dim() == 4 else x.unsqueeze(0)) for x in images]
        groups = {}
        for i, x in enumerate(imgs):
            groups.setdefault((x.shape[2], x.shape[3]), []).append(i)
        by_image = {}
        with torch.no_grad():
            for (h, w), idxs in groups.items():
                xb = torch.cat([imgs[i] for i in idxs], 0).to(device)
                pad, unpad = compute_padding(h, w, 64)
                xp = F.pad(xb, pad, mode="constant", value=0)
                datas = model.compress_levels(xp, pr_list, mask_pol=mask_pol)
                if xb.is_cuda:
                    torch.cuda.synchronize()
                t0 = time.time()
                outs = model.decompress_levels([d["strings"] for d in datas], datas[0]["shape"], pr_list, mask_pol=mask_pol)
                if xb.is_cuda:
                    torch.cuda.synchronize()
                dec_time = (time.time() - t0) / (len(pr_list) * len(idxs))
                for p, data, out_dec in zip(pr_list, datas, outs):
                    x_hat = F.pad(out_dec["x_hat"], unpad).clamp_(0, 1)
                    y_strings, z_strings = data["strings"]
                    mses = torch.mean((xb - x_hat) ** 2, dim=(1, 2, 3)).tolist()          # one reduction and one read-back per level, not per image
                    for b, i in enumerate(idxs):
                        mse = mses[b]
                        nbytes = sum(len(s[b]) for s in y_strings) + len(z_strings[b])
                        by_image.setdefault(i, []).append({"quality": p, "bpp": 8.0 * nbytes / (h * w),
                                                           "psnr": -10.0 * math.log10(mse) if mse > 0 else float("inf"), "dec_time": dec_time})
        for i in range(len(imgs)):
            rows.extend(by_image[i])
        n_img = max(1, len(imgs))
        avg = lambda key, p: sum(r[key] for r in rows if r["quality"] == p) / n_img
        return ([avg("bpp", p) for p in pr_list], [avg("psnr", p) for p in pr_list], [avg("dec_time", p) for p in pr_list], rows)
    with torch.no_grad():
        for x in images:
            x = x if x.dim() == 4 else x.unsqueeze(0)
            x = x.to(device)
            h, w = x.shape[2:]
            pad, unpad = compute_padding(h, w, 64)
            x_padded = F.pad(x, pad, mode="constant", value=0)
            if shared_base:
                datas = model.compress_levels(x_padded, pr_list, mask_pol=mask_pol)
                if x.is_cuda:
                    torch.cuda.synchronize()
                t0 = time.time()
                outs = model.decompress_levels([d["strings"] for d in datas], datas[0]["shape"], pr_list, mask_pol=mask_pol)
                if x.is_cuda:
                    torch.cuda.synchronize()
                dec_time = (time.time() - t0) / len(pr_list)
                for p, data, out_dec in zip(pr_list, datas, outs):
                    x_hat = F.pad(out_dec["x_hat"], unpad).clamp_(0, 1)
                    mse = torch.mean((x - x_hat) ** 2).item()
                    psnr = -10.0 * math.log10(mse) if mse > 0 else float("inf")
                    y_strings, z_strings = data["strings"]
                    nbytes = sum(len(s[0]) for s in y_strings) + sum(len(s) for s in z_strings)
                    rows.append({"quality": p, "bpp": 8.0 * nbytes / (h * w), "psnr": psnr, "dec_time": dec_time})
                continue
            for p in pr_list:
                data = model.compress(x_padded, quality=p, mask_pol=mask_pol)
                if x.is_cuda:
                    torch.cuda.synchronize()
                t0 = time.time()
                out_dec = model.decompress(data["strings"], data["shape"], quality=p, mask_pol=mask_pol)
                if x.is_cuda:
                    torch.cuda.synchronize()
                dec_time = time.time() - t0
                x_hat = F.pad(out_dec["x_hat"], unpad).clamp_(0, 1)
                mse = torch.mean((x - x_hat) ** 2).item()
                psnr = -10.0 * math.log10(mse) if mse > 0 else float("inf")
                y_strings, z_strings = data["strings"]
                nbytes = sum(len(s[0]) for s in y_strings) + sum(len(s) for s in z_strings)   # step.py:357-365 (B = 1)
                rows.append({"quality": p, "bpp": 8.0 * nbytes / (h * w), "psnr": psnr, "dec_time": dec_time})
    n_img = max(1, len(rows) // max(1, len(pr_list)))
    avg = lambda key, p: sum(r[key] for r in rows if r["quality"] == p) / n_img
    return ([avg("bpp", p) for p in pr_list], [avg("psnr", p) for p in pr_list], [avg("dec_time", p) for p in pr_list], rows)


def estimate_rd(model, images, pr_list=None, mask_pol="point-based-std", device="cuda"):
    """The likelihood-based RD table of test_epoch (training/step.py:215-267): per level, bpp = sum(log(likelihoods)) /
    (-log(2) * pixels) (training/loss.py, the 'bpp' term) and PSNR of forward_single_quality's x_hat, averaged over the images.
    No entropy coding runs.  Pixel count = the padded size, as the reference's criterion sees the padded tensor."""
    import torch
    import torch.nn.functional as F
    pr_list = list(PR_LIST if pr_list is None else pr_list)
    rows = []
    with torch.no_grad():
        for x in images:
            x = x if x.dim() == 4 else x.unsqueeze(0)
            x = x.to(device)
            h, w = x.shape[2:]
            pad, unpad = compute_padding(h, w, 64)
            xp = F.pad(x, pad, mode="constant", value=0)
            n_pix = xp.shape[0] * xp.shape[2] * xp.shape[3]
            for p in pr_list:
                out = model.forward_single_quality(xp, quality=p, mask_pol=mask_pol)
                bpp = sum(torch.log(l.double()).sum().item() for l in out["likelihoods"].values()) / (-math.log(2) * n_pix)
                x_hat = F.pad(out["x_hat"], unpad).clamp_(0, 1)
                mse = torch.mean((x - x_hat) ** 2).item()
                rows.append({"quality": p, "bpp": bpp, "psnr": -10.0 * math.log10(mse) if mse > 0 else float("inf")})
    n_img = max(1, len(rows) // max(1, len(pr_list)))
    avg = lambda key, p: sum(r[key] for r in rows if r["quality"] == p) / n_img
    return [avg("bpp", p) for p in pr_list], [avg("psnr", p) for p in pr_list], rows
