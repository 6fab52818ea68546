#!/usr/bin/env python3
"""Golden fixture of the REAL reference on bench.py's exact Config-2 batch (VERDICT r02 "Next round" 1a).

  x = torch.rand(32, 3, 256, 256, generator=torch.Generator().manual_seed(1)); quality 0.5; mask_pol "point-based-std";
  synthetic seeded weights (progressivecodec_amd.synth) -- the batch rank 0 of `python bench.py` codes.

Run once in the build container (imports the reference from /root/reference through tests/golden/ref_env.py):
    python3 tests/golden/make_golden_config2.py [threads ...]          (default: 8 and 1)

Output (data only): tests/golden/config2.json
  per thread count of the reference run: sha256 + length of every one of the 20 x 32 y strings and 32 z strings, mask sums
  per slice and image, per-image bpp and PSNR (step.py:349-365 applied per image), batch bpp / PSNR, sha256 of x_hat.
  The first entry (8 threads, the build container's core count) is THE golden; the other entries say what a different
  oneDNN thread team does to the same strings on the same machine (nothing, if the hashes are equal).
"""
import hashlib
import json
import math
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)

import ref_env  # noqa: E402

net = ref_env.canonical_model()
import torch  # noqa: E402

from progressivecodec_amd.synth import synthetic_state_dict  # noqa: E402

B, S, Q, SEED = 32, 256, 0.5, 1
net.load_state_dict(synthetic_state_dict())
net.update(force=True)
x = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(SEED))
sha = lambda b: hashlib.sha256(b).hexdigest()

threads = [int(a) for a in sys.argv[1:]] or [8, 1]
OUT = os.path.join(HERE, "config2.json")
runs = json.load(open(OUT))["runs"] if os.path.exists(OUT) else []       # thread counts already recorded are kept, new ones appended
for nt in threads:
    if any(r["threads"] == nt for r in runs):
        continue
    torch.set_num_threads(nt)
    t0 = time.perf_counter()
    with torch.no_grad():
        out = net.compress(x, quality=Q, mask_pol="point-based-std")
        t1 = time.perf_counter()
        dec = net.decompress(out["strings"], out["shape"], Q, mask_pol="point-based-std")
    t2 = time.perf_counter()
    x_hat = dec["x_hat"].clamp_(0, 1)                                         # step.py:343 (no padding at 256x256)
    ys, zs = out["strings"]
    per_img_bytes = [sum(len(ys[s][b]) for s in range(len(ys))) + len(zs[b]) for b in range(B)]
    runs.append(dict(
        threads=nt, torch=torch.__version__, enc_s=round(t1 - t0, 2), dec_s=round(t2 - t1, 2),
        shape=list(out["shape"]),
        y_sha=[[sha(s) for s in sl] for sl in ys], y_len=[[len(s) for s in sl] for sl in ys],
        z_sha=[sha(s) for s in zs], z_len=[len(s) for s in zs],
        mask_sums=[[int(m[b].sum().item()) for b in range(B)] for m in out["masks"]],
        bpp_per_image=[8.0 * n / (S * S) for n in per_img_bytes],
        psnr_per_image=[-10.0 * math.log10(torch.mean((x[b] - x_hat[b]) ** 2).item()) for b in range(B)],
        bpp=8.0 * sum(per_img_bytes) / (B * S * S),
        psnr=-10.0 * math.log10(torch.mean((x - x_hat) ** 2).item()),
        x_hat_sha=sha(x_hat.numpy().tobytes())))
    print(f"threads {nt}: enc {t1 - t0:.1f} s dec {t2 - t1:.1f} s bpp {runs[-1]['bpp']:.6f} psnr {runs[-1]['psnr']:.6f}", flush=True)
    json.dump(dict(config="Config 2", B=B, H=S, W=S, seed=SEED, quality=Q, mask_pol="point-based-std", runs=runs),
              open(OUT, "w"))
r0 = runs[0]
for r in runs[1:]:
    same_y = sum(a == b for sa, sb in zip(r0["y_sha"], r["y_sha"]) for a, b in zip(sa, sb))
    print(f"threads {r['threads']} vs {r0['threads']}: y strings identical {same_y}/{20 * B}, z {sum(a == b for a, b in zip(r0['z_sha'], r['z_sha']))}/{B}, "
          f"x_hat identical {r['x_hat_sha'] == r0['x_hat_sha']}")
print("done")
