#!/usr/bin/env python3
"""bench.py -- megapixels/s, encode + decode end to end, on BASELINE.json's Config 2
(batch of 32 random 256x256 crops per GPU, one mask level q=0.5, mask_pol "point-based-std").

A "step" is one pass of the hot path over one batch: ChannelProgresssiveWACNN.compress()
followed by .decompress() through the C ABI of libpcodec.so, with the input batch already
resident in HBM; the timed region covers device compute, entropy coding and producing the
host-visible byte strings (training/step.py:322-340 without file I/O).

  python bench.py --gpus N --steps K --warmup W
N > 1: launched by torchrun, one rank per GPU; images shard per rank (weak scaling), no
data-path collective; the only collective is the final gather of string lengths (outside
the timed region) -- see DESIGN.md "multi-GPU".

Rank 0 prints ONE JSON line with `roofline` (the MFMA convolution kernel family, HIP-event
timed, against the 157.3 TFLOP/s dense f32 matrix peak) and `cpu_baseline` (the CPU oracle
port with ATen CPU ops, on a bounded sample of the same workload).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFLOP_PER_PX_Q = 2.3251      # enc+dec, q > 0  (SURVEY.md section 8d / BASELINE.md section 3)
PEAK_F32_MFMA = 157.3        # TFLOP/s, MI355X_MICROARCH.md "Peak FP32 (matrix)"


def host_cores(cap=16):
    """CPU threads this process may really use: affinity mask, cgroup quota, and the GPU box's
    per-GPU CPU share (16) -- oversubscribed OpenMP teams spin and look like a hang."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU (Config 2: 32)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--quality", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-images", type=int, default=32, help="images of rank 0's batch the CPU port codes (x2 repetitions)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")

    from progressivecodec_amd import ChannelProgresssiveWACNN
    from progressivecodec_amd._lib import check, lib
    from progressivecodec_amd.synth import synthetic_state_dict

    print(f"[bench] rank {rank}: generating synthetic weights", file=sys.stderr, flush=True)
    sd = synthetic_state_dict()
    net = ChannelProgresssiveWACNN(device=str(dev))
    net.load_state_dict(sd)
    net.update()

    B, S, q = args.batch, args.size, args.quality
    g = torch.Generator().manual_seed(1 + rank)
    x = torch.rand(B, 3, S, S, generator=g).to(dev)          # resident in HBM before the timed region

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    def step():
        out = net.compress(x, q, "point-based-std")
        dec = net.decompress(out["strings"], out["shape"], q, "point-based-std")
        return out, dec

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    log(f"weights loaded, tables built; batch {B}x3x{S}x{S} resident; warmup x{args.warmup}")
    for i in range(args.warmup):
        tw = time.perf_counter()
        out, dec = step()
        torch.cuda.synchronize(dev)
        log(f"warmup step {i}: {time.perf_counter() - tw:.3f} s")
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, dec = step()
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    elapsed = t1 - t0
    log(f"timed {args.steps} steps: {elapsed:.3f} s")
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    barrier()

    # encode / decode split (informational; one extra step)
    torch.cuda.synchronize(dev)
    ta = time.perf_counter()
    out = net.compress(x, q, "point-based-std")
    torch.cuda.synchronize(dev)
    tb = time.perf_counter()
    dec = net.decompress(out["strings"], out["shape"], q, "point-based-std")
    torch.cuda.synchronize(dev)
    tc = time.perf_counter()

    nbytes = sum(len(s) for sl in out["strings"][0] for s in sl) + sum(len(s) for s in out["strings"][1])
    bpp = 8.0 * nbytes / (B * S * S)
    psnr = -10.0 * torch.log10(torch.mean((x - dec["x_hat"]) ** 2)).item()
    if world > 1:   # the final (tiny) gather: total coded bytes of the job
        tt = torch.tensor([nbytes], device=dev, dtype=torch.int64)
        dist.all_reduce(tt)
        total_bytes = int(tt.item())
    else:
        total_bytes = nbytes

    # ---- roofline leg: every launch of the MFMA conv family in one more step, HIP-event timed on its stream
    # (profiling forces the chain onto the caller's stream, one lane, so that a launch's duration is its own)
    h = net._h
    check(lib().pc_codec_profile_begin(h))
    step()
    nl, ms, fl = C.c_int64(), C.c_double(), C.c_double()
    check(lib().pc_codec_profile_end(h, C.byref(nl), C.byref(ms), C.byref(fl)))
    achieved = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
    # HBM bytes per launch cannot be read inside this process: they come from the committed rocprofv3 PMC passes of this
    # same command (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes; profiles/r*_hbm_traffic.json, folded by tools/pmc_traffic.py), or null
    traffic, traffic_src = None, None
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic.json")))      # newest round / letter last
    tj = cands[-1] if cands else ""
    if tj and B == 32 and S == 256:
        fam = json.load(open(tj))["families"]
        n = sum(f["launches_per_2_steps"] for k, f in fam.items() if k.startswith("conv_igemm"))
        traffic = round(sum(f["launches_per_2_steps"] * f["hbm_bytes_per_launch"] for k, f in fam.items() if k.startswith("conv_igemm")) / n)
        traffic_src = f"profiles/{os.path.basename(tj)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)"
    roofline = {"bound": "mfma", "kernel": "conv_igemm_dma_kernel + conv_igemm_kernel (f32 MFMA 32x32x2 implicit GEMM family)",
                "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_F32_MFMA, 4), "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": None,
                "launches_per_step": int(nl.value), "kernel_ms_per_step": round(ms.value, 3),
                "algorithmic_gflop_per_step": round(fl.value / 1e9, 2),
                "avg_launch_us": round(1e3 * ms.value / max(1, nl.value), 2)}

    mp = world * B * S * S * args.steps / 1e6
    value = mp / elapsed
    line = {
        "metric": "megapixels/s encode+decode (256x256 batches)", "value": round(value, 3), "unit": "MP/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"Config 2: batch {B} x {S}x{S} random crops per GPU, quality {q}, mask_pol point-based-std, "
                               "synthetic seeded weights (canonical ChannelProgresssiveWACNN)",
                   "images_per_gpu": B, "height": S, "width": S, "quality": q, "sharding": f"images x {world} ranks"},
        "enc_ms": round(1e3 * (tb - ta), 2), "dec_ms": round(1e3 * (tc - tb), 2),
        "bpp": round(bpp, 4), "psnr_db": round(psnr, 4), "coded_bytes_job": total_bytes,
        "path_frac_of_f32_mfma_peak": round(value * 1e6 * MFLOP_PER_PX_Q * 1e6 / world / (PEAK_F32_MFMA * 1e12), 4),
        "roofline": roofline,
    }

    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # rank 0 at N = 1 only (bench contract)
        # CPU baseline: the oracle port (same ATen CPU op sequence as the reference, bit-identical strings on one
        # machine -- tests/test_oracle_vs_golden.py), on a bounded sample of the same workload.
        from oracle.codec_ref import RefCodec
        log("cpu_baseline leg (oracle port, ATen CPU ops)")
        cores = host_cores()
        torch.set_num_threads(cores)
        orc = RefCodec(sd, "torch")
        orc.update()
        n_img = max(1, args.cpu_images)
        xc = x[:n_img].cpu()
        reps = 2
        t0 = time.perf_counter()
        with torch.no_grad():
            for _ in range(reps):
                o = orc.compress(xc, q)
                d = orc.decompress(o["strings"], o["shape"], q)
        dt = time.perf_counter() - t0
        pairs = [(a[i], b[i]) for a, b in zip(out["strings"][0], o["strings"][0]) for i in range(n_img)]
        same = sum(a == b for a, b in pairs)
        line["cpu_baseline"] = {"value": round(reps * n_img * S * S / 1e6 / dt, 4), "unit": "MP/s", "cores": cores, "kind": "port",
                                "sample": f"{n_img} of the {B} images of rank 0 ({S}x{S}, q={q}) as one batch, encode+decode x{reps}, "
                                          f"torch {torch.__version__} CPU ops + C rANS, {dt:.1f} s",
                                "y_strings_identical_to_gpu": f"{same}/{len(pairs)}"}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
