#!/usr/bin/env python3
"""bench.py -- megapixels/s, encode + decode end to end, on BASELINE.json's Config 2
(batch of 32 random 256x256 crops per GPU, one mask level q=0.5, mask_pol "point-based-std").

A "step" is one pass of the hot path over one batch: ChannelProgresssiveWACNN.compress()
followed by .decompress() through the C ABI of libpcodec.so, with the input batch already
resident in HBM; the timed region covers device compute, entropy coding and producing the
host-visible byte strings (training/step.py:322-340 without file I/O).

  python bench.py --gpus N --steps K --warmup W

N ranks, one per GPU.  Under torchrun (RANK / WORLD_SIZE in the environment) this process IS one
rank; started bare with --gpus N > 1 it launches the N ranks itself as child processes -- before
anything touches the GPU -- and fails loudly if the node has fewer than N devices.  Images shard
per rank (weak scaling: 32 images per GPU), weights replicate, there is no data-path collective;
after the timed region the variable-length byte strings of every rank are gathered over RCCL
(`parallel.gather_bitstreams`, two all-gathers over xGMI) and checked -- DESIGN.md "multi-GPU".

Rank 0 prints ONE JSON line with `roofline` (the MFMA convolution kernel family, HIP-event
timed, against the 157.3 TFLOP/s dense f32 matrix peak) and, at N = 1, `cpu_baseline` (the CPU
oracle port with ATen CPU ops, on a bounded sample of the same workload, with its bpp / PSNR and
the symbol-level differences between it and the GPU).
"""
import argparse
import ctypes as C
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFLOP_PER_PX_Q = 2.3251      # enc+dec, q > 0  (SURVEY.md section 8d / BASELINE.md section 3)
PEAK_F32_MFMA = 157.3        # TFLOP/s, MI355X_MICROARCH.md "Peak FP32 (matrix)"
MASK_POL = "point-based-std"


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


# ----------------------------------------------------------------------------- multi-rank plumbing (GPU-free: tests/test_parallel.py)
def launch_ranks(n, argv):
    """Start `n` ranks of this script as child processes (one per GPU) and relay rank 0's JSON line.  Runs BEFORE any GPU call of
    this process (a process that has initialised the GPU must not exec / fork GPU workers).  Returns the exit code."""
    import socket

    from progressivecodec_amd.parallel import count_gpus_without_hip
    # this process goes on to start the ranks: it must never initialise HIP / HSA itself (torch.cuda.device_count() can, on ROCm) --
    # the count comes from the KFD topology in sysfs, or from a short-lived child (tests/test_parallel.py)
    have = count_gpus_without_hip()
    if have < n:
        print(f"[bench] --gpus {n} asked for, this node exposes {have} GPU(s): refusing to run a smaller job under that label "
              f"(launch under torchrun on a node with {n} GPUs)", file=sys.stderr, flush=True)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


def job_plan(world, rank, images_per_gpu):
    """Weak scaling: the job is world * images_per_gpu images, rank r codes the contiguous shard parallel.shard_range gives it."""
    from progressivecodec_amd.parallel import shard_range
    total = world * images_per_gpu
    b, e = shard_range(total, rank, world)
    return total, b, e


def final_gather(strings, world, rank, images_per_gpu, group=None, device=None):
    """The job's only collective, after the timed region: every rank's y and z strings to every rank over the process group's
    backend (nccl = RCCL over xGMI on the GPU box, gloo in the CPU tests); checks the gathered lists against this rank's own
    strings and the plan.  Returns (total coded bytes of the job, number of images gathered)."""
    from progressivecodec_amd.parallel import gather_bitstreams
    y_strings, z_strings = strings
    total, b, e = job_plan(world, rank, images_per_gpu)
    all_y = gather_bitstreams(y_strings, group=group, device=device)
    all_z = gather_bitstreams([z_strings], group=group, device=device)[0]
    if len(all_z) != total or any(len(sl) != total for sl in all_y) or len(all_y) != len(y_strings):
        raise RuntimeError(f"bitstream gather: expected {len(y_strings)} slices x {total} images, got {len(all_y)} x {[len(s) for s in all_y][:3]}...")
    if all_z[b:e] != list(z_strings) or any(sl[b:e] != list(mine) for sl, mine in zip(all_y, y_strings)):
        raise RuntimeError("bitstream gather: this rank's strings are not where the shard plan puts them")
    return sum(len(s) for sl in all_y for s in sl) + sum(len(s) for s in all_z), total


def source_hash():
    """sha256 over the kernel / runtime sources: ties a profile under profiles/ to the build it was measured on (there is no .git
    on the GPU box)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "progressivecodec_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "progressivecodec_amd", "csrc", "*.cpp")) +
                    glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def newest_profile(pattern, src):
    """newest profiles/<pattern> whose recorded source hash is this build's; (None, reason) otherwise"""
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    for f in reversed(cands):
        try:
            j = json.load(open(f))
        except (OSError, ValueError):
            continue
        if j.get("source_hash") == src:
            return j, os.path.basename(f)
    return None, (f"no profiles/{pattern} carries this build's source hash {src} (newest: {os.path.basename(cands[-1])})" if cands
                  else f"no profiles/{pattern}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU (Config 2: 32)")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--quality", type=float, default=0.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lean", action="store_true", help="profiling runs: only the timed steps, the enc/dec split and the roofline leg -- no CPU baseline, "
                    "no rANS leg (it codes a 4K frame), no second (sequential) timed run; the kernel trace then holds the headline schedule only")
    ap.add_argument("--overlap", type=int, default=1, help="1 (default): the decode of step i runs beside the encode of step i+1 -- an encoder and a "
                    "decoder codec object on their own streams and host threads; 0: compress() then decompress(), one after the other")
    ap.add_argument("--queue-depth", type=int, default=2, help="items the encoder may run ahead of the decoder (CodecPipeline)")
    ap.add_argument("--pairs", type=int, default=2, help="encoder / decoder pairs of the CodecPipeline (n_pairs): 1 keeps ~2.4 conv kernels in flight, "
                    "2 (default) ~4.5 -- each pair is two more weights copies and its workspaces")
    ap.add_argument("--serial-schedule", action="store_true", help="pc_codec_set_option serial_schedule on every codec object: one lane, one stream, no "
                    "chain pipelining inside the codec -- with --overlap 0 the form in which a launch's duration is its own (rocprofv3 / PMC passes)")
    ap.add_argument("--cpu-images", type=int, default=32, help="images of rank 0's batch the CPU port codes (x2 repetitions)")
    args = ap.parse_args()
    if args.lean:
        args.no_cpu_baseline = True

    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not under_launcher and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but the launcher started {world} rank(s): the two must agree", file=sys.stderr, flush=True)
        sys.exit(2)
    os.environ.setdefault("LOCAL_WORLD_SIZE", str(world))    # one node: the host entropy-coding pool takes 1 / world of the CPUs, pinned
    # (importing progressivecodec_amd asks the HIP runtime for 16 hardware queues before HIP starts -- pipeline.request_hw_queues: the
    # overlapped schedule keeps ~20 streams busy and is bimodal on the default 4, profiles/r02_n_hw_queues.log)
    import progressivecodec_amd  # noqa: F401

    import torch
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if local_rank >= torch.cuda.device_count():                 # every rank checks its own device (the launcher only counted, GPU-free)
        print(f"[bench] rank {rank}: LOCAL_RANK {local_rank} but this process sees {torch.cuda.device_count()} GPU(s)", file=sys.stderr, flush=True)
        sys.exit(2)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    torch.cuda.set_device(local_rank)
    dev = torch.device(f"cuda:{local_rank}")

    from progressivecodec_amd import ChannelProgresssiveWACNN, CodecPipeline
    from progressivecodec_amd._lib import check, lib
    from progressivecodec_amd.synth import synthetic_state_dict

    print(f"[bench] rank {rank}: generating synthetic weights", file=sys.stderr, flush=True)
    sd = synthetic_state_dict()
    # The schedule is the LIBRARY's: progressivecodec_amd.CodecPipeline owns the encoder and the decoder object, their
    # streams and the decoder host thread; bench.py only feeds it jobs.  --overlap 0: one plain model object, calls in turn.
    pipe = None
    if args.overlap:
        pipe = CodecPipeline(sd, device=str(dev), queue_depth=args.queue_depth, n_pairs=args.pairs)
        net = pipe.enc
    else:
        net = ChannelProgresssiveWACNN(device=str(dev))
        net.load_state_dict(sd)
        net.update()
    if args.serial_schedule:
        for o in (pipe.objects if pipe else [net]):
            o.set_option("serial_schedule", 1)

    B, S, q = args.batch, args.size, args.quality
    g = torch.Generator().manual_seed(1 + rank)
    x = torch.rand(B, 3, S, S, generator=g).to(dev)          # resident in HBM before the timed region

    def log(msg):
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    def step():
        out = net.compress(x, q, MASK_POL)
        dec = net.decompress(out["strings"], out["shape"], q, MASK_POL)
        return out, dec

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    def jobs(n):
        return ({"x": x, "quality": q, "mask_pol": MASK_POL} for _ in range(n))

    def run_steps(n):
        """n steps; every step is one compress() and one decompress() of the batch.  Overlapped: through CodecPipeline.code() -- the decode
        of step i beside the encode of step i+1; else the calls in turn on one object."""
        res = None
        if pipe is None:
            for _ in range(n):
                res = step()
            return res
        for _, o, d in pipe.code(jobs(n)):
            res = (o, d)
        return res

    log(f"weights loaded, tables built; batch {B}x3x{S}x{S} resident; warmup x{args.warmup}")
    if pipe is not None and args.pairs > 1 and args.warmup > 0:
        # the W warm-up steps as ONE pipelined run, so that every pair's objects (workspaces, row tables) are exercised before the timed region
        tw = time.perf_counter()
        out, dec = run_steps(max(args.warmup, args.pairs))
        torch.cuda.synchronize(dev)
        log(f"warmup: {max(args.warmup, args.pairs)} pipelined steps over {args.pairs} pairs: {time.perf_counter() - tw:.3f} s")
    else:
        for i in range(args.warmup):
            tw = time.perf_counter()
            out, dec = run_steps(1)
            torch.cuda.synchronize(dev)
            log(f"warmup step {i}: {time.perf_counter() - tw:.3f} s")
    barrier()
    t0 = time.perf_counter()
    out, dec = run_steps(args.steps)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    elapsed = t1 - t0
    log(f"timed {args.steps} steps: {elapsed:.3f} s")
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    barrier()

    # per-rank spread (a straggler must be visible the day the scaling curve is run): min / max over ranks of this rank's own time
    rank_ms = 1e3 * (t1 - t0) / args.steps
    rank_ms_min = rank_ms_max = rank_ms
    if world > 1:
        t = torch.tensor([rank_ms, -rank_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rank_ms_max, rank_ms_min = float(t[0].item()), -float(t[1].item())

    # the same K steps strictly one after the other on ONE codec object (what a drop-in compress_with_ac caller sees): `sequential_value`
    seq_value = None
    if args.overlap and not args.lean:
        barrier()
        ts0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize(dev)
        seq_elapsed = time.perf_counter() - ts0
        if world > 1:
            t = torch.tensor([seq_elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            seq_elapsed = float(t.item())
        seq_value = world * B * S * S * args.steps / 1e6 / seq_elapsed
        log(f"sequential {args.steps} steps: {seq_elapsed:.3f} s")

    # encode / decode split (informational; one extra step)
    torch.cuda.synchronize(dev)
    ta = time.perf_counter()
    out = net.compress(x, q, MASK_POL)
    torch.cuda.synchronize(dev)
    tb = time.perf_counter()
    dec = net.decompress(out["strings"], out["shape"], q, MASK_POL)
    torch.cuda.synchronize(dev)
    tc = time.perf_counter()

    nbytes = sum(len(s) for sl in out["strings"][0] for s in sl) + sum(len(s) for s in out["strings"][1])
    bpp = 8.0 * nbytes / (B * S * S)
    psnr = -10.0 * torch.log10(torch.mean((x - dec["x_hat"]) ** 2)).item()
    gather = None
    if world > 1:   # the job's only collective: the variable-length strings of all ranks, over RCCL / xGMI, outside the timed region
        tg = time.perf_counter()
        total_bytes, n_img = final_gather(out["strings"], world, rank, B, device=dev)
        torch.cuda.synchronize(dev)
        gather = {"backend": "nccl (RCCL over xGMI)", "images": n_img, "bytes": total_bytes, "ms": round(1e3 * (time.perf_counter() - tg), 3),
                  "checked": "every rank holds all strings; own shard in place"}
    else:
        total_bytes = nbytes

    # ---- roofline leg (dominant kernel: the f32-MFMA conv family), HIP events on the streams the kernels are launched on.
    # (1) launch by launch: one more step with profiling forcing the chain onto the caller's stream in one lane, so that a launch's
    #     duration is its own; achieved = sum 2MNK / sum of the launch durations.
    h = net._h
    check(lib().pc_codec_profile_begin(h))
    step()
    nl, ms, fl, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
    check(lib().pc_codec_profile_end(h, C.byref(nl), C.byref(ms), C.byref(fl)))
    check(lib().pc_codec_profile_bytes(h, C.byref(by)))
    lbl_achieved = fl.value / (ms.value * 1e-3) / 1e12 if ms.value > 0 else 0.0
    # (2) IN the schedule `value` was timed on (VERDICT r03 item 1 / "What's weak" 4): the same K steps through CodecPipeline once more with
    #     every conv launch of both objects bracketed and the schedule left alone; the two objects keep ~2.5 conv kernels in flight, so a
    #     launch's own duration means little -- achieved = FLOPs / time during which at least one conv kernel runs (the union of the
    #     launch intervals on one timeline).  The brackets cost a few us per launch: the pass's own ms/step is reported beside it.
    insitu = None
    if pipe is not None:
        barrier()
        ti = time.perf_counter()
        pr = pipe.profile_conv_in_schedule(jobs(args.steps))
        ti = time.perf_counter() - ti
        if pr["busy_ms"] > 0:
            insitu = {"achieved": round(pr["algorithmic_flops"] / (pr["busy_ms"] * 1e-3) / 1e12, 2),
                      "conv_busy_ms_per_step": round(pr["busy_ms"] / args.steps, 3), "conv_busy_frac_of_window": round(pr["busy_ms"] / pr["window_ms"], 4),
                      "sum_of_launch_ms_per_step": round(pr["sum_ms"] / args.steps, 3), "mean_conv_kernels_in_flight": round(pr["mean_in_flight"], 2),
                      "launches": pr["launches"], "steps": args.steps, "instrumented_ms_per_step": round(1e3 * ti / args.steps, 3),
                      "uninstrumented_ms_per_step": round(1e3 * elapsed / args.steps, 3)}
    # HBM bytes per launch cannot be read inside this process: they come from the committed rocprofv3 PMC passes of this same command
    # (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes; profiles/r*_hbm_traffic.json, folded by tools/pmc_traffic.py) -- and only
    # from a file measured on THIS build (source hash), else null
    src = source_hash()
    traffic, traffic_src = None, None
    tj, why = newest_profile("r*_hbm_traffic.json", src)
    if tj is not None and B == 32 and S == 256:
        fam = tj["families"]
        n = sum(f["launches_per_2_steps"] for k, f in fam.items() if k.startswith("conv_igemm"))
        traffic = round(sum(f["launches_per_2_steps"] * f["hbm_bytes_per_launch"] for k, f in fam.items() if k.startswith("conv_igemm")) / n)
        traffic_src = f"profiles/{why} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on this build)"
    else:
        traffic_src = why if tj is None else "traffic profile is of the default workload only"
    alg_bytes = by.value / max(1, nl.value)
    achieved = insitu["achieved"] if insitu else lbl_achieved
    roofline = {"bound": "mfma", "kernel": "conv_igemm_uni_kernel + conv_igemm_in_gdn_kernel + conv_igemm_kernel (f32 MFMA 32x32x2 implicit GEMM family)",
                "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_F32_MFMA, 4),
                "measured_on": ("the timed schedule (CodecPipeline), second pass of the same K steps with every conv launch of all its codec "
                                "objects bracketed by HIP events on its own stream: FLOPs / time during which >= 1 conv kernel runs") if insitu else
                               "one step, launch by launch (serial schedule): sum 2MNK / sum of HIP-event launch durations",
                "in_schedule": insitu,
                "launch_by_launch": {"achieved": round(lbl_achieved, 2), "frac": round(lbl_achieved / PEAK_F32_MFMA, 4),
                                     "kernel_ms_per_step": round(ms.value, 3), "avg_launch_us": round(1e3 * ms.value / max(1, nl.value), 2),
                                     "note": "profiling forces the serial schedule here: a launch's duration is its own (the rocprofv3 kernel trace "
                                             "of `bench.py --overlap 0 --serial-schedule` measures the same thing)"},
                "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": round(alg_bytes),
                "traffic_over_algorithmic": round(traffic / alg_bytes, 3) if traffic else None,
                "launches_per_step": int(nl.value), "kernel_ms_per_step": round(ms.value, 3),
                "algorithmic_gflop_per_step": round(fl.value / 1e9, 2),
                "avg_launch_us": round(1e3 * ms.value / max(1, nl.value), 2), "source_hash": src}

    mp = world * B * S * S * args.steps / 1e6
    value = mp / elapsed
    line = {
        "metric": "megapixels/s encode+decode (256x256 batches)", "value": round(value, 3), "unit": "MP/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"Config 2: batch {B} x {S}x{S} random crops per GPU, quality {q}, mask_pol {MASK_POL}, "
                               "synthetic seeded weights (canonical ChannelProgresssiveWACNN)",
                   "images_per_gpu": B, "height": S, "width": S, "quality": q, "sharding": f"images x {world} ranks",
                   "step_schedule": ("progressivecodec_amd.CodecPipeline.code(): every step = compress() + decompress() of the batch, all inside the timed "
                                     f"region; {args.pairs} encoder / decoder pair(s) of the library's objects, each with its own streams and host "
                                     "threads: a pair decodes step i beside the encode of its next step, and the pairs take the steps in turn")
                   if args.overlap else "every step = compress() then decompress(), strictly one after the other, on one model object",
                   "api": f"CodecPipeline(n_pairs={args.pairs}).code" if args.overlap else "ChannelProgresssiveWACNN.compress / .decompress",
                   "encoder_decoder_pairs": args.pairs if args.overlap else 0,
                   "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "default (4)")},
        "sequential_value": round(seq_value, 3) if seq_value else (None if args.overlap else round(value, 3)),
        "sequential_note": "the same steps strictly one after the other on one codec object (--overlap 0 schedule): what a drop-in "
                           "compress_with_ac caller sees; `value` is the overlapped schedule named in config.step_schedule",
        "rank_ms_per_step": {"min": round(rank_ms_min, 3), "max": round(rank_ms_max, 3)},
        "serial_step_ms": round(1e3 * (tc - ta), 2),
        "enc_ms": round(1e3 * (tb - ta), 2), "dec_ms": round(1e3 * (tc - tb), 2),
        "bpp": round(bpp, 4), "psnr_db": round(psnr, 4), "coded_bytes_job": total_bytes,
        "path_frac_of_f32_mfma_peak": round(value * 1e6 * MFLOP_PER_PX_Q * 1e6 / world / (PEAK_F32_MFMA * 1e12), 4),
        "roofline": roofline,
    }
    if gather:
        line["bitstream_gather"] = gather
    # what ONE rank holds (x N on a node): codec objects (each a 608 MB weights copy + workspaces + ~10 streams), HBM in use on its device,
    # the hardware queues it asked for, its host entropy-coding pool
    free_b, total_b = torch.cuda.mem_get_info(dev)
    used_gib = (total_b - free_b) / 2.0 ** 30
    if world > 1:
        t = torch.tensor([used_gib], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        used_gib = float(t.item())
    nt_, first_, allowed_ = C.c_int(), C.c_int(), C.c_int()
    lib().pc_host_pool_plan(C.byref(nt_), C.byref(first_), C.byref(allowed_))
    line["per_rank_resources"] = {"codec_objects": 2 * args.pairs if pipe is not None else 1, "weights_bytes_per_object": int(sum(v.numel() * 4 for k, v in sd.items()
                                                                                                          if hasattr(v, "numel") and v.dtype == torch.float32)),
                                  "hbm_in_use_gib_max_over_ranks": round(used_gib, 2), "hbm_total_gib": round(total_b / 2.0 ** 30, 1),
                                  "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "default (4)"),
                                  "host_pool_threads": nt_.value, "host_pool_first_cpu": first_.value, "cpus_allowed": allowed_.value}
    # mask / entropy-prep stage: rocprofv3 kernel-trace durations of tools/stage_bench.py on a Config-4 slice (HBM-bound kernels)
    sj, swhy = newest_profile("r*_stage_kernels_rocprof.json", src)
    line["mask_entropy_stage"] = ({"source": f"profiles/{swhy}", **{k: v for k, v in sj.items() if k not in ("source_hash",)}} if sj is not None
                                  else {"source": None, "why": swhy})

    if rank == 0 and not args.lean:
        try:
            line["rans"] = rans_leg(net, x, q, log)
        except Exception as e:                                     # a measurement aid must not take the headline down
            line["rans"] = {"error": repr(e)}
    oj, owhy = newest_profile("r*_overlap_schedule_mfma.json", src)
    line["roofline"]["overlapped_schedule"] = ({"source": f"profiles/{owhy}", **{k: v for k, v in oj.items() if k != "source_hash"}} if oj is not None
                                               else {"source": None, "why": owhy})
    if rank == 0:                                                  # rank 0's batch is the golden's batch (seed 1)
        line["reference_parity"] = reference_parity(net, x, q)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:      # rank 0 at N = 1 only (bench contract)
        line["cpu_baseline"] = cpu_baseline_leg(net, sd, x, q, args, log)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def rans_leg(net, x, q, log):
    """"rANS: report Msym/s per stream and streams in flight".  Host entropy coder on the REAL symbol planes of
    this batch (read back from the codec's taps): serial per-stream rates, the pool's rate on one slice step, the host ms the codec itself
    spent coding in one compress / decompress call and how much of it nothing hides -- for Config 2, and for one Config-5 frame
    (3840x2160, one level), where a stream is 1.04 M symbols and the per-slice decode sits on the serial chain."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    from progressivecodec_amd import entropy
    from progressivecodec_amd.harness import compute_padding
    L = lib_()
    nt, first, allowed = C.c_int(), C.c_int(), C.c_int()
    L.pc_host_pool_plan(C.byref(nt), C.byref(first), C.byref(allowed))

    def one(xb, label):
        B, H, W = xb.shape[0], xb.shape[2], xb.shape[3]
        per = 32 * (H // 16) * (W // 16)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        o = net.compress(xb, q, MASK_POL)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        st = (C.c_double * 6)()
        L.pc_codec_host_stats(net._h, st, 6)
        enc = list(st)
        sym = net.read_tap("sym", np.int32)[: 20 * B * per].reshape(20 * B, per)
        idx = net.read_tap("idx", np.int32)[: 20 * B * per].reshape(20 * B, per)
        t2 = time.perf_counter()
        net.decompress(o["strings"], o["shape"], q, MASK_POL)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        L.pc_codec_host_stats(net._h, st, 6)
        dec = list(st)
        rates = entropy.measure_rans_rates(sym, idx, net._gc, nt.value, streams_per_call=B, sample=48 if per <= 65536 else 6, reps=2)
        nbytes = sum(len(s_) for sl in o["strings"][0] for s_ in sl)
        log(f"rans leg {label}: enc {1e3 * (t1 - t0):.1f} ms (host coding {enc[1]:.2f}, exposed {enc[2]:.2f}), dec {1e3 * (t3 - t2):.1f} ms (host decode {dec[3]:.2f})")
        return {"workload": label, "streams_per_slice_step": B, "symbols_per_stream": per, "slice_steps_per_call": 20,
                "symbols_per_call": int(enc[4]), "y_bytes": nbytes,
                "compress_ms": round(1e3 * (t1 - t0), 2), "host_encode_ms_in_compress": round(enc[1], 3), "host_encode_ms_exposed": round(enc[2], 3),
                "decompress_ms": round(1e3 * (t3 - t2), 2), "host_decode_ms_in_decompress_summed_over_chains": round(dec[3], 3),
                "host_decode_note": "each slice's decode sits between two GPU steps of its chain (base and enhancement chains run one slice apart "
                                    "on two host threads, so each hides behind the other's kernels; in the overlapped bench also behind the encoder object)",
                **rates}

    out = {"pool": {"threads": nt.value, "first_cpu": first.value, "cpus_allowed": allowed.value, "decode_streams_per_thread": 2},
           "config2": one(x, f"Config 2: {x.shape[0]} x {x.shape[2]}x{x.shape[3]}, q={q}")}
    try:
        g = torch.Generator().manual_seed(5)
        lo = torch.rand(1, 3, 270, 480, generator=g)
        f = (F.interpolate(lo, size=(2160, 3840), mode="bilinear", align_corners=False) + 0.03 * torch.randn(1, 3, 2160, 3840, generator=g)).clamp(0, 1)
        pad, _ = compute_padding(2160, 3840)
        xf = F.pad(f, pad).to(x.device)
        one(xf, "warm-up")
        out["config5_frame"] = one(xf, f"Config 5: one 3840x2160 frame (padded to 3840x2176), one level q={q}")
        del xf
        torch.cuda.empty_cache()
    except Exception as e:
        out["config5_frame"] = {"error": repr(e)}
    return out


def lib_():
    from progressivecodec_amd._lib import lib
    return lib()


def cpu_baseline_leg(net, sd, x, q, args, log):
    """CPU baseline: the oracle port (same ATen CPU op sequence as the reference; bit-identical strings with it on one machine --
    tests/test_oracle_vs_golden.py), on a bounded sample of the same workload, plus what separates its output from the GPU's:
    both encode the same images, so every difference is float rounding (oneDNN's summation order vs the contract's fmaf chain)
    flipping a round() or a compare.  Reported: bpp / PSNR of both, the images with any flipped symbol, the slice where each first
    diverges, the symbol / index mismatch rates overall and among the root slices."""
    import numpy as np
    import torch
    from oracle.codec_ref import RefCodec
    log("cpu_baseline leg (oracle port, ATen CPU ops)")
    cores = host_cores()
    torch.set_num_threads(cores)
    orc = RefCodec(sd, "torch")
    orc.update()
    S = x.shape[-1]
    n_img = max(1, min(args.cpu_images, x.shape[0]))
    xg = x[:n_img].contiguous()
    xc = xg.cpu()
    reps = 2
    taps = {}
    t0 = time.perf_counter()
    with torch.no_grad():
        for r in range(reps):
            o = orc.compress(xc, q, taps=taps if r == 0 else None)
            d = orc.decompress(o["strings"], o["shape"], q)
    dt = time.perf_counter() - t0
    g = net.compress(xg, q, MASK_POL)
    HW = (S // 16) ** 2
    gsym = net.read_tap("sym", np.int32)[: 20 * n_img * 32 * HW].reshape(20, n_img, 32 * HW)
    gidx = net.read_tap("idx", np.int32)[: 20 * n_img * 32 * HW].reshape(20, n_img, 32 * HW)
    gd = net.decompress(g["strings"], g["shape"], q, MASK_POL)["x_hat"].cpu()
    csym = np.stack([taps[("b%d" % i) if i < 10 else ("e%d" % (i - 10))]["sym"].numpy().reshape(n_img, -1) for i in range(20)])
    cidx = np.stack([taps[("b%d" % i) if i < 10 else ("e%d" % (i - 10))]["idx"].numpy().reshape(n_img, -1) for i in range(20)])
    same = np.array([[a[b] == c[b] for b in range(n_img)] for a, c in zip(g["strings"][0], o["strings"][0])])    # [20][n_img]
    first = [int(np.argmin(same[:, b])) if not same[:, b].all() else None for b in range(n_img)]
    hist = {}
    for f in first:
        if f is not None:
            hist[str(f)] = hist.get(str(f), 0) + 1
    root_sym = sum(int((gsym[f, b] != csym[f, b]).sum()) for b, f in enumerate(first) if f is not None)
    root_idx = sum(int((gidx[f, b] != cidx[f, b]).sum()) for b, f in enumerate(first) if f is not None)
    nb = lambda st: sum(len(s) for sl in st[0] for s in sl) + sum(len(s) for s in st[1])
    psnr = lambda a, b: -10.0 * torch.log10(torch.mean((a - b) ** 2)).item()
    flip_free = [b for b, f in enumerate(first) if f is None]
    d_ff = max((abs(psnr(xc[b], d["x_hat"][b]) - psnr(xc[b], gd[b])) for b in flip_free), default=None)
    port = {"value": round(reps * n_img * S * S / 1e6 / dt, 4), "unit": "MP/s", "cores": cores, "kind": "port",
            "sample": f"{n_img} of the {x.shape[0]} images of rank 0 ({S}x{S}, q={q}) as one batch, encode+decode x{reps}, "
                      f"torch {torch.__version__} CPU ops + C rANS, {dt:.1f} s",
            "bpp": round(8.0 * nb(o["strings"]) / (n_img * S * S), 6), "psnr_db": round(psnr(xc, d["x_hat"]), 6),
            "gpu_bpp_same_images": round(8.0 * nb(g["strings"]) / (n_img * S * S), 6), "gpu_psnr_db_same_images": round(psnr(xc, gd), 6),
            "z_strings_identical_to_gpu": f"{sum(a == b for a, b in zip(g['strings'][1], o['strings'][1]))}/{n_img}",
            "y_strings_identical_to_gpu": f"{int(same.sum())}/{same.size}",
            "images_with_any_flip": f"{n_img - len(flip_free)}/{n_img}",
            "first_diverging_slice_histogram": hist,
            "symbol_mismatch_rate": float((gsym != csym).mean()), "index_mismatch_rate": float((gidx != cidx).mean()),
            "root_flips": {"symbols": root_sym, "indexes": root_idx,
                           "note": "differing elements inside each flipped image's FIRST diverging slice (the float-rounding flips themselves; "
                                   "later slices differ because their context differs)"},
            "max_abs_psnr_diff_db_flip_free_images": d_ff, "north_star_tolerance_db": 1e-4}
    # the port is the TIMED baseline; its symbol-level differences above are against this box's oneDNN.  The parity figure of the
    # headline configuration is the one against the reference itself:
    # is the reference's CPU bitstream machine-dependent at all?  The port equals the reference string for string in the build container
    # (tests/test_oracle_vs_golden.py); here it runs on THIS box's cores and oneDNN build, against the build container's committed output
    gp = os.path.join(ROOT, "tests", "golden", "config2.json")
    if os.path.exists(gp):
        from progressivecodec_amd.harness import compare_with_golden_strings
        gj = json.load(open(gp))
        r0 = gj["runs"][0]
        same_batch = (n_img, S, q) == (gj["B"], gj["H"], gj["quality"]) and torch.equal(xc, torch.rand(n_img, 3, S, S, generator=torch.Generator().manual_seed(gj["seed"])))
        if same_batch:
            cg = compare_with_golden_strings(o["strings"], r0["y_sha"], r0["z_sha"])
            port["port_strings_identical_to_reference_golden"] = {
                "z": f"{cg['z_strings_identical']}/{n_img}", "y": f"{cg['y_strings_identical']}/{cg['y_strings']}",
                "images_identical": f"{len(cg['flip_free_images'])}/{n_img}", "first_diverging_slice_histogram": cg["first_diverging_slice_histogram"],
                "meaning": "the oracle's ATen port (string-for-string equal to the reference in the build container) run on this box's "
                           f"{cores} cores against the reference's committed build-container output: anything short of all-identical means the "
                           "reference's CPU bitstream depends on the machine (oneDNN kernel selection / core count), not only on the code"}
        else:
            port["port_strings_identical_to_reference_golden"] = None
    port["vs_port_note"] = ("the *_to_gpu / flip / mismatch fields above compare the GPU with the CPU port run on THIS box (its oneDNN build, "
                            f"{cores} threads); the line's `reference_parity` compares the GPU with the committed output of the reference itself")
    return port


def root_flips_vs_reference(gsym, gidx, first):
    """Elements of the GPU's symbol / index planes that differ from the REFERENCE's inside each image's first diverging slice
    (tests/golden/config2_roots.npz, made by tests/golden/make_golden_config2_roots.py from the imported reference): up to that slice the
    two coders saw identical context, so these are the float-rounding flips themselves.  gsym / gidx: [20][B][8192]; first[b]: the GPU's
    first diverging slice against the reference's string hashes (None: flip-free, -1: hyper-latent)."""
    import numpy as np
    rp = os.path.join(ROOT, "tests", "golden", "config2_roots.npz")
    if not os.path.exists(rp):
        return {"source": None, "why": "tests/golden/config2_roots.npz missing"}
    R = np.load(rp)
    ns = ni = n_el = 0
    per_image, moved = {}, []
    for n, (b, s) in enumerate(zip(R["image"].tolist(), R["slice"].tolist())):
        if s < 0:
            continue
        if first[b] != s:                     # the fixture was made with the contract oracle, which the GPU equals bit for bit: never expected
            moved.append(b)
            continue
        ds, di = int((gsym[s, b] != R["sym"][n]).sum()), int((gidx[s, b] != R["idx"][n]).sum())
        ns += ds; ni += di; n_el += gsym.shape[2]
        per_image[str(b)] = [s, ds, di]
    return {"source": "tests/golden/config2_roots.npz (the reference's symbols / indexes of each image's first diverging slice)",
            "symbols": ns, "indexes": ni, "elements_in_root_slices": n_el, "images": len(per_image),
            "expected_by_fixture": {"symbols": int(R["contract_sym_flips"][R["slice"] >= 0].sum()), "indexes": int(R["contract_idx_flips"][R["slice"] >= 0].sum())},
            "per_image_slice_symbols_indexes": per_image, "images_whose_first_slice_moved": moved,
            "rate_per_coded_symbol": (ns + ni) / float(20 * gsym.shape[1] * gsym.shape[2])}


def reference_parity(net, x, q):
    """The GPU's strings and reconstruction of this batch against the REAL reference's (tests/golden/config2.json: made in the build
    container by tests/golden/make_golden_config2.py, which imports /root/reference; 8 threads): the parity figure of the headline
    configuration is pinned to the reference itself, not to the CPU port run on this box (VERDICT r02 "What's weak" 1)."""
    import math
    import numpy as np
    from progressivecodec_amd.harness import compare_with_golden_strings
    gp = os.path.join(ROOT, "tests", "golden", "config2.json")
    if not os.path.exists(gp):
        return {"source": None, "why": "tests/golden/config2.json missing"}
    gj = json.load(open(gp))
    r = gj["runs"][0]
    xc = x.cpu()
    B, S = xc.shape[0], xc.shape[-1]
    if (B, S, q) != (gj["B"], gj["H"], gj["quality"]):
        return {"source": None, "why": f"golden is for {gj['B']} x {gj['H']}^2 at q={gj['quality']}, this run is {B} x {S}^2 at q={q}"}
    g = net.compress(x, q, MASK_POL)                                     # a fresh call: the taps below are this call's planes
    per = 32 * (S // 16) ** 2
    gsym = net.read_tap("sym", np.int32)[: 20 * B * per].reshape(20, B, per)
    gidx = net.read_tap("idx", np.int32)[: 20 * B * per].reshape(20, B, per)
    gd = net.decompress(g["strings"], g["shape"], q, MASK_POL)["x_hat"].cpu()
    import torch
    x_ref = torch.rand(B, 3, S, S, generator=torch.Generator().manual_seed(gj["seed"]))
    if not torch.equal(x_ref, xc):
        return {"source": None, "why": "this rank's batch is not the golden's batch"}
    c = compare_with_golden_strings(g["strings"], r["y_sha"], r["z_sha"])
    ps = lambda a, b: -10.0 * math.log10(torch.mean((a - b) ** 2).item())
    d = [abs(ps(xc[b], gd[b].clamp(0, 1)) - r["psnr_per_image"][b]) for b in range(B)]
    ff = c["flip_free_images"]
    flipped = [b for b in range(B) if b not in ff]
    edges = [1e-4, 3e-4, 1e-3, 3e-3]
    hist = {("<=%g" % e): 0 for e in edges}
    hist[">3e-3"] = 0
    for b in flipped:
        hist[next(("<=%g" % e for e in edges if d[b] <= e), ">3e-3")] += 1
    nb = sum(len(s) for sl in g["strings"][0] for s in sl) + sum(len(s) for s in g["strings"][1])
    other = [{"threads": o["threads"], "y_strings_identical_to_8_threads": sum(a == b_ for sa, sb in zip(r["y_sha"], o["y_sha"]) for a, b_ in zip(sa, sb)),
              "z_strings_identical_to_8_threads": sum(a == b_ for a, b_ in zip(r["z_sha"], o["z_sha"]))} for o in gj["runs"][1:]]
    return {"source": "tests/golden/config2.json (the reference itself, imported in the build container, torch %s CPU, %d threads)" % (r["torch"], r["threads"]),
            "reference_bpp": round(r["bpp"], 6), "reference_psnr_db": round(r["psnr"], 6),
            "gpu_bpp": round(8.0 * nb / (B * S * S), 6), "gpu_psnr_db": round(ps(xc, gd.clamp(0, 1)), 6),
            "z_strings_identical": f"{c['z_strings_identical']}/{B}", "y_strings_identical": f"{c['y_strings_identical']}/{c['y_strings']}",
            "flip_free_images": f"{len(ff)}/{B}", "first_diverging_slice_histogram": c["first_diverging_slice_histogram"],
            "max_abs_psnr_diff_db_flip_free_images": max((d[b] for b in ff), default=None),
            "flip_free_images_within_1e-4_db": f"{sum(d[b] <= 1e-4 for b in ff)}/{len(ff)}",
            "max_abs_psnr_diff_db_flipped_images": max((d[b] for b in range(B) if b not in ff), default=None),
            "images_within_1e-4_db": f"{sum(v <= 1e-4 for v in d)}/{B}",
            "abs_psnr_diff_db_histogram_flipped_images": hist,
            "root_flips": root_flips_vs_reference(gsym, gidx, c["first_diverging_slice"]),
            "reference_at_other_thread_counts_same_machine": other,
            "north_star_tolerance_db": 1e-4}


if __name__ == "__main__":
    main()
