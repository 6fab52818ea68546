"""Multi-GPU: images shard across ranks, weights replicate, no collective on the coding path.

Every image is coded independently (convs are per-sample, the mask quantile is per image
-- reference layers/masking.py:215-222 -- and there is one rANS stream per image and slice
-- entropy_models.py:227), so the batch splits contiguously over the ranks of one node (one
process per GPU).  The only exchange is the optional final gather of the variable-length byte
strings (a length table, then the padded bytes): two all-gathers over RCCL/xGMI, after coding.
The reference has no counterpart (it is single-device: SURVEY.md section 2.1).
"""
from typing import List, Sequence, Tuple


def shard_range(n_images: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [begin, end) image range of `rank`; remainders go to the first ranks."""
    if world <= 0 or not (0 <= rank < world) or n_images < 0:
        raise ValueError("invalid shard arguments")
    base, rem = divmod(n_images, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def _visible_list(env):
    """The device subset the HIP runtime will expose, from the environment a child rank inherits: ROCR_VISIBLE_DEVICES filters first
    (at the ROCr level), then HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES index into what is left.  Returns the number of entries of the
    narrowest list, or None when no variable is set.  An empty string hides every device (0)."""
    n = None
    for name in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(name)
        if v is None:
            continue
        k = len([t for t in v.split(",") if t.strip() != ""])
        n = k if n is None else min(n, k)
    return n


def count_gpus_without_hip(env=None, sysfs_root="/sys/class/kfd/kfd/topology/nodes", probe_in_child=True):
    """Number of GPUs a rank started from this process would see -- WITHOUT initialising HIP / HSA here.  A launcher that goes on to
    start its ranks (fork + exec) must not have touched the GPU runtime first: torch.cuda.device_count() can fall through to
    hipGetDeviceCount on ROCm, so the count is taken from the KFD topology in sysfs (a node with simd_count > 0 is a GPU; CPU
    nodes have 0), narrowed by the *_VISIBLE_DEVICES variables.  Without the sysfs tree (no amdgpu driver: the build container) the
    count comes from a short-lived child process that exits before any rank is started; with neither, 0."""
    import glob
    import os
    env = os.environ if env is None else env
    nodes = sorted(glob.glob(os.path.join(sysfs_root, "*", "properties")))
    n = None
    if nodes:
        n = 0
        for f in nodes:
            try:
                props = dict(line.split(None, 1) for line in open(f).read().splitlines() if " " in line)
            except OSError:                          # a node this cgroup may not read is not a device this process can use
                continue
            if int(props.get("simd_count", "0").strip() or 0) > 0:
                n += 1
    elif probe_in_child:
        import subprocess
        import sys
        try:
            r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], env=dict(env), capture_output=True,
                               text=True, timeout=300)
            n = int(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 and r.stdout.strip() else 0
            return n                                 # the child already honoured the *_VISIBLE_DEVICES variables
        except (OSError, ValueError, subprocess.SubprocessError):
            n = 0
    else:
        n = 0
    vis = _visible_list(env)
    return n if vis is None else min(n, vis)


def gather_bitstreams(strings: Sequence[Sequence[bytes]], group=None, device=None) -> List[List[bytes]]:
    """All-gather per-slice lists of per-image byte strings from every rank, in rank order.

    `strings` is [n_slices][n_local_images] (the y_strings of compress(), or [z_strings]).  Returns
    [n_slices][n_total_images] on every rank.  Two collectives: int64 lengths, then uint8 payload padded
    to the largest per-rank total.  Works with the nccl (= RCCL) backend on GPU tensors and with gloo on CPU.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    n_slices = len(strings)
    n_local = len(strings[0]) if n_slices else 0
    counts = torch.tensor([n_slices, n_local], dtype=torch.int64, device=device)
    all_counts = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts, group=group)
    if any(int(c[0]) != n_slices for c in all_counts):
        raise ValueError("ranks disagree on the number of slices")
    max_local = max(int(c[1]) for c in all_counts)
    lens = torch.zeros((n_slices, max_local), dtype=torch.int64, device=device)
    for s in range(n_slices):
        for b in range(n_local):
            lens[s, b] = len(strings[s][b])
    all_lens = [torch.zeros_like(lens) for _ in range(world)]
    dist.all_gather(all_lens, lens, group=group)
    total_max = max(int(l.sum()) for l in all_lens)
    payload = torch.zeros(max(total_max, 1), dtype=torch.uint8, device=device)
    flat = b"".join(strings[s][b] for s in range(n_slices) for b in range(n_local))
    if flat:
        payload[: len(flat)] = torch.frombuffer(bytearray(flat), dtype=torch.uint8).to(device)
    all_payload = [torch.zeros_like(payload) for _ in range(world)]
    dist.all_gather(all_payload, payload, group=group)
    out: List[List[bytes]] = [[] for _ in range(n_slices)]
    for r in range(world):
        buf = all_payload[r].cpu().numpy().tobytes()
        ln = all_lens[r].cpu()
        off = 0
        for s in range(n_slices):
            for b in range(int(all_counts[r][1])):
                n = int(ln[s, b])
                out[s].append(buf[off:off + n])
                off += n
    return out
