"""N > 1 path on CPU: world_size-2 gloo processes exercise the image sharding and the final
variable-length bitstream gather (progressivecodec_amd/parallel.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from progressivecodec_amd.parallel import gather_bitstreams, shard_range


def test_shard_range_partitions_everything():
    for n in (0, 1, 7, 32, 256):
        for world in (1, 2, 3, 8):
            r = [shard_range(n, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            sizes = [e - b for b, e in r]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _strings_for(rank, n_local):
    return [[bytes([(7 * rank + 3 * s + b) % 251]) * (4 * (1 + (rank + s + 2 * b) % 5)) for b in range(n_local)] for s in range(3)]


def _worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b, e = shard_range(5, rank, world)                     # ragged: 3 + 2 images
    mine = _strings_for(rank, e - b)
    got = gather_bitstreams(mine)
    want = [sum((_strings_for(r, shard_range(5, r, world)[1] - shard_range(5, r, world)[0])[s] for r in range(world)), [])
            for s in range(3)]
    assert got == want, (rank, got, want)
    dist.destroy_process_group()


def test_gather_bitstreams_gloo_world2():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port), nprocs=2, join=True)


# ------------------------------------------------------------------ bench.py's own multi-rank path
def _bench_worker(rank, world, port, images_per_gpu):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    total, b, e = bench.job_plan(world, rank, images_per_gpu)
    assert total == world * images_per_gpu and e - b == images_per_gpu and b == rank * images_per_gpu
    # the strings a rank's compress() would return: 20 y slices x its images, then its z strings; content identifies (rank, slice, image)
    ys = [[bytes([(rank * 31 + s * 7 + i) % 251]) * (8 + 4 * ((rank + s + i) % 9)) for i in range(images_per_gpu)] for s in range(20)]
    zs = [bytes([200 + rank]) * (8 + 4 * (i % 3)) for i in range(images_per_gpu)]
    nbytes, n_img = bench.final_gather([ys, zs], world, rank, images_per_gpu)
    assert n_img == total
    want = sum(8 + 4 * ((r + s + i) % 9) for r in range(world) for s in range(20) for i in range(images_per_gpu)) + \
        sum(8 + 4 * (i % 3) for r in range(world) for i in range(images_per_gpu))
    assert nbytes == want, (rank, nbytes, want)
    dist.destroy_process_group()


def test_bench_sharding_and_gather_at_world_size_8_gloo():
    """bench.py's job plan (weak scaling, contiguous shards) and its final bitstream gather + checks, 8 ranks on CPU over gloo --
    the code path the 8-GPU run takes after its timed region, with nccl in place of gloo."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_bench_worker, args=(8, port, 3), nprocs=8, join=True)


def test_bench_refuses_more_ranks_than_gpus():
    """`python bench.py --gpus N` on a node with fewer than N GPUs must fail loudly, never report a smaller job as n_gpus = N."""
    import subprocess
    import sys
    n_have = torch.cuda.device_count()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(max(2, n_have + 1)), "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "refusing" in r.stderr and "{" not in r.stdout
    # under a launcher whose world size disagrees with --gpus it also refuses
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=env2, capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "must agree" in r.stderr


def _fake_kfd(tmp_path, simd_counts):
    root = tmp_path / "nodes"
    for i, sc in enumerate(simd_counts):
        d = root / str(i)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count {0 if sc else 64}\nsimd_count {sc}\nmem_banks_count 1\ngfx_target_version {90500 if sc else 0}\n")
    return str(root)


def test_gpu_count_comes_from_the_kfd_topology_not_from_hip(tmp_path, monkeypatch):
    """count_gpus_without_hip: GPU nodes of the KFD sysfs tree (simd_count > 0; CPU nodes have 0), narrowed by the *_VISIBLE_DEVICES
    variables -- and torch.cuda is never consulted (VERDICT r03 item 4 / ADVICE r03: the launcher's parent must stay GPU-free)."""
    from progressivecodec_amd.parallel import count_gpus_without_hip

    def boom(*a, **k):
        raise AssertionError("the launcher's parent touched torch.cuda")
    monkeypatch.setattr(torch.cuda, "device_count", boom)
    monkeypatch.setattr(torch.cuda, "is_available", boom)
    root = _fake_kfd(tmp_path, [0, 0, 1024, 1024, 1024, 1024, 1024, 1024, 1024, 1024])       # two CPU sockets + eight MI355X
    assert count_gpus_without_hip(env={}, sysfs_root=root) == 8
    assert count_gpus_without_hip(env={"HIP_VISIBLE_DEVICES": "0,1,2"}, sysfs_root=root) == 3
    assert count_gpus_without_hip(env={"ROCR_VISIBLE_DEVICES": "0,1,2,3", "HIP_VISIBLE_DEVICES": "0,1"}, sysfs_root=root) == 2
    assert count_gpus_without_hip(env={"CUDA_VISIBLE_DEVICES": ""}, sysfs_root=root) == 0
    assert count_gpus_without_hip(env={"HIP_VISIBLE_DEVICES": "0,1,2,3,4,5,6,7,8,9,10,11"}, sysfs_root=root) == 8
    assert count_gpus_without_hip(env={}, sysfs_root=str(tmp_path / "absent"), probe_in_child=False) == 0


def test_launcher_parent_never_touches_the_gpu_runtime(tmp_path, monkeypatch):
    """bench.launch_ranks: counts devices GPU-free, refuses a job larger than the node, and starts exactly N children with the rank
    environment -- with torch.cuda.* rigged to raise in the parent, and Popen replaced by a recorder (no process is started)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    import progressivecodec_amd.parallel as par

    def boom(*a, **k):
        raise AssertionError("the launcher's parent touched torch.cuda")
    for name in ("device_count", "is_available", "init", "current_device", "set_device"):
        monkeypatch.setattr(torch.cuda, name, boom)
    root = _fake_kfd(tmp_path, [0, 1024, 1024, 1024, 1024])
    real = par.count_gpus_without_hip
    monkeypatch.setattr(par, "count_gpus_without_hip", lambda env=None, **kw: real(env=env, sysfs_root=root, probe_in_child=False))
    for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    started = []

    class FakeProc:
        returncode = 0

        def __init__(self, argv, env=None, stdout=None):
            started.append((argv, env))

        def communicate(self):
            return b'{"fake": 1}\n', None

        def wait(self):
            return 0
    monkeypatch.setattr(bench.subprocess, "Popen", FakeProc)
    assert bench.launch_ranks(8, ["--gpus", "8"]) == 2 and not started          # four GPUs: an 8-rank job is refused, nothing started
    assert bench.launch_ranks(4, ["--gpus", "4", "--steps", "3"]) == 0
    assert len(started) == 4
    for r, (argv, env) in enumerate(started):
        assert argv[-4:] == ["--gpus", "4", "--steps", "3"]
        assert (env["RANK"], env["LOCAL_RANK"], env["WORLD_SIZE"], env["LOCAL_WORLD_SIZE"], env["MASTER_ADDR"]) == (str(r), str(r), "4", "4", "127.0.0.1")
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" or "HSA_ENABLE_IPC_MODE_LEGACY" in os.environ
