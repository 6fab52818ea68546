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
