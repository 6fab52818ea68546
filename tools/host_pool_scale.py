#!/usr/bin/env python3
"""Host-only stress of the multi-GPU layout (VERDICT r02 "Next round" 8; no GPU, no 8-GPU node needed): N processes -- one per would-be
rank, each with the pinned host entropy-coding pool `pc_host_pool_plan` gives that rank (LOCAL_RANK / LOCAL_WORLD_SIZE) -- code
Config-2-sized symbol planes concurrently, slice step by slice step, as the codec does (32 streams x 8192 symbols per step: encode,
then decode with the byte-index fast path).  Reports per-rank Msym/s against what one GPU needs from its pool: at V MP/s per GPU a rank
codes and decodes V * 1e6 * 2.5 symbols per second each way (640 latent channels per 256 input pixels = 2.5 symbols per pixel).
usage: python tools/host_pool_scale.py [--ranks 1 2 4 8] [--seconds 3] [--need-mp-s 47]        prints one JSON line per rank count"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def planes(seed, n_streams=32, n=8192):
    """Symbol / index planes with the statistics of the synthetic-weight codec at Config 2 (indexes spread over 0..27, ~3.5 bpp)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    idx = np.clip(rng.normal(14, 6, size=(n_streams, n)), 0, 27).astype(np.int32)   # ~1.4 coded bits per symbol = 3.56 bpp / 2.5 symbols per pixel
    from tests.util import tables_npz
    t = tables_npz()
    scale = np.exp(np.linspace(np.log(0.11), np.log(256), 64))[idx]
    sym = np.round(rng.standard_normal((n_streams, n)) * scale).astype(np.int32)
    return sym, idx


def worker(rank, world, seconds, start_evt, q):
    os.environ["LOCAL_RANK"], os.environ["LOCAL_WORLD_SIZE"] = str(rank), str(world)
    import ctypes as C
    import numpy as np
    from progressivecodec_amd import entropy
    from progressivecodec_amd._lib import check, lib
    from tests.util import tables_npz
    L = lib()
    nt, first, allowed = C.c_int(), C.c_int(), C.c_int()
    L.pc_host_pool_plan(C.byref(nt), C.byref(first), C.byref(allowed))
    t = tables_npz()
    T = entropy.CdfTables(t["gc_cdf"], t["gc_len"], t["gc_off"])
    sym, idx = planes(100 + rank)
    k, n = sym.shape
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    tab = (P(T.cdf), T.cdf.shape[0], T.cdf.shape[1], P(T.length), P(T.offset))
    cap = L.pc_rans_bound(n)
    ob = np.empty((k, cap), np.uint8)
    lens = (C.c_size_t * k)()
    ptrs = (C.c_void_p * k)(*[ob[i].ctypes.data for i in range(k)])
    i8 = np.ascontiguousarray(idx.astype(np.uint8))
    out = np.empty((k, n), np.int32)
    check(L.pc_rans_encode_batch(P(sym), P(idx), k, n, *tab, P(ob), cap, lens, nt.value))      # warm-up, and the streams the decoder reads
    start_evt.wait()
    te = td = 0.0
    steps = 0
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        t0 = time.perf_counter()
        check(L.pc_rans_encode_batch(P(sym), P(idx), k, n, *tab, P(ob), cap, lens, nt.value))
        t1 = time.perf_counter()
        check(L.pc_rans_decode_batch_u8(ptrs, lens, k, P(i8), n, *tab, P(out), nt.value))
        t2 = time.perf_counter()
        te += t1 - t0
        td += t2 - t1
        steps += 1
    assert np.array_equal(out, sym)
    q.put({"rank": rank, "pool_threads": nt.value, "first_cpu": first.value, "cpus_allowed": allowed.value, "slice_steps": steps,
           "encode_msym_s": k * n * steps / te / 1e6, "decode_msym_s": k * n * steps / td / 1e6,
           "coded_bits_per_symbol": 8.0 * sum(lens) / (k * n)})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--need-mp-s", type=float, default=47.0, help="per-GPU MP/s the pool must keep up with (encode AND decode)")
    a = ap.parse_args()
    need = a.need_mp_s * 2.5
    ctx = mp.get_context("spawn")
    for world in a.ranks:
        q, evt = ctx.Queue(), ctx.Event()
        ps = [ctx.Process(target=worker, args=(r, world, a.seconds, evt, q)) for r in range(world)]
        for p in ps:
            p.start()
        time.sleep(2.0 + 0.5 * world)                      # every rank has built its pool and warmed up
        evt.set()
        res = sorted((q.get(timeout=120 + 10 * a.seconds) for _ in ps), key=lambda r: r["rank"])
        for p in ps:
            p.join()
        enc = [r["encode_msym_s"] for r in res]
        dec = [r["decode_msym_s"] for r in res]
        print(json.dumps({"ranks_on_this_host": world, "host_cpus": len(os.sched_getaffinity(0)), "pool_threads_per_rank": res[0]["pool_threads"],
                          "first_cpu_per_rank": [r["first_cpu"] for r in res],
                          "encode_msym_s_per_rank": {"min": round(min(enc), 1), "max": round(max(enc), 1)},
                          "decode_msym_s_per_rank": {"min": round(min(dec), 1), "max": round(max(dec), 1)},
                          "needed_msym_s_per_rank_each_way": round(need, 1), "need_basis": f"{a.need_mp_s} MP/s per GPU x 2.5 symbols per pixel",
                          "headroom_encode": round(min(enc) / need, 2), "headroom_decode": round(min(dec) / need, 2),
                          "coded_bits_per_symbol": round(res[0]["coded_bits_per_symbol"], 3),
                          "workload": "per rank: 32 streams x 8192 symbols per slice step (Config 2), encode then decode (byte-index fast path), back to back for "
                                      f"{a.seconds} s, all ranks at once"}), flush=True)


if __name__ == "__main__":
    main()
