#!/usr/bin/env python3
"""Per-stream rate of the host rANS coder on ONE long stream (a 4K frame is one stream of 1 044 480 symbols per
slice step and its decode sits on the serial chain -- DESIGN.md section 6 `rans`).  CPU only: loads a host-only build of
csrc/pc_host.cpp (`make -C progressivecodec_amd/csrc host` -> libpc_host.so; any other build through --lib for an A/B) and codes the
REFERENCE's own symbol / index planes (tests/golden/config2_roots.npz: 21 slices x 8192 symbols, real statistics), tiled to the length of
a Config-5 slice.

  python tools/host_decode_rate.py [--lib path.so] [--n 1044480] [--reps 7]
"""
import argparse
import ctypes as C
import json
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "progressivecodec_amd", "libpc_host.so"))
    ap.add_argument("--n", type=int, default=32 * 136 * 240)
    ap.add_argument("--reps", type=int, default=7)
    a = ap.parse_args()
    L = C.CDLL(a.lib)
    L.pc_rans_bound.restype = C.c_size_t
    L.pc_rans_bound.argtypes = [C.c_size_t]
    T = np.load(os.path.join(ROOT, "tests", "golden", "tables.npz"))
    cdf, ln, off = (np.ascontiguousarray(T[k], np.int32) for k in ("gc_cdf", "gc_len", "gc_off"))
    R = np.load(os.path.join(ROOT, "tests", "golden", "config2_roots.npz"))
    keep = R["slice"] >= 0
    sym = np.ascontiguousarray(np.resize(R["sym"][keep].astype(np.int32).ravel(), a.n))
    idx = np.ascontiguousarray(np.resize(R["idx"][keep].astype(np.int32).ravel(), a.n))
    p = lambda x: x.ctypes.data_as(C.c_void_p)
    cap = L.pc_rans_bound(a.n)
    buf = np.empty(cap, np.uint8)
    n_out = C.c_size_t()
    tab = (p(cdf), C.c_int(cdf.shape[0]), C.c_int(cdf.shape[1]), p(ln), p(off))
    t_enc = 1e9
    for _ in range(a.reps):
        t0 = time.perf_counter()
        rc = L.pc_rans_encode_with_indexes(p(sym), p(idx), C.c_size_t(a.n), *tab, p(buf), C.c_size_t(cap), C.byref(n_out))
        t_enc = min(t_enc, time.perf_counter() - t0)
        assert rc == 0, rc
    enc = np.ascontiguousarray(buf[: n_out.value])
    out = np.empty(a.n, np.int32)
    t_ref = 1e9
    for _ in range(a.reps):
        t0 = time.perf_counter()
        rc = L.pc_rans_decode_with_indexes(p(enc), C.c_size_t(enc.size), p(idx), C.c_size_t(a.n), *tab, p(out))
        t_ref = min(t_ref, time.perf_counter() - t0)
        assert rc == 0, rc
    assert np.array_equal(out, sym)
    i8 = np.ascontiguousarray(idx.astype(np.uint8))
    ptrs = (C.c_void_p * 1)(enc.ctypes.data)
    lens = (C.c_size_t * 1)(enc.size)
    out[:] = 0
    t_fast = 1e9
    for _ in range(a.reps):
        t0 = time.perf_counter()
        rc = L.pc_rans_decode_batch_u8(ptrs, lens, C.c_size_t(1), p(i8), C.c_size_t(a.n), *tab, p(out), C.c_int(1))
        t_fast = min(t_fast, time.perf_counter() - t0)
        assert rc == 0, rc
    assert np.array_equal(out, sym)
    print(json.dumps({"lib": os.path.relpath(a.lib, ROOT) if a.lib.startswith(ROOT) else a.lib, "symbols": a.n, "coded_bits_per_symbol": round(8.0 * enc.size / a.n, 3),
                      "encode_msym_s": round(a.n / t_enc / 1e6, 1), "decode_reference_api_msym_s": round(a.n / t_ref / 1e6, 1),
                      "decode_fast_one_stream_msym_s": round(a.n / t_fast / 1e6, 1), "decode_fast_ms_per_slice_step": round(1e3 * t_fast, 3),
                      "note": "one stream on one host thread; pc_rans_decode_batch_u8's time includes building the start table (64 rows)"}))


if __name__ == "__main__":
    main()
