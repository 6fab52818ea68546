#!/usr/bin/env python3
"""Malformed-stream fuzz of the host entropy coder under AddressSanitizer + UBSan (SURVEY.md section 5 "sanitizers").

Runs against progressivecodec_amd/libpc_host_asan.so (`make -C progressivecodec_amd/csrc host-asan`: csrc/pc_host.cpp alone, g++
-fsanitize=address,undefined -fno-sanitize-recover) -- the code that parses untrusted byte strings, and replaces a reference coder whose
only guards are asserts (rans_interface.cpp:110-111, 170-172: a malformed stream or index is undefined behaviour there).  Every call must
RETURN -- PC_OK or a PC_ERR_* code; a sanitizer report aborts the process.  tests/test_host_fuzz.py starts this file in a child process
with the ASan runtime preloaded.

  LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python tools/host_fuzz.py [--cases 12000] [--seed 1]
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODES = {0, -1, -2, -3, -4, -5}                      # PC_OK, ARG, INDEX, BUFFER, TRUNCATED, CDF: what the host entry points may return


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=os.path.join(ROOT, "progressivecodec_amd", "libpc_host_asan.so"))
    ap.add_argument("--cases", type=int, default=12000)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    L = C.CDLL(a.lib)
    L.pc_rans_bound.restype = C.c_size_t
    L.pc_rans_bound.argtypes = [C.c_size_t]
    rng = np.random.default_rng(a.seed)
    T = np.load(os.path.join(ROOT, "tests", "golden", "tables.npz"))
    cdf, ln, off = (np.ascontiguousarray(T[k], np.int32) for k in ("gc_cdf", "gc_len", "gc_off"))
    NC, ST = cdf.shape
    p = lambda x: x.ctypes.data_as(C.c_void_p) if x is not None else None
    sz = C.c_size_t
    hist = {}

    def note(what, rc):
        assert rc in CODES, (what, rc)
        hist.setdefault(what, {}).setdefault(rc, 0)
        hist[what][rc] += 1
        return rc

    def tab(c=cdf, l=ln, o=off):
        return (p(c), C.c_int(c.shape[0]), C.c_int(c.shape[1]), p(l), p(o))

    def encode(sym, idx, t=None, cap=None):
        n = sym.size
        cap = L.pc_rans_bound(n) if cap is None else cap
        buf = np.zeros(max(cap, 4) + 8, np.uint8)[:max(cap, 0)] if cap else np.zeros(0, np.uint8)
        buf = np.ascontiguousarray(np.zeros(cap, np.uint8))
        k = sz()
        rc = L.pc_rans_encode_with_indexes(p(sym), p(idx), sz(n), *(t or tab()), p(buf), sz(cap), C.byref(k))
        return rc, (buf[: k.value].copy() if rc == 0 else None)

    # ---- known-answer vectors through the sanitizer build
    kc = np.array([[0, 8192, 57344, 61440, 65536]], np.int32)
    kl, ko = np.array([5], np.int32), np.array([-1], np.int32)
    ks = np.array([0, 1, -1, 0, 7, -4], np.int32)
    rc, e = encode(ks, np.zeros(6, np.int32), tab(kc, kl, ko))
    assert rc == 0 and e.tobytes().hex() == "a141ad217f1cc771", (rc, e.tobytes().hex() if e is not None else None)
    out = np.empty(6, np.int32)
    assert L.pc_rans_decode_with_indexes(p(e), sz(e.size), p(np.zeros(6, np.int32)), sz(6), *tab(kc, kl, ko), p(out)) == 0 and np.array_equal(out, ks)
    rc, e = encode(np.zeros(8192, np.int32), np.zeros(8192, np.int32), tab(kc, kl, ko))
    assert rc == 0 and e.size == 432 and e.tobytes().hex().startswith("888fafdd7e0000008636922bc39d987c")
    q = np.empty(5, np.uint32)
    assert L.pc_pmf_to_quantized_cdf(p(np.array([0.1, 0.7, 0.15, 0.05], np.float32)), 4, 16, p(q)) == 0 and q.tolist() == [0, 6554, 52429, 62259, 65536]
    assert L.pc_pmf_to_quantized_cdf(p(np.array([1e-9, 0.999, 1e-9, 1e-3], np.float32)), 4, 16, p(q)) == 0 and q.tolist() == [0, 1, 65471, 65472, 65536]

    def sample(n, wild=False):
        idx = rng.integers(0, 28 if not wild else NC, n).astype(np.int32)
        spread = np.where(idx < 8, 1, np.where(idx < 20, 3, 30))
        sym = np.rint(rng.standard_normal(n) * spread).astype(np.int32)
        if wild:                                               # a few symbols far outside the tables: the bypass path, multi-nibble counts
            k = rng.integers(0, n, max(1, n // 50))
            sym[k] = rng.integers(-2 ** 20, 2 ** 20, k.size).astype(np.int32)
        return sym, idx

    # ---- round trips through every decoder form
    for it in range(200):
        n = int(rng.integers(1, 700))
        sym, idx = sample(n, wild=it % 3 == 0)
        rc, e = encode(sym, idx)
        assert rc == 0, rc
        out = np.empty(n, np.int32)
        assert L.pc_rans_decode_with_indexes(p(e), sz(e.size), p(idx), sz(n), *tab(), p(out)) == 0 and np.array_equal(out, sym)
        st = (C.c_uint64 * 2)(0, 0)                            # set_stream / decode_stream in pieces
        cut = int(rng.integers(0, n + 1))
        o2 = np.empty(n, np.int32)
        assert L.pc_rans_decode_stream(p(e), sz(e.size), st, p(idx), sz(cut), *tab(), p(o2)) == 0
        assert L.pc_rans_decode_stream(p(e), sz(e.size), st, p(idx[cut:]) if cut < n else p(idx), sz(n - cut), *tab(), p(o2[cut:]) if cut < n else p(o2)) == 0
        assert np.array_equal(o2, sym)
        i8 = np.ascontiguousarray(idx.astype(np.uint8))
        ptrs, lens = (C.c_void_p * 1)(e.ctypes.data), (sz * 1)(e.size)
        o3 = np.empty(n, np.int32)
        assert L.pc_rans_decode_batch_u8(ptrs, lens, sz(1), p(i8), sz(n), *tab(), p(o3), 1) == 0 and np.array_equal(o3, sym)
    # batches over the thread pool, odd stream counts (a lone last stream is paired with itself)
    for ns in (1, 2, 3, 7):
        n = 257
        S = [sample(n, wild=True) for _ in range(ns)]
        sym = np.ascontiguousarray(np.stack([s for s, _ in S]))
        idx = np.ascontiguousarray(np.stack([i for _, i in S]))
        cap = L.pc_rans_bound(n)
        ob = np.zeros((ns, cap), np.uint8)
        lens = (sz * ns)()
        assert L.pc_rans_encode_batch(p(sym), p(idx), sz(ns), sz(n), *tab(), p(ob), sz(cap), lens, 0) == 0
        ptrs = (C.c_void_p * ns)(*[ob[i].ctypes.data for i in range(ns)])
        o = np.empty((ns, n), np.int32)
        assert L.pc_rans_decode_batch(ptrs, lens, sz(ns), p(idx), sz(n), *tab(), p(o), 0) == 0 and np.array_equal(o, sym)
        o[:] = 0
        assert L.pc_rans_decode_batch_u8(ptrs, lens, sz(ns), p(np.ascontiguousarray(idx.astype(np.uint8))), sz(n), *tab(), p(o), 0) == 0 and np.array_equal(o, sym)

    # ---- mutated / truncated / index-corrupted streams: every call returns a code
    n_bad = 0
    for it in range(a.cases):
        n = int(rng.integers(1, 400))
        sym, idx = sample(n, wild=it % 4 == 0)
        rc, e = encode(sym, idx)
        assert rc == 0
        e = e.copy()
        kind = it % 8
        didx = idx.copy()
        if kind == 0:                                          # bit flips
            for _ in range(int(rng.integers(1, 6))):
                e[rng.integers(0, e.size)] ^= np.uint8(1 << rng.integers(0, 8))
        elif kind == 1:                                        # truncation (incl. < 8 bytes, and lengths that are not multiples of 4)
            e = e[: int(rng.integers(0, e.size))].copy()
        elif kind == 2:                                        # random bytes of random length
            e = rng.integers(0, 256, int(rng.integers(0, 64)), dtype=np.uint8)
        elif kind == 3:                                        # indexes out of range / negative
            k = rng.integers(0, n, max(1, n // 10))
            didx[k] = rng.choice(np.array([-1, NC, NC + 7, 2 ** 31 - 1, -2 ** 31, 255, 256], np.int64), k.size).astype(np.int32)
        elif kind == 4:                                        # other valid indexes than the encoder used: desynchronised decode
            didx = rng.integers(0, NC, n).astype(np.int32)
        elif kind == 5:                                        # all-ones / all-zero words: the longest bypass runs
            e[8:] = np.uint8(0xFF if it % 16 < 8 else 0)
        elif kind == 6:                                        # the state words themselves
            e[: 8] = rng.integers(0, 256, 8, dtype=np.uint8)
        else:                                                  # more symbols asked for than the stream holds
            pass
        nn = n if kind != 7 else n + int(rng.integers(1, 3000))
        if kind == 7:
            didx = np.resize(didx, nn)
        e = np.ascontiguousarray(e) if e.size else np.zeros(1, np.uint8)[:0]
        eb = np.ascontiguousarray(np.concatenate([e, np.zeros(0, np.uint8)]))
        elen = eb.size
        if elen == 0:
            eb = np.zeros(4, np.uint8)
        out = np.empty(nn, np.int32)
        r1 = note("decode_with_indexes", L.pc_rans_decode_with_indexes(p(eb), sz(elen), p(didx), sz(nn), *tab(), p(out)))
        st = (C.c_uint64 * 2)(0, 0)
        note("decode_stream", L.pc_rans_decode_stream(p(eb), sz(elen), st, p(didx), sz(nn // 2), *tab(), p(out)))
        st2 = (C.c_uint64 * 2)(int(rng.integers(0, 2 ** 62)), int(rng.integers(0, 2 + 2 * elen)))      # a corrupt caller-held state
        note("decode_stream_bad_state", L.pc_rans_decode_stream(p(eb), sz(elen), st2, p(didx), sz(nn), *tab(), p(out)))
        ptrs, lens = (C.c_void_p * 2)(eb.ctypes.data, eb.ctypes.data), (sz * 2)(elen, elen)
        i8 = np.ascontiguousarray(np.concatenate([didx, didx]).astype(np.uint8))
        o2 = np.empty(2 * nn, np.int32)
        r2 = note("decode_batch_u8", L.pc_rans_decode_batch_u8(ptrs, lens, sz(2), p(i8), sz(nn), *tab(), p(o2), 1))
        note("decode_batch", L.pc_rans_decode_batch(ptrs, lens, sz(2), p(np.ascontiguousarray(np.concatenate([didx, didx]))), sz(nn), *tab(), p(o2), 1))
        n_bad += (r1 != 0) + (r2 != 0)
    # ---- malformed tables, encoder extremes, pmf extremes
    for it in range(600):
        n = 64
        sym, idx = sample(n)
        c2, l2, o2 = cdf.copy(), ln.copy(), off.copy()
        r = int(idx[0])
        k = it % 6
        if k == 0:
            c2[r, : l2[r]] = np.sort(rng.integers(-5, 70000, l2[r]).astype(np.int32))[::-1]          # decreasing
        elif k == 1:
            l2[r] = int(rng.choice([0, 1, -3, ST + 1, 2 ** 30]))
        elif k == 2:
            c2[r, int(rng.integers(0, l2[r]))] = int(rng.choice([-1, 2 ** 31 - 1, -2 ** 31, 70000]))
        elif k == 3:
            c2[r, : l2[r]] = 0
        elif k == 4:
            o2[r] = int(rng.choice([2 ** 31 - 1, -2 ** 31]))
        else:
            sym[:8] = np.array([2 ** 31 - 1, -2 ** 31, 2 ** 31 - 2, -2 ** 31 + 1, 2 ** 30, -2 ** 30, 0, 1], np.int32)
        cap = L.pc_rans_bound(n)
        buf = np.zeros(cap, np.uint8)
        kk = sz()
        rc = note("encode_malformed", L.pc_rans_encode_with_indexes(p(sym), p(idx), sz(n), *tab(c2, l2, o2), p(buf), sz(cap), C.byref(kk)))
        _, good = encode(*sample(n))
        out = np.empty(n, np.int32)
        note("decode_malformed_tables", L.pc_rans_decode_with_indexes(p(good), sz(good.size), p(idx), sz(n), *tab(c2, l2, o2), p(out)))
        ptrs, lens = (C.c_void_p * 1)(good.ctypes.data), (sz * 1)(good.size)
        note("decode_u8_malformed_tables", L.pc_rans_decode_batch_u8(ptrs, lens, sz(1), p(np.ascontiguousarray(idx.astype(np.uint8))), sz(n), *tab(c2, l2, o2), p(out), 1))
        small = int(rng.integers(0, 40))                       # output buffers that are too small
        b2 = np.zeros(max(small, 1), np.uint8)
        note("encode_small_buffer", L.pc_rans_encode_with_indexes(p(sym), p(idx), sz(n), *tab(), p(b2), sz(small & ~3), C.byref(kk)))
    for pm in ([float("nan"), 0.5], [float("inf"), 0.1], [-0.1, 1.1], [0.0, 0.0, 0.0], [1e30, 1.0], [3e9, 1.0], [1e-30] * 7, [1.0], [0.5] * 4000):
        arr = np.array(pm, np.float32)
        q = np.empty(arr.size + 1, np.uint32)
        for prec in (16, 1, 0, 17, 8):
            note("pmf_to_quantized_cdf", L.pc_pmf_to_quantized_cdf(p(arr), C.c_int(arr.size), C.c_int(prec), p(q)))
    print(json.dumps({"lib": os.path.basename(a.lib), "mutated_streams": a.cases, "calls_that_returned_an_error": n_bad,
                      "return_codes": {k: {str(c): v for c, v in sorted(h.items())} for k, h in hist.items()}, "sanitizer_reports": 0}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
