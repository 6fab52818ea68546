"""Entropy-model tables (init-time host logic) and the drop-in entropy-coder classes.

* ``gaussian_conditional_tables`` / ``entropy_bottleneck_tables`` mirror
  ``GaussianConditional.update`` (reference src/compress/entropy_models/entropy_models.py:599-624)
  and ``EntropyBottleneck.update`` (:354-393): float PMFs are evaluated with the same ATen
  CPU ops the reference uses and quantised by ``pc_pmf_to_quantized_cdf`` (the native
  replacement of ``compressai._CXX.pmf_to_quantized_cdf``, cpp_exts/ops/ops.cpp:10-67).
* ``RansEncoder`` / ``RansDecoder`` / ``pmf_to_quantized_cdf`` reproduce the Python surface of
  ``compressai.ans`` and ``compressai._CXX`` (cpp_exts/rans/rans_interface.cpp:352-372,
  ops.cpp:69-76) on top of the C ABI, so ``entropy_models.py:13,33-36`` can import them unchanged.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from ._lib import check, lib


@dataclass
class CdfTables:
    cdf: np.ndarray      # [n, stride] int32   (_quantized_cdf)
    length: np.ndarray   # [n] int32           (_cdf_length)
    offset: np.ndarray   # [n] int32           (_offset)

    def __post_init__(self):
        self.cdf = np.ascontiguousarray(self.cdf, np.int32)
        self.length = np.ascontiguousarray(self.length, np.int32).reshape(-1)
        self.offset = np.ascontiguousarray(self.offset, np.int32).reshape(-1)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def pmf_to_quantized_cdf(pmf, precision=16):
    """compressai._CXX.pmf_to_quantized_cdf(pmf: list[float], precision) -> list[int]  (ops.cpp:10)."""
    p = np.ascontiguousarray(pmf, np.float32).reshape(-1)
    out = np.zeros(p.size + 1, np.uint32)
    check(lib().pc_pmf_to_quantized_cdf(_ptr(p), p.size, int(precision), _ptr(out)), "pmf_to_quantized_cdf")
    return out.tolist()


def _tables_from_lists(cdfs, sizes, offsets):
    n = len(cdfs)
    stride = max(len(r) for r in cdfs)
    dense = np.zeros((n, stride), np.int32)
    for i, r in enumerate(cdfs):
        dense[i, : len(r)] = r
    return CdfTables(dense, np.asarray(sizes, np.int32), np.asarray(offsets, np.int32))


class RansEncoder:
    """compressai.ans.RansEncoder (rans_interface.cpp:193-204)."""

    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets) -> bytes:
        t = cdfs if isinstance(cdfs, CdfTables) else _tables_from_lists(cdfs, cdfs_sizes, offsets)
        return rans_encode(symbols, indexes, t)


class BufferedRansEncoder:
    """compressai.ans.BufferedRansEncoder (rans_interface.cpp:99-191): symbols of several encode_with_indexes() calls -- each with
    its own tables -- go into ONE stream, written by flush().  The reference pushes (start, range) records and pops them in reverse
    at flush time; that is, byte for byte, one encode over the concatenation, which is what flush() asks the C ABI for (the calls'
    tables are stacked into one table set and each call's indexes shifted to its rows)."""

    def __init__(self):
        self._sym, self._idx, self._tabs = [], [], []

    def encode_with_indexes(self, symbols, indexes, cdfs, cdfs_sizes, offsets) -> None:
        t = cdfs if isinstance(cdfs, CdfTables) else _tables_from_lists(cdfs, cdfs_sizes, offsets)
        s = np.ascontiguousarray(symbols, np.int32).reshape(-1)
        i = np.ascontiguousarray(indexes, np.int32).reshape(-1)
        if s.size != i.size:
            raise ValueError("`symbols` and `indexes` should have the same size.")
        base = 0
        for k, u in enumerate(self._tabs):
            if u is t:
                break
            base += u.cdf.shape[0]
        else:
            self._tabs.append(t)
        self._sym.append(s)
        self._idx.append(i + base)

    def flush(self) -> bytes:
        if not self._tabs:
            t = CdfTables(np.array([[0, 1 << 16]], np.int32), np.array([2], np.int32), np.array([0], np.int32))
            out = rans_encode(np.zeros(0, np.int32), np.zeros(0, np.int32), t)
        else:
            stride = max(u.cdf.shape[1] for u in self._tabs)
            cdf = np.concatenate([np.pad(u.cdf, ((0, 0), (0, stride - u.cdf.shape[1]))) for u in self._tabs])
            t = CdfTables(cdf, np.concatenate([u.length for u in self._tabs]), np.concatenate([u.offset for u in self._tabs]))
            out = rans_encode(np.concatenate(self._sym), np.concatenate(self._idx), t)
        self._sym, self._idx, self._tabs = [], [], []
        return out


class RansDecoder:
    """compressai.ans.RansDecoder (rans_interface.cpp:206-350): decode_with_indexes, and set_stream / decode_stream for a stream
    that is decoded in several calls (the decoder state lives in this object, as in the reference)."""

    def __init__(self):
        self._stream = None
        self._state = None

    def decode_with_indexes(self, encoded, indexes, cdfs, cdfs_sizes, offsets):
        t = cdfs if isinstance(cdfs, CdfTables) else _tables_from_lists(cdfs, cdfs_sizes, offsets)
        return rans_decode(encoded, indexes, t).tolist()

    def set_stream(self, encoded) -> None:
        self._stream = np.frombuffer(bytes(encoded), np.uint8)
        self._state = np.zeros(2, np.uint64)

    def decode_stream(self, indexes, cdfs, cdfs_sizes, offsets):
        if self._stream is None:
            raise ValueError("set_stream() first")
        t = cdfs if isinstance(cdfs, CdfTables) else _tables_from_lists(cdfs, cdfs_sizes, offsets)
        i = np.ascontiguousarray(indexes, np.int32).reshape(-1)
        out = np.empty(i.size, np.int32)
        check(lib().pc_rans_decode_stream(_ptr(self._stream), self._stream.size, _ptr(self._state), _ptr(i), i.size, _ptr(t.cdf),
                                          t.cdf.shape[0], t.cdf.shape[1], _ptr(t.length), _ptr(t.offset), _ptr(out)), "rans_decode_stream")
        return out.tolist()


def rans_encode(symbols, indexes, t: CdfTables) -> bytes:
    s = np.ascontiguousarray(symbols, np.int32).reshape(-1)
    i = np.ascontiguousarray(indexes, np.int32).reshape(-1)
    if s.size != i.size:
        raise ValueError("`symbols` and `indexes` should have the same size.")
    cap = lib().pc_rans_bound(s.size)
    out = np.empty(cap // 4, np.uint32)
    n = C.c_size_t(0)
    check(lib().pc_rans_encode_with_indexes(_ptr(s), _ptr(i), s.size, _ptr(t.cdf), t.cdf.shape[0], t.cdf.shape[1],
                                            _ptr(t.length), _ptr(t.offset), _ptr(out), cap, C.byref(n)), "rans_encode")
    return out.view(np.uint8)[: n.value].tobytes()


def rans_decode(encoded: bytes, indexes, t: CdfTables) -> np.ndarray:
    i = np.ascontiguousarray(indexes, np.int32).reshape(-1)
    buf = np.frombuffer(encoded, np.uint8)
    out = np.empty(i.size, np.int32)
    check(lib().pc_rans_decode_with_indexes(_ptr(buf), buf.size, _ptr(i), i.size, _ptr(t.cdf), t.cdf.shape[0],
                                            t.cdf.shape[1], _ptr(t.length), _ptr(t.offset), _ptr(out)), "rans_decode")
    return out


def _pmf_to_cdf(pmf, tail_mass, pmf_length, max_length):
    """EntropyModel._pmf_to_cdf (entropy_models.py:172-180)."""
    cdf = np.zeros((len(pmf_length), max_length + 2), np.int32)
    for i in range(len(pmf_length)):
        prob = np.concatenate([pmf[i, : pmf_length[i]], tail_mass[i]])
        c = pmf_to_quantized_cdf(prob, 16)
        cdf[i, : len(c)] = c
    return cdf


def gaussian_conditional_tables(scale_table, tail_mass=1e-9) -> CdfTables:
    """GaussianConditional.update (entropy_models.py:599-624)."""
    import scipy.stats
    import torch
    st = torch.as_tensor(np.asarray(scale_table, np.float32))
    multiplier = -scipy.stats.norm.ppf(tail_mass / 2)
    pmf_center = torch.ceil(st * multiplier).int()
    pmf_length = 2 * pmf_center + 1
    max_length = int(pmf_length.max())
    samples = torch.abs(torch.arange(max_length).int() - pmf_center[:, None]).float()
    s = st.unsqueeze(1).float()

    def cum(v):                                   # _standardized_cumulative :578-582
        return 0.5 * torch.erfc(float(-(2 ** -0.5)) * v)

    upper = cum((0.5 - samples) / s)
    lower = cum((-0.5 - samples) / s)
    pmf = upper - lower
    tail = 2 * lower[:, :1]
    cdf = _pmf_to_cdf(pmf.numpy(), tail.numpy(), pmf_length.numpy(), max_length)
    return CdfTables(cdf, (pmf_length + 2).numpy(), (-pmf_center).numpy())


def entropy_bottleneck_tables(sd, prefix="entropy_bottleneck") -> CdfTables:
    """EntropyBottleneck.update (entropy_models.py:354-393, _logits_cumulative :400-419)."""
    import torch
    import torch.nn.functional as F
    g = lambda k: torch.as_tensor(np.asarray(sd[f"{prefix}.{k}"])).float().cpu()
    q = g("quantiles")
    medians = q[:, 0, 1]
    minima = torch.clamp(torch.ceil(medians - q[:, 0, 0]).int(), min=0)
    maxima = torch.clamp(torch.ceil(q[:, 0, 2] - medians).int(), min=0)
    pmf_start = medians - minima
    pmf_length = maxima + minima + 1
    max_length = int(pmf_length.max())
    samples = torch.arange(max_length)[None, :] + pmf_start[:, None, None]

    def logits(x):
        for i in range(5):
            x = torch.matmul(F.softplus(g(f"_matrix{i}")), x)
            x = x + g(f"_bias{i}")
            if i < 4:
                x = x + torch.tanh(g(f"_factor{i}")) * torch.tanh(x)
        return x

    lower, upper = logits(samples - 0.5), logits(samples + 0.5)
    sign = -torch.sign(lower + upper)
    pmf = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))[:, 0, :]
    tail = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
    cdf = _pmf_to_cdf(pmf.numpy(), tail.numpy(), pmf_length.numpy(), max_length)
    return CdfTables(cdf, (pmf_length + 2).numpy(), (-minima).numpy())


def measure_rans_rates(sym, idx, t: CdfTables, n_threads, streams_per_call=32, sample=48, reps=3):
    """Host rANS coder rates on real symbol planes ("rANS: report Msym/s per stream and streams in flight").
    sym, idx: [n_streams][n] int32 as the encoder chain produced them (one stream per image and slice).  Measures
      * one stream at a time on one host thread: pc_rans_encode_with_indexes / pc_rans_decode_with_indexes (the serial per-stream rate:
        a stream is one chain of dependent state updates, rans_interface.cpp:99-275) and the decoder's fast form on two streams in lock
        step (pc_rans_decode_batch_u8, one thread);
      * the process's pool coding `streams_per_call` streams per call on `n_threads` threads, as one slice step of the codec does.
    Returns a dict of Msym/s figures.  Used by bench.py and tools/host_pool_scale.py; pure host code."""
    import time
    L = lib()
    sym = np.ascontiguousarray(sym, np.int32)
    idx = np.ascontiguousarray(idx, np.int32)
    ns, n = sym.shape
    pick = np.unique(np.linspace(0, ns - 1, min(sample, ns)).astype(int))
    cap = L.pc_rans_bound(n)
    buf = np.empty(cap, np.uint8)
    ln = C.c_size_t(0)
    tab = (_ptr(t.cdf), t.cdf.shape[0], t.cdf.shape[1], _ptr(t.length), _ptr(t.offset))
    enc = {}
    t_enc = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        for s in pick:
            check(L.pc_rans_encode_with_indexes(_ptr(sym[s]), _ptr(idx[s]), n, *tab, _ptr(buf), cap, C.byref(ln)), "rans_encode")
            enc[int(s)] = buf[: ln.value].tobytes()
        t_enc = min(t_enc, time.perf_counter() - t0)
    out = np.empty(n, np.int32)
    t_dec = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        for s in pick:
            e = np.frombuffer(enc[int(s)], np.uint8)
            check(L.pc_rans_decode_with_indexes(_ptr(e), e.size, _ptr(idx[s]), n, *tab, _ptr(out)), "rans_decode")
        t_dec = min(t_dec, time.perf_counter() - t0)
    assert np.array_equal(out, sym[pick[-1]])
    # pool: one slice step = streams_per_call streams per call
    k = min(streams_per_call, ns)
    s0 = (ns // 2) // k * k if ns >= 2 * k else 0                 # a slice from the middle of the chain (an enhancement slice if there is one)
    sb, ib = sym[s0:s0 + k], idx[s0:s0 + k]
    ob = np.empty((k, cap), np.uint8)
    lens = (C.c_size_t * k)()
    t_pe = 1e30
    for _ in range(reps + 2):
        t0 = time.perf_counter()
        check(L.pc_rans_encode_batch(_ptr(sb), _ptr(ib), k, n, *tab, _ptr(ob), cap, lens, n_threads), "rans_encode_batch")
        t_pe = min(t_pe, time.perf_counter() - t0)
    ptrs = (C.c_void_p * k)(*[ob[i].ctypes.data for i in range(k)])
    i8 = np.ascontiguousarray(ib.astype(np.uint8))
    ob2 = np.empty((k, n), np.int32)
    t_pd, t_1d = 1e30, 1e30
    for _ in range(reps + 2):
        t0 = time.perf_counter()
        check(L.pc_rans_decode_batch_u8(ptrs, lens, k, _ptr(i8), n, *tab, _ptr(ob2), n_threads), "rans_decode_batch_u8")
        t_pd = min(t_pd, time.perf_counter() - t0)
    assert np.array_equal(ob2, sb)
    k2 = min(2, k)
    for _ in range(reps + 2):
        t0 = time.perf_counter()
        check(L.pc_rans_decode_batch_u8(ptrs, lens, k2, _ptr(i8), n, *tab, _ptr(ob2), 1), "rans_decode_batch_u8")
        t_1d = min(t_1d, time.perf_counter() - t0)
    coded_bytes = sum(len(v) for v in enc.values())
    return {"symbols_per_stream": int(n), "streams_sampled": int(len(pick)),
            "bits_per_symbol_sampled": round(8.0 * coded_bytes / (len(pick) * n), 3),
            "encode_msym_s_per_stream": round(len(pick) * n / t_enc / 1e6, 1),
            "decode_msym_s_per_stream": round(len(pick) * n / t_dec / 1e6, 1),
            "decode_fast_msym_s_per_stream_two_in_lock_step": round(n / t_1d / 1e6, 1),
            "decode_fast_msym_s_per_thread": round(k2 * n / t_1d / 1e6, 1),
            "pool_threads": int(n_threads), "streams_in_flight": int(k),
            "pool_encode_msym_s": round(k * n / t_pe / 1e6, 1), "pool_decode_msym_s": round(k * n / t_pd / 1e6, 1),
            "pool_encode_ms_per_slice_step": round(1e3 * t_pe, 3), "pool_decode_ms_per_slice_step": round(1e3 * t_pd, 3)}
