"""CPU-side tests of the product library: it loads, exports every symbol include/pcodec.h declares,
and its host entry points (entropy coder, pmf->CDF) reproduce the reference's known answers.
No GPU compute is called here."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from progressivecodec_amd import entropy
from progressivecodec_amd._lib import EXPORTS, PcodecError, lib
from tests.util import GOLD, ROOT, tables_npz


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "pcodec.h")).read()
    declared = set(re.findall(r"PC_API\s+[\w\s\*]+?\b(pc_\w+)\s*\(", hdr))
    assert declared == set(EXPORTS), declared ^ set(EXPORTS)
    L = lib()
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in L.pc_version()


def gc_tables():
    t = tables_npz()
    return entropy.CdfTables(t["gc_cdf"], t["gc_len"], t["gc_off"])


def test_rans_known_answers_from_reference():
    for k in json.load(open(os.path.join(GOLD, "kat_rans.json"))):
        if k.get("table") == "gc":
            t = gc_tables()
        else:
            t = entropy._tables_from_lists(k["cdfs"], k["sizes"], k["offsets"])
        enc = entropy.rans_encode(k["symbols"], k["indexes"], t)
        assert enc.hex() == k["encoded_hex"], k["name"]
        assert len(enc) % 4 == 0 and len(enc) >= 8
        dec = entropy.rans_decode(bytes.fromhex(k["encoded_hex"]), k["indexes"], t)
        assert dec.tolist() == k["symbols"], k["name"]


def test_drop_in_ans_surface():
    """compressai.ans.RansEncoder / RansDecoder call signature (rans_interface.cpp:352-372)."""
    s = entropy.RansEncoder().encode_with_indexes([0, 1, -1, 0, 7, -4], [0] * 6, [[0, 8192, 57344, 61440, 65536]], [5], [-1])
    assert s.hex() == "a141ad217f1cc771"
    assert entropy.RansDecoder().decode_with_indexes(s, [0] * 6, [[0, 8192, 57344, 61440, 65536]], [5], [-1]) == [0, 1, -1, 0, 7, -4]


def test_rans_matches_oracle_on_random_streams_and_batches():
    from oracle import liboracle as lo
    t = gc_tables()
    ot = lo.Tables(t.cdf, t.length, t.offset)
    rng = np.random.default_rng(1)
    st = tables_npz()["scale_table"]
    n_streams, n = 9, 3000
    idx = rng.integers(0, 64, (n_streams, n)).astype(np.int32)
    sym = np.rint(rng.standard_normal((n_streams, n)) * st[idx] * 1.3).astype(np.int32)
    sym[0, ::50] = rng.integers(-5000, 5000, sym[0, ::50].size)
    L = lib()
    stride = L.pc_rans_bound(n)
    out = np.zeros((n_streams, stride), np.uint8)
    lens = np.zeros(n_streams, np.uint64)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = L.pc_rans_encode_batch(P(sym), P(idx), n_streams, n, P(t.cdf), 64, t.cdf.shape[1], P(t.length), P(t.offset), P(out),
                                stride, P(lens), 0)
    assert rc == 0
    enc = [out[i, : int(lens[i])].tobytes() for i in range(n_streams)]
    for i in range(n_streams):
        assert enc[i] == lo.rans_encode(sym[i], idx[i], ot)
    ptrs = (C.c_char_p * n_streams)(*enc)
    ln = (C.c_size_t * n_streams)(*[len(e) for e in enc])
    dec = np.zeros((n_streams, n), np.int32)
    assert L.pc_rans_decode_batch(ptrs, ln, n_streams, P(idx), n, P(t.cdf), 64, t.cdf.shape[1], P(t.length), P(t.offset), P(dec), 0) == 0
    assert np.array_equal(dec, sym)
    # the decoder's fast form (byte indexes, start table, two streams per thread; odd stream count: the last one is alone)
    idx8 = idx.astype(np.uint8)
    for nt in (0, 1):
        dec8 = np.zeros((n_streams, n), np.int32)
        assert L.pc_rans_decode_batch_u8(ptrs, ln, n_streams, P(idx8), n, P(t.cdf), 64, t.cdf.shape[1], P(t.length), P(t.offset), P(dec8), nt) == 0
        assert np.array_equal(dec8, sym)
    short = (C.c_size_t * n_streams)(*[len(e) // 2 // 4 * 4 for e in enc])
    assert L.pc_rans_decode_batch_u8(ptrs, short, n_streams, P(idx8), n, P(t.cdf), 64, t.cdf.shape[1], P(t.length), P(t.offset), P(dec8), 0) == -4
    bad = idx8.copy(); bad[3, 5] = 64
    assert L.pc_rans_decode_batch_u8(ptrs, ln, n_streams, P(bad), n, P(t.cdf), 64, t.cdf.shape[1], P(t.length), P(t.offset), P(dec8), 0) == -2


def test_host_pool_serves_concurrent_callers():
    """An encoder and a decoder object of one process (bench.py's overlapped steps), or the lanes of one decoder, are inside the host
    pool's parallel_for at the same time: six threads issue batch encodes and decodes of different sizes concurrently; every result
    must equal the single-caller result."""
    import threading
    t = gc_tables()
    st = tables_npz()["scale_table"]
    L = lib()
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    errors = []

    def caller(seed):
        try:
            rng = np.random.default_rng(seed)
            for it in range(12):
                n_streams, n = int(rng.integers(2, 40)), int(rng.integers(50, 1500))
                idx = rng.integers(0, 64, (n_streams, n)).astype(np.int32)
                sym = np.rint(rng.standard_normal((n_streams, n)) * st[idx]).astype(np.int32)
                stride = L.pc_rans_bound(n)
                out = np.zeros((n_streams, stride), np.uint8)
                lens = np.zeros(n_streams, np.uint64)
                assert L.pc_rans_encode_batch(P(sym), P(idx), n_streams, n, P(t.cdf), 64, t.cdf.shape[1], P(t.length), P(t.offset), P(out), stride, P(lens), 0) == 0
                enc = [out[i, : int(lens[i])].tobytes() for i in range(n_streams)]
                assert enc[0] == entropy.rans_encode(sym[0], idx[0], t) and enc[-1] == entropy.rans_encode(sym[-1], idx[-1], t)
                ptrs = (C.c_char_p * n_streams)(*enc)
                ln = (C.c_size_t * n_streams)(*[len(e) for e in enc])
                dec = np.zeros((n_streams, n), np.int32)
                assert L.pc_rans_decode_batch_u8(ptrs, ln, n_streams, P(idx.astype(np.uint8)), n, P(t.cdf), 64, t.cdf.shape[1], P(t.length), P(t.offset), P(dec), 0) == 0
                assert np.array_equal(dec, sym)
        except BaseException as e:           # noqa: BLE001  (reported by the main thread)
            errors.append(e)

    threads = [threading.Thread(target=caller, args=(s,)) for s in range(6)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
    assert not any(th.is_alive() for th in threads), "a caller is stuck in the pool"
    assert not errors, errors


def test_rans_edge_cases_and_errors():
    t = gc_tables()
    # empty input: the reference asserts / UB (rans_interface.cpp:170-172); we emit the 8-byte flush of the initial state
    e = entropy.rans_encode([], [], t)
    assert len(e) == 8 and entropy.rans_decode(e, [], t).size == 0
    # n = 1
    e1 = entropy.rans_encode([3], [10], t)
    assert entropy.rans_decode(e1, [10], t).tolist() == [3]
    with pytest.raises(PcodecError) as ei:
        entropy.rans_encode([0], [64], t)
    assert ei.value.code == -2
    with pytest.raises(PcodecError) as ei:
        entropy.rans_decode(b"\x00" * 4, [0], t)
    assert ei.value.code == -4
    big = entropy.rans_encode(np.arange(-3000, 3000), np.zeros(6000, np.int32), t)    # all bypass at index 0
    assert np.array_equal(entropy.rans_decode(big, np.zeros(6000, np.int32), t), np.arange(-3000, 3000))
    with pytest.raises(PcodecError) as ei:
        entropy.rans_decode(big[: len(big) // 2], np.zeros(6000, np.int32), t)
    assert ei.value.code == -4


def test_pmf_to_quantized_cdf_known_answers():
    t = tables_npz()
    for key in [k[4:] for k in t if k.startswith("pmf_")]:
        got = np.array(entropy.pmf_to_quantized_cdf(t["pmf_" + key], 16), np.uint32)
        assert np.array_equal(got, t["cdf_" + key]), key
    assert entropy.pmf_to_quantized_cdf([0.1, 0.7, 0.15, 0.05], 16) == [0, 6554, 52429, 62259, 65536]
    with pytest.raises(PcodecError):
        entropy.pmf_to_quantized_cdf([0.0, 0.0], 16)


def test_tables_rebuilt_by_update_equal_the_reference_tables():
    """GaussianConditional.update / EntropyBottleneck.update restated (entropy.py) == the reference's buffers."""
    from progressivecodec_amd.synth import synthetic_state_dict
    sd = {k: v.numpy() for k, v in synthetic_state_dict().items()}
    t = tables_npz()
    g = entropy.gaussian_conditional_tables(sd["gaussian_conditional.scale_table"])
    assert np.array_equal(g.cdf, t["gc_cdf"]) and np.array_equal(g.length, t["gc_len"]) and np.array_equal(g.offset, t["gc_off"])
    e = entropy.entropy_bottleneck_tables(sd)
    assert np.array_equal(e.cdf, t["eb_cdf"]) and np.array_equal(e.length, t["eb_len"]) and np.array_equal(e.offset, t["eb_off"])
    assert g.cdf.shape == (64, 3133) and g.length.min() == 5 and g.length.max() == 3133 and g.offset.min() == -1565


def test_codec_needs_a_gpu_and_fails_loudly_without_one():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    assert lib().pc_codec_create(C.byref(h), 0) == -6          # PC_ERR_HIP: no device, no CPU fallback
    from progressivecodec_amd import ChannelProgresssiveWACNN
    with pytest.raises((PcodecError, RuntimeError)):
        ChannelProgresssiveWACNN(device="cuda:0")
    with pytest.raises(RuntimeError):
        ChannelProgresssiveWACNN(device="cpu")


def test_arch_spec_matches_reference_state_dict_layout():
    from progressivecodec_amd.arch import param_spec
    spec = param_spec()
    assert len(spec) == 1019
    n_params = sum(int(np.prod(s)) for k, (s, d, kind) in spec.items()
                   if kind not in ("table", "pedestal", "beta_bound", "gamma_bound", "relpos_index", "eb_target",
                                   "likelihood_bound", "scale_bound", "scale_table"))
    assert n_params == 152137398                                 # authors' log, SURVEY.md section 6


def eb_tables():
    t = tables_npz()
    return entropy.CdfTables(t["eb_cdf"], t["eb_len"], t["eb_off"])


def test_streaming_ans_surface_known_answers_from_reference():
    """BufferedRansEncoder.encode_with_indexes (several calls, different tables) + flush and RansDecoder.set_stream +
    decode_stream (rans_interface.cpp:99-191, 277-350) against vectors made by the reference's own module
    (tests/golden/make_golden_rans_stream.py): same bytes, same symbols call by call."""
    tabs = {"gc": gc_tables(), "eb": eb_tables()}
    for k in json.load(open(os.path.join(GOLD, "kat_rans_stream.json"))):
        enc = entropy.BufferedRansEncoder()
        for ch in k["chunks"]:
            assert enc.encode_with_indexes(ch["symbols"], ch["indexes"], tabs[ch["table"]], None, None) is None
        data = enc.flush()
        assert data.hex() == k["encoded_hex"], k["name"]
        assert enc.flush() == entropy.rans_encode([], [], gc_tables())          # flushed: the encoder is empty again
        dec = entropy.RansDecoder()
        dec.set_stream(bytes.fromhex(k["encoded_hex"]))
        for ch in k["chunks"]:
            assert dec.decode_stream(ch["indexes"], tabs[ch["table"]], None, None) == ch["symbols"], k["name"]
    # list-of-lists tables (the reference's calling convention) and the error paths
    enc = entropy.BufferedRansEncoder()
    enc.encode_with_indexes([0, 1, -1], [0] * 3, [[0, 8192, 57344, 61440, 65536]], [5], [-1])
    enc.encode_with_indexes([0, 7, -4], [0] * 3, [[0, 8192, 57344, 61440, 65536]], [5], [-1])
    assert enc.flush().hex() == "a141ad217f1cc771"                              # == the one-call KAT of SURVEY.md section 8c
    with pytest.raises(ValueError):
        entropy.RansDecoder().decode_stream([0], gc_tables(), None, None)       # no stream set
    d = entropy.RansDecoder()
    d.set_stream(b"\x00" * 4)
    with pytest.raises(PcodecError) as ei:
        d.decode_stream([0], gc_tables(), None, None)
    assert ei.value.code == -4


def test_host_pool_plan_divides_cpus_among_local_ranks():
    """pc_host_pool_plan: threads = min(16, allowed CPUs / local ranks); a rank's slice starts at its own first CPU (run in
    subprocesses: the plan reads LOCAL_WORLD_SIZE / LOCAL_RANK from the environment)."""
    import subprocess
    import sys
    code = ("import ctypes as C; from progressivecodec_amd._lib import lib; n, f, a = C.c_int(), C.c_int(), C.c_int();"
            "lib().pc_host_pool_plan(C.byref(n), C.byref(f), C.byref(a)); print(n.value, f.value, a.value)")
    def plan(**env):
        e = dict(os.environ, **{k: str(v) for k, v in env.items()})
        e.pop("PC_HOST_THREADS", None)
        return tuple(map(int, subprocess.check_output([sys.executable, "-c", code], env=e, cwd=ROOT).split()))
    n1, f1, a = plan(LOCAL_WORLD_SIZE=1, LOCAL_RANK=0)
    assert n1 == min(16, a) and a >= 1
    if a >= 2:
        n2a, f2a, _ = plan(LOCAL_WORLD_SIZE=2, LOCAL_RANK=0)
        n2b, f2b, _ = plan(LOCAL_WORLD_SIZE=2, LOCAL_RANK=1)
        assert n2a == n2b == min(16, a // 2) and f2a != f2b
    n8, _, _ = plan(LOCAL_WORLD_SIZE=8, LOCAL_RANK=3)
    assert n8 == max(1, min(16, a // 8))
