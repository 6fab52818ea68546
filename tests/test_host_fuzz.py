"""CPU sanitizer build + malformed-stream fuzz of the host entropy coder (SURVEY.md section 5 "Race detection /
sanitizers").  csrc/pc_host.cpp parses untrusted byte strings (rans_decode_core, pc_rans_decode_batch_u8, the decoder behind
pc_codec_decompress_packed) and replaces a reference coder whose only guard is `assert` (rans_interface.cpp:110-111, 170-172).  It is
compiled alone with g++ -fsanitize=address,undefined (`make -C progressivecodec_amd/csrc host-asan`; no HIP header is involved) and
tools/host_fuzz.py drives the known-answer vectors, round trips through every decoder form and >= 10^4 mutated / truncated /
index-corrupted streams through it in a child process with the ASan runtime preloaded: every call returns PC_OK or a PC_ERR_* code, a
sanitizer report aborts the child.  GPU sanitizers are not available on the pool, so this is the CPU-build-only coverage the survey asks for."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "progressivecodec_amd", "csrc")


def _asan_runtime():
    gcc = shutil.which("gcc")
    if not gcc or not shutil.which("g++"):
        return None
    p = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_host_coder_under_asan_ubsan_survives_10k_malformed_streams():
    rt = _asan_runtime()
    if rt is None:
        pytest.skip("no g++ / libasan in this environment")
    subprocess.check_call(["make", "-C", CSRC, "host-asan"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "host_fuzz.py"), "--cases", "12000", "--seed", "4"], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, f"sanitizer report or failed assertion:\n{r.stderr[-3000:]}"
    j = json.loads(r.stdout.strip().splitlines()[-1])
    assert j["mutated_streams"] >= 10000 and j["sanitizer_reports"] == 0
    codes = j["return_codes"]
    # the fuzz really reaches the error paths: truncated streams, bad indexes, malformed tables, small buffers -- as CODES, never as reports
    assert codes["decode_with_indexes"].get("-4", 0) > 1000 and codes["decode_with_indexes"].get("-2", 0) > 500
    assert codes["decode_batch_u8"].get("-4", 0) > 1000 and codes["decode_u8_malformed_tables"].get("-5", 0) > 100
    assert codes["encode_small_buffer"] == {"-3": 600} and codes["encode_malformed"].get("-5", 0) > 100
    assert codes["decode_stream_bad_state"].get("-1", 0) > 1000


def test_host_only_build_matches_the_product_library_on_the_kats():
    """the plain host-only build (tools/host_decode_rate.py measures on it) is the same coder: KATs of SURVEY.md section 8c"""
    if not shutil.which("g++"):
        pytest.skip("no g++")
    import ctypes as C
    import numpy as np
    subprocess.check_call(["make", "-C", CSRC, "host"], stdout=subprocess.DEVNULL)
    L = C.CDLL(os.path.join(ROOT, "progressivecodec_amd", "libpc_host.so"))
    cdf, ln, off = np.array([[0, 8192, 57344, 61440, 65536]], np.int32), np.array([5], np.int32), np.array([-1], np.int32)
    sym, idx = np.array([0, 1, -1, 0, 7, -4], np.int32), np.zeros(6, np.int32)
    buf, k = np.zeros(64, np.uint8), C.c_size_t()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert L.pc_rans_encode_with_indexes(p(sym), p(idx), C.c_size_t(6), p(cdf), 1, 5, p(ln), p(off), p(buf), C.c_size_t(64), C.byref(k)) == 0
    assert buf[: k.value].tobytes().hex() == "a141ad217f1cc771"
