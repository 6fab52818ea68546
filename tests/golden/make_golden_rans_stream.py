#!/usr/bin/env python3
"""Known-answer vectors for the streaming surface of compressai.ans -- BufferedRansEncoder.encode_with_indexes (several calls) +
flush, RansDecoder.set_stream + decode_stream (several calls) -- produced by the reference's OWN module (built from
/root/reference/src/compress/cpp_exts/rans by oracle/build_ref.sh).  Run once in the build container:
    python3 tests/golden/make_golden_rans_stream.py        -> tests/golden/kat_rans_stream.json
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_env  # noqa: E402

ref_env.setup()
from compressai import ans  # noqa: E402

t = np.load(os.path.join(HERE, "tables.npz"))
gc = (t["gc_cdf"].tolist(), t["gc_len"].tolist(), t["gc_off"].tolist())
eb = (t["eb_cdf"].tolist(), t["eb_len"].tolist(), t["eb_off"].tolist())
rng = np.random.default_rng(11)
kats = []
for name, parts in (("gc_then_eb", [("gc", 700, 20), ("eb", 300, 190), ("gc", 64, 40)]),
                    ("bypass_across_calls", [("gc", 200, 3), ("gc", 200, 3)]),
                    ("single_call", [("eb", 128, 100)])):
    enc = ans.BufferedRansEncoder()
    chunks = []
    for tab, n, imax in parts:
        cdfs, lens, offs = gc if tab == "gc" else eb
        idx = rng.integers(0, imax + 1, n)
        sym = np.rint(rng.normal(0, 3.0 if tab == "gc" else 1.5, n)).astype(np.int64)
        if name == "bypass_across_calls":
            sym[::17] = rng.integers(-5000, 5000, sym[::17].size)
        enc.encode_with_indexes(sym.tolist(), idx.tolist(), cdfs, lens, offs)
        chunks.append(dict(table=tab, symbols=sym.tolist(), indexes=idx.tolist()))
    data = enc.flush()
    dec = ans.RansDecoder()
    dec.set_stream(data)
    for ch in chunks:
        cdfs, lens, offs = gc if ch["table"] == "gc" else eb
        assert dec.decode_stream(ch["indexes"], cdfs, lens, offs) == ch["symbols"]
    kats.append(dict(name=name, chunks=chunks, encoded_hex=data.hex()))
json.dump(kats, open(os.path.join(HERE, "kat_rans_stream.json"), "w"))
print("wrote", len(kats), "KATs", [len(k["encoded_hex"]) // 2 for k in kats])
