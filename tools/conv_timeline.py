#!/usr/bin/env python3
"""Per-wave timeline of one launch of the unified conv kernel (PC_CONV_DBG=64 build: s_memrealtime stamps + HW_ID / XCC_ID).
usage: PC_CONV_DBG=64 python tools/conv_timeline.py <shape name from conv_tune.SHAPES> [more shapes]
Prints, per shape: kernel span, blocks per CU, how many blocks are resident per CU over time, the K-loop duration of a wave by
its arrival order on the CU, and the full timeline of two CUs."""
import ctypes as C
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from progressivecodec_amd._lib import check, lib
from tools.conv_tune import SHAPES


def main():
    L = lib()
    rng = np.random.default_rng(0)
    for name, B, H, W, ci, co, k, s in SHAPES:
        if name not in sys.argv[1:]:
            continue
        x = torch.from_numpy(rng.standard_normal((B, H, W, ci)).astype(np.float32)).cuda()
        w = torch.from_numpy((rng.standard_normal((k * k, ci, co)) * 0.05).astype(np.float32)).cuda()
        b = torch.zeros(co, device="cuda")
        Ho, Wo = (H + 2 * (k // 2) - k) // s + 1, (W + 2 * (k // 2) - k) // s + 1
        out = torch.empty((B, Ho, Wo, co), device="cuda")
        P = lambda t: C.c_void_p(t.data_ptr())
        for _ in range(3):
            check(L.pc_conv2d_nhwc(P(x), B, H, W, ci, P(w), P(b), 0, co, k, s, int(os.environ.get('PC_TUNE_ACT', '0')), 0, P(out), None))
        torch.cuda.synchronize()
        st = np.zeros((8192, 16), np.uint64)
        check(L.pc_debug_read_stamps(st.ctypes.data_as(C.c_void_p), 8192))
        st = st[st[:, 3] > 0]
        t0 = st[:, 0].min()
        T = (st[:, :4].astype(np.int64) - int(t0)) / 100.0            # microseconds
        hw, xcc = st[:, 4].astype(np.int64), st[:, 5].astype(np.int64) & 15
        simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
        cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        blk, wave_nl = st[:, 7].astype(np.int64), st[:, 6].astype(np.int64)
        print(f"== {name}: {len(st)} waves stamped, kernel span {T[:, 3].max():.1f} us, distinct CUs {len(set(cuid))}")
        per_cu = defaultdict(set)
        for c, bk in zip(cuid, blk):
            per_cu[c].add(bk)
        hist = defaultdict(int)
        for c, v in per_cu.items():
            hist[len(v)] += 1
        print("   blocks per CU:", dict(sorted(hist.items())))
        # residency over time (blocks with start <= t < end on the CU), sampled
        ts = np.linspace(0, T[:, 3].max(), 21)[:-1]
        res = []
        for t in ts:
            n = 0
            for c in per_cu:
                m = cuid == c
                n += len(set(blk[m & (T[:, 0] <= t) & (T[:, 3] > t)]))
            res.append(n / max(1, len(per_cu)))
        print("   resident blocks per CU over time:", " ".join(f"{r:.1f}" for r in res))
        # K-loop duration by arrival order of the block on its CU (live waves only)
        order_dur = defaultdict(list)
        for c in per_cu:
            m = (cuid == c) & (wave_nl > 0)
            starts = sorted(set(zip(T[m, 0].round(2), blk[m])), key=lambda z: z[0])
            seen, rank = {}, 0
            for t, bk in starts:
                if bk not in seen:
                    seen[bk] = rank; rank += 1
            for i in np.nonzero(m)[0]:
                order_dur[seen[blk[i]]].append((T[i, 1] - T[i, 0], T[i, 2] - T[i, 1], T[i, 3] - T[i, 2]))
        for r in sorted(order_dur):
            a = np.array(order_dur[r])
            print(f"   block #{r} on its CU: prologue {a[:,0].mean():6.1f} us  K loop {a[:,1].mean():7.1f} us (min {a[:,1].min():.1f} max {a[:,1].max():.1f})  epilogue {a[:,2].mean():6.1f} us  [{len(a)} waves]")
        for c in list(per_cu)[:int(os.environ.get('PC_TIMELINE_CUS', '2'))]:
            m = cuid == c
            print(f"   CU {c}:")
            for i in sorted(np.nonzero(m)[0], key=lambda i: (T[i, 0], blk[i], simd[i])):
                print(f"      block {blk[i]:5d} simd {simd[i]} live {wave_nl[i]}: start {T[i,0]:7.1f} loop {T[i,1]:7.1f} .. {T[i,2]:7.1f} end {T[i,3]:7.1f}")


if __name__ == "__main__":
    main()
