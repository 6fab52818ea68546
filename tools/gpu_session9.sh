#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
PC_TIMING=1 PC_PROFILE_CSV=$O/r02_c_conv_launches_bench_b32.csv timeout -k 10 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/r02_timing.log 2>&1
grep "pcodec\]" $O/r02_timing.log | tail -8
python - <<'PY' 2>&1 | tail -30
import time, torch, sys, ctypes as C
sys.path.insert(0,'.')
from progressivecodec_amd import ChannelProgresssiveWACNN
from progressivecodec_amd.synth import synthetic_state_dict
from progressivecodec_amd._lib import lib, check
net = ChannelProgresssiveWACNN(device="cuda:0"); net.load_state_dict(synthetic_state_dict()); net.update()
x = torch.rand(32,3,256,256, generator=torch.Generator().manual_seed(1)).cuda()
for _ in range(3):
    o = net.compress(x,0.5,"point-based-std"); d = net.decompress(o["strings"],o["shape"],0.5,"point-based-std")
torch.cuda.synchronize()
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    o = net.compress(x,0.5,"point-based-std"); d = net.decompress(o["strings"],o["shape"],0.5,"point-based-std")
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
PY
