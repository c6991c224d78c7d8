"""tools/xtc_leg.py -- bench.py's xtc_cfg4 leg alone (device decode vs host decode), for several chunk sizes"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from moleculekit_amd import _lib
ctx = _lib.default_context(0); dev = torch.device("cuda", 0)
raw = float(sys.argv[1]) if len(sys.argv) > 1 else 1.066
for chunk_gpu in ((int(os.environ["XTC_LEG_ONE"]),) if os.environ.get("XTC_LEG_ONE") else (512, 1024, 2048)):       # XTC_LEG_ONE=<frames per chunk>
    r = bench.bench_xtc_cfg4(ctx, dev, raw, frames_gpu=8 * chunk_gpu, chunk_gpu=chunk_gpu)
    print(json.dumps(r))
