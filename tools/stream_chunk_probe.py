import sys, os, json
sys.path.insert(0, "/root/repo")
import torch, bench
from moleculekit_amd import _lib
ctx = _lib.default_context(0); dev = torch.device("cuda", 0)
for frames, chunk in ((4096, 256), (8192, 512), (8192, 1024), (16384, 2048)):
    r = bench.bench_stream_cfg4(ctx, dev, 1.066, frames=frames, chunk=chunk)
    print(chunk, json.dumps(r)[:700])
