# round 6, session 40: the XTC-fed leg at 2 048 frames per chunk with a small first chunk
mkdir -p gpurun_out
export TMPDIR=/tmp
cat > /tmp/xtcchunks.py <<'PY'
import sys, os, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, bench
from moleculekit_amd import _lib
ctx = _lib.default_context(0); dev = torch.device("cuda", 0)
for rep in range(3):
    for chunk, ramp in ((2048, 0), (2048, 256), (2048, 512), (2048, 1024)):
        r = bench.bench_xtc_cfg4(ctx, dev, 0.92, frames_gpu=16384, chunk_gpu=chunk, ramp_gpu=ramp)
        x = r["device_decode"]
        print("chunk", x["frames_per_call"], "ramp", ramp, "frames/s", x["frames_per_s"], "steady", x.get("steady_frames_per_s"), "busy", x["gpu_busy_fraction"], flush=True)
PY
timeout 900 python /tmp/xtcchunks.py 2>&1 | grep -v amdgpu | tail -12
