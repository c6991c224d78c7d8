# tools/gpu_session.sh -- the commands of the CURRENT gpurun session
# round 5, session 20: k_frames_to_items as one-wave-pair workgroups over 32 x 32 tiles (4.2 KB of LDS: fits beside the tile kernel's waves)
# against the 64 x 64 form (16.6 KB), a trial knob: alone, and inside the streamed trajectory driver
mkdir -p gpurun_out
export TMPDIR=/tmp
(timeout 900 python -m pytest tests/test_gpu_api.py -m gpu -q -x -k "stream or frames or traject" 2>&1 | tail -2)
cat > /tmp/stream_once.py <<'PY'
import sys, os, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, bench
from moleculekit_amd import _lib
ctx = _lib.default_context(0); dev = torch.device("cuda", 0)
# the kernel alone: 256 frames x 30 000 atoms
src = torch.rand((90000, 2048), device=dev); dst = torch.empty((256, 90000), device=dev)
st = torch.cuda.current_stream(dev).cuda_stream
for _ in range(5): ctx.frames_to_items_dev(st, src.data_ptr(), 90000, 2048, 256, 1.0, dst.data_ptr())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): ctx.frames_to_items_dev(st, src.data_ptr(), 90000, 2048, 256, 1.0, dst.data_ptr())
e1.record(); torch.cuda.synchronize()
alone = e0.elapsed_time(e1) / 50
ok = bool(torch.equal(dst, src[:, :256].t().contiguous()))
for i in range(2):
    r = bench.bench_stream_cfg4(ctx, dev, 0.914, frames=8192, chunk=256)
print(os.environ.get("MKAMD_F2I_TILE"), "alone %.1f us (%.2f TB/s) correct=%s" % (alone * 1e3, 2 * 90000 * 256 * 4 / alone / 1e9, ok), "stream steady", r["steady_ms_per_call"], "whole", r["ms_per_call"])
PY
for r in 1 2 3; do
  MKAMD_F2I_TILE=64 timeout 200 python /tmp/stream_once.py 2>&1 | tail -1
  MKAMD_F2I_TILE=32 timeout 200 python /tmp/stream_once.py 2>&1 | tail -1
done
