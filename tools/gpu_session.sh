# tools/gpu_session.sh -- the commands of the CURRENT gpurun session
# round 5, session 16: k_frames_to_items with non-temporal loads / stores (read once, written once: it runs beside the tile kernel, whose records
# live in the L2) against the committed build, alternating: the streamed trajectory driver's steady ms per call
mkdir -p gpurun_out
export TMPDIR=/tmp
V=$PWD/.variants/libmkamd_base.so
cat > /tmp/stream_once.py <<'PY'
import sys, os, json
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch, bench
from moleculekit_amd import _lib
ctx = _lib.default_context(0); dev = torch.device("cuda", 0)
for i in range(2):
    r = bench.bench_stream_cfg4(ctx, dev, 0.914, frames=8192, chunk=256)
print(os.environ.get("TAGX"), r["steady_ms_per_call"], r["ms_per_call"], r["frames_per_s"])
PY
for r in 1 2 3; do
  TAGX=base MKAMD_LIB=$V MKAMD_ALLOW_DIAGNOSTICS=1 timeout 200 python /tmp/stream_once.py 2>&1 | tail -1
  TAGX=new timeout 200 python /tmp/stream_once.py 2>&1 | tail -1
done
