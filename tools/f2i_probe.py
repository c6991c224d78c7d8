import sys, time
sys.path.insert(0, ".")
import numpy as np, torch, json
import bench
from moleculekit_amd import _lib
ctx = _lib.default_context(0); dev = torch.device("cuda", 0)
N, F, n = 30000, 2048, 256
src = torch.rand((N, 3, F), device=dev)
dst = torch.empty((n, N, 3), device=dev)
s = torch.cuda.current_stream(dev)
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: ctx.frames_to_items_dev(s.cuda_stream, src.data_ptr() + 4 * 512, 3 * N, F, n, 10.0, dst.data_ptr()))
ref = (src[:, :, 512:512 + n].permute(2, 0, 1) * 10.0).contiguous()
print("k_frames_to_items: %.4f ms  %.0f GB/s (read + write)  equal to torch: %s" % (ms, 2 * n * N * 3 * 4 / ms / 1e6, torch.equal(dst, ref)))
print("torch permute copy: %.4f ms" % t(lambda: dst.copy_(src[:, :, 512:512 + n].permute(2, 0, 1))))
for i in range(2): print(json.dumps({k: v for k, v in bench.bench_stream_cfg4(ctx, dev, None).items() if k in ("ms_per_call", "steady_ms_per_call", "frames_per_s")}))
