# tools/gpu_session.sh -- the commands of the CURRENT gpurun session
# round 5, session 18: counters of k_dist_frame on the 300 x 30 x 2 048 call (what bounds it at 2.3 TB/s: it is insensitive to its
# instruction count and to how many blocks it is cut into)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/frame_once.py <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import numpy as np, torch
from moleculekit_amd import _lib
dev = torch.device("cuda", 0)
N, F = 30000, 2048
rng = np.random.default_rng(4)
coords = torch.rand((N, 3, F), device=dev) * 66.9
box = torch.full((3, F), 66.9, device=dev)
chains = torch.as_tensor((np.arange(N) // 1000).astype(np.int32), device=dev)
ctx = _lib.default_context(0); ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
s1 = torch.as_tensor(np.sort(rng.choice(N, 300, replace=False)).astype(np.int32), device=dev)
s2 = torch.as_tensor(np.sort(rng.choice(N, 30, replace=False)).astype(np.int32), device=dev)
out = torch.empty((F, 9000), device=dev)
for pbc in (False, True):
    for _ in range(6):
        ctx.dist_trajectory_dev(coords.data_ptr(), F, box.data_ptr(), s1.data_ptr(), 300, s2.data_ptr(), 30, chains.data_ptr(), False, pbc, False, out.data_ptr())
torch.cuda.synchronize()
PY
rm -rf gpurun_out/pmc_frame_*
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_IFETCH" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_ANY" "WRITE_SIZE" "FETCH_SIZE"; do
  n=$(echo $grp | cut -d' ' -f1)
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_frame_$n -- python /tmp/frame_once.py > $R/gpurun_out/pmc_frame_$n.log 2>&1)
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_frame_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_dist_frame" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][-45:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c, v in sorted(acc[k].items()): print("   %-24s %14.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
