# tools/gpu_session.sh -- the commands of the CURRENT gpurun session
# round 5, session 23: counters of k_dist_pairs on the 450 x 450 selfdist call (open and periodic), every launch alone
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/pairs_once.py <<'PY'
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
s = torch.as_tensor(np.sort(rng.choice(N, 450, replace=False)).astype(np.int32), device=dev)
out = torch.empty((F, 450 * 449 // 2), device=dev)
for pbc in (False, True):
    for _ in range(4):
        ctx.dist_trajectory_dev(coords.data_ptr(), F, box.data_ptr(), s.data_ptr(), 450, s.data_ptr(), 450, chains.data_ptr(), True, pbc, False, out.data_ptr())
torch.cuda.synchronize()
PY
rm -rf gpurun_out/pmc_pairs_*
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_IFETCH" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_WAIT_INST_ANY" "WRITE_SIZE" "FETCH_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $grp | cut -d' ' -f1)
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_pairs_$n -- python /tmp/pairs_once.py > $R/gpurun_out/pmc_pairs_$n.log 2>&1)
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob("gpurun_out/pmc_pairs_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_dist_pairs" in r["Kernel_Name"]:
            acc[r["Dispatch_Id"] if False else "k_dist_pairs"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    for c, v in sorted(acc[k].items()):
        h = len(v) // 2
        print("   %-28s open %14.0f   periodic %14.0f  (n=%d)" % (c, sum(v[:h]) / max(h, 1), sum(v[h:]) / max(len(v) - h, 1), len(v)))
PY
