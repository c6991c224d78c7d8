#!/bin/bash
# tools/gpu_r4_xtc_pmc.sh -- what the device XTC decoder's kernels spend their time on: instruction counts and wait cycles per
# launch (k_xtc_scan: a lane walks a frame; k_xtc_expand: a thread decodes a group), one probe file per pass
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/xtc_pmc; rm -rf $O; mkdir -p $O
for which in syn real; do
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_MISC SQ_IFETCH"; do
  n=${which}_$(echo $set | cut -d' ' -f1)
  XTC_PROBE_ONLY=$which timeout 300 rocprofv3 --pmc $set -d $O/$n -o p --output-format csv -- python $R/tools/xtc_gpu_probe.py > $O/$n.log 2>&1
  python - <<PY
import csv, glob, collections
for f in glob.glob("$O/$n/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "xtc" not in k: continue
        key = ("$which", k.split("(")[0], "grid " + r["Grid_Size"])
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(key, r["Counter_Name"])] += 1
    for key, d in sorted(acc.items()):
        print(key, {c: round(v / cnt[(key, c)], 1) for c, v in d.items()})
PY
done
done 2>&1 | tee $O/summary.txt
