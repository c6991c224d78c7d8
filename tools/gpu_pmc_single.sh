# tools/gpu_pmc_single.sh -- PMC counters of the kernels of ONE cfg2 grid per call (tools/single_timeline.py)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${1:-cfg2}
cd /tmp
run() { name=$1; shift; rm -rf $R/gpurun_out/pmcs_$name; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmcs_$name -- python $R/tools/single_timeline.py $W > $R/gpurun_out/pmcs_$name.log 2>&1; echo "rc=$?" >> $R/gpurun_out/pmcs_$name.log; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM
cd $R
python tools/pmc_summary.py gpurun_out/pmcs_sq1 gpurun_out/pmcs_sq2 2>&1 | grep -v "^gpurun_out" | cut -c1-600
