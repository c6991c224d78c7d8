# tools/gpu_pmc_bin2.sh -- where k_bin_count's cycles go (cfg2, in order): active cycles per instruction class, two PMC passes
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAVES"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pb2_$i
  (MKAMD_DIAG=1 timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pb2_$i -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra --no-pipeline > $R/gpurun_out/pb2_$i.log 2>&1)
done
cd $R
python tools/pmc_summary.py gpurun_out/pb2_1 gpurun_out/pb2_2 | grep -A0 "k_bin_count\|k_bin_fill"
