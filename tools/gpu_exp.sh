mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -E "Counter_Name" | grep -E "SQ_WAIT|SQ_INST_LEVEL|SQ_INSTS_VMEM|SQ_INSTS_SMEM|TCP_|TCC_HIT|TCC_MISS|TCC_REQ|SQ_INSTS_FLAT|SQ_INSTS_GDS|SQ_ACTIVE_INST|SQ_WAVE_CYCLES|SQ_LEVEL_WAVES|SQ_BUSY_CU" | sed 's/Counter_Name *:\t*//' | tr '\n' ' ' > $R/gpurun_out/pmc_names.txt
for wl in cfg5; do
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/pmcy_$wl -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pipeline --workload $wl > $R/gpurun_out/pmcy_$wl.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmcz_$wl -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pipeline --workload $wl > $R/gpurun_out/pmcz_$wl.log 2>&1
done
cd $R
python tools/pmc_summary.py gpurun_out/pmcy_cfg5 gpurun_out/pmcz_cfg5 | grep -i "voxelize_tiles<8" | cut -c1-700
cat gpurun_out/pmc_names.txt | cut -c1-1500
