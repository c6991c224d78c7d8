mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for wl in cfg1 cfg2; do
rm -rf $R/gpurun_out/pmcl_$wl
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pmcl_$wl -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-pipeline --workload $wl > $R/gpurun_out/pmcl_$wl.log 2>&1
done
cd $R
python tools/pmc_summary.py gpurun_out/pmcl_cfg1 gpurun_out/pmcl_cfg2 | grep -i "voxelize_tiles<8" | cut -c1-500
