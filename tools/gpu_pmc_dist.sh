# tools/gpu_pmc_dist.sh -- instructions per wave, VALU-busy and wait share of k_dist_pairs on the `--workload dist` bench
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rm -rf $R/gpurun_out/pd
(timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pd -- python $R/bench.py --workload dist --no-cpu-baseline > $R/gpurun_out/pd.log 2>&1)
cd $R
python tools/pmc_summary.py gpurun_out/pd | grep -A1 "k_dist_pairs\|csv"
