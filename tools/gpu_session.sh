# round 6, session 14: counters of the periodic row kernel, packed build
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_IFETCH"
rm -rf gpurun_out/pmc_rows_*
(cd /tmp && MKAMD_DIST_ONLY=periodic timeout 300 rocprofv3 --kernel-trace --pmc $SQ1 --output-format csv -d $R/gpurun_out/pmc_rows_sq1 -- python $R/bench.py --workload dist --no-cpu-baseline --settle-seconds 0 --steps 8 --warmup 2 > $R/gpurun_out/pmc_rows_sq1.log 2>&1)
(cd /tmp && MKAMD_DIST_ONLY=periodic timeout 300 rocprofv3 --kernel-trace --pmc $SQ2 --output-format csv -d $R/gpurun_out/pmc_rows_sq2 -- python $R/bench.py --workload dist --no-cpu-baseline --settle-seconds 0 --steps 8 --warmup 2 > $R/gpurun_out/pmc_rows_sq2.log 2>&1)
python tools/pmc_summary.py gpurun_out/pmc_rows_sq1 2>&1 | grep dist_rows
python tools/pmc_summary.py gpurun_out/pmc_rows_sq2 2>&1 | grep dist_rows
find gpurun_out -name "*_kernel_trace.csv" -size +1M -delete
