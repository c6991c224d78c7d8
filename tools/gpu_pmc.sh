# tools/gpu_pmc.sh -- PMC counter passes over bench.py (cfg2), outputs under gpurun_out/pmc_*/
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|SQ_INST_LEVEL|SQ_WAVES|OCCUP" | head -40 > $R/gpurun_out/pmc_list.txt
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$name -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra $EXTRA > $R/gpurun_out/pmc_$name.log 2>&1; echo "rc=$?" >> $R/gpurun_out/pmc_$name.log; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_IFETCH
run sqc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES
run fetch FETCH_SIZE
run write WRITE_SIZE
cd $R
cat gpurun_out/pmc_list.txt | cut -c1-150 | head -20
tail -2 gpurun_out/pmc_sqc.log
