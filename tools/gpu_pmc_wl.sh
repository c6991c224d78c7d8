# tools/gpu_pmc_wl.sh <workload> -- SQ counters of the tile kernel on another workload (one PMC pass)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
wl=$1
cd /tmp
rm -rf $R/gpurun_out/pmc_$wl
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/pmc_$wl -- python $R/bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --no-extra > $R/gpurun_out/pmc_$wl.log 2>&1
rm -rf $R/gpurun_out/pmc2_$wl
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc2_$wl -- python $R/bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline --no-extra > $R/gpurun_out/pmc2_$wl.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
for d in ("pmc_$wl","pmc2_$wl"):
    fs=sorted(glob.glob(f'gpurun_out/{d}/*/*counter_collection.csv'))
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[-1])):
        k=r['Kernel_Name'].split('(')[0].replace('void ','').strip()
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        if 'voxelize_tiles' in k and 'dense' not in k:
            print(k[:44], {c: f"{max(x):.4g}" for c,x in v.items()})
PY
grep -E "^\{" gpurun_out/pmc_$wl.log | cut -c1-200
