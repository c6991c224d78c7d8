# tools/gpu_r2d.sh -- VALU instruction counts of the tile kernel, fine vs coarse cells (PMC pass only: no tracing besides kernel-trace)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for mode in 0 1; do
  rm -rf $R/gpurun_out/pmcv_cells$mode
  (MKAMD_COARSE_CELLS=$mode timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VMEM --output-format csv -d $R/gpurun_out/pmcv_cells$mode -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra --no-pipeline > $R/gpurun_out/pmcv_cells$mode.log 2>&1; echo "rc=$?" >> $R/gpurun_out/pmcv_cells$mode.log)
done
cd $R
python - <<'PY'
import csv, glob, collections
for mode in (0,1):
    fs=sorted(glob.glob(f'gpurun_out/pmcv_cells{mode}/*/*counter_collection.csv'))
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[-1])):
        k=r['Kernel_Name'].split('(')[0].replace('void ','').strip()
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        if 'voxelize_tiles<8' in k or 'bin_' in k:
            big={c: max(x) for c,x in v.items()}
            print(mode, k[:40], {c: f"{val:.4g}" for c,val in big.items()})
PY
