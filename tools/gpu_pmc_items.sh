# tools/gpu_pmc_items.sh [cfg3|cfg5] -- instructions per tile and busy cycles of k_voxelize_items (one PMC pass)
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=${1:-cfg3}
cd /tmp
rm -rf $R/gpurun_out/pi_$WL
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pi_$WL -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra --workload $WL > $R/gpurun_out/pi_$WL.log 2>&1
cd $R
python - $WL <<'PY'
import csv, glob, collections, sys, os
wl=sys.argv[1]
fs=sorted(glob.glob(f'gpurun_out/pi_{wl}/*/*counter_collection.csv'), key=os.path.getmtime)
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[-1])):
    k=r['Kernel_Name'].split('(')[0]
    if 'voxelize' in k or 'k_tail' in k or 'prepass_items' in k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
tiles={'cfg3':32768*27,'cfg5':65536*27}[wl]
for k,c in acc.items():
    w=max(c['SQ_WAVES']); i=c['SQ_WAVES'].index(w); g=lambda n: c[n][i]
    print(f"{k[-34:]:34s} waves {w:9.0f} VALU/tile {g('SQ_INSTS_VALU')/tiles:7.0f} SALU/tile {g('SQ_INSTS_SALU')/tiles:6.0f} LDS/tile {g('SQ_INSTS_LDS')/tiles:5.0f} VALU-busy quad/tile {g('SQ_ACTIVE_INST_VALU')/tiles:7.0f} busy Mcycles {g('SQ_BUSY_CYCLES')/32e6:6.3f} waitcnt share {g('SQ_WAIT_INST_ANY')/g('SQ_WAVE_CYCLES'):5.2f}")
PY
