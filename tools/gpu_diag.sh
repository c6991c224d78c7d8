# tools/gpu_diag.sh -- what each part of the tile kernel costs in VALU instructions and in time: diagnostic builds of the
# library with parts compiled out (-DMK_DIAG=bits, kernels.h), one PMC pass each (cfg2, in order).  bench.py's sanity
# assert is bypassed with MKAMD_DIAG=1 (the diagnostic builds produce garbage by design).
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for d in 0 1 3 4 11 27; do
  lib=$R/.variants/libmkamd_diag$d.so; [ $d = 0 ] && lib=$R/moleculekit_amd/csrc/libmkamd.so
  rm -rf $R/gpurun_out/diag$d
  (MKAMD_DIAG=1 MKAMD_LIB=$lib timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/diag$d -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra --no-pipeline > $R/gpurun_out/diag$d.log 2>&1; echo "rc=$?" >> $R/gpurun_out/diag$d.log)
done
cd $R
python - <<'PY'
import csv, glob, collections, json
names={0:"full kernel",1:"- pair loops",3:"- pair loops - flushes",4:"- epilogue arithmetic",11:"- placement - all class work",27:"- both traversals - all class work"}
for d in (0,1,3,4,11,27):
    fs=sorted(glob.glob(f'gpurun_out/diag{d}/*/*counter_collection.csv'))
    if not fs: print(d,"no data"); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        if 'k_voxelize_tiles<8' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    ms=None
    for l in open(f'gpurun_out/diag{d}.log'):
        if l.startswith('{'): ms=json.loads(l)['roofline']['kernel_avg_ms']
    tiles=256*512
    print(f"{names[d]:36s} VALU/tile {max(acc['SQ_INSTS_VALU'])/tiles:8.0f}  SALU/tile {max(acc['SQ_INSTS_SALU'])/tiles:7.0f}  LDS/tile {max(acc['SQ_INSTS_LDS'])/tiles:6.0f}  kernel ms (under the profiler) {ms}")
PY
