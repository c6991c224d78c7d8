# tools/gpu_pmc_libs.sh lib... -- VALU / SALU / LDS instructions per tile and kernel time (cfg2, in order) for several builds, same box
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for lib in moleculekit_amd/csrc/libmkamd.so "$@"; do
  tag=$(basename $lib .so)
  rm -rf $R/gpurun_out/pl_$tag
  (MKAMD_LIB=$R/$lib timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pl_$tag -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra --no-pipeline > $R/gpurun_out/pl_$tag.log 2>&1)
  for rep in 1 2; do (MKAMD_LIB=$R/$lib timeout 300 python $R/bench.py --no-cpu-baseline --no-extra --no-pipeline > $R/gpurun_out/pl_${tag}_t$rep.log 2>&1); done
done
cd $R
python - "$@" <<'PY'
import csv, glob, collections, json, sys, os
for lib in ["moleculekit_amd/csrc/libmkamd.so"]+sys.argv[1:]:
    tag=os.path.basename(lib)[:-3]
    fs=sorted(glob.glob(f'gpurun_out/pl_{tag}/*/*counter_collection.csv'))
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        if 'k_voxelize_tiles<8' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    ms=[]
    for rep in (1,2):
        for l in open(f'gpurun_out/pl_{tag}_t{rep}.log'):
            if l.startswith('{'): ms.append(json.loads(l)['roofline']['kernel_avg_ms'])
    t=256*512
    print(f"{tag:24s} VALU/tile {max(acc['SQ_INSTS_VALU'])/t:7.0f} SALU/tile {max(acc['SQ_INSTS_SALU'])/t:6.0f} LDS/tile {max(acc['SQ_INSTS_LDS'])/t:5.0f} VALU-busy quad-cycles/tile {max(acc['SQ_ACTIVE_INST_VALU'])/t:7.0f}  kernel ms {ms}")
PY
