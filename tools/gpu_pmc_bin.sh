# tools/gpu_pmc_bin.sh lib... -- instructions per wave and busy cycles of k_bin_count (cfg2, in order) for the in-tree build and diagnostic ones
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for lib in moleculekit_amd/csrc/libmkamd.so "$@"; do
  tag=$(basename $lib .so)
  rm -rf $R/gpurun_out/pb_$tag
  (MKAMD_DIAG=1 MKAMD_LIB=$R/$lib timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/pb_$tag -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra --no-pipeline ${PB_ARGS:-} > $R/gpurun_out/pb_$tag.log 2>&1)
done
cd $R
python - "$@" <<'PY'
import csv, glob, collections, sys, os
for lib in ["moleculekit_amd/csrc/libmkamd.so"]+sys.argv[1:]:
    tag=os.path.basename(lib)[:-3]
    fs=sorted(glob.glob(f'gpurun_out/pb_{tag}/*/*counter_collection.csv'), key=os.path.getmtime)
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[-1])):
        k=r['Kernel_Name'].split('(')[0]
        if 'k_bin' in k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,c in acc.items():
        w=max(c['SQ_WAVES']); i=c['SQ_WAVES'].index(w)
        g=lambda n: c[n][i]
        print(f"{tag:16s} {k[-28:]:28s} waves {w:8.0f} VALU/wave {g('SQ_INSTS_VALU')/w:6.0f} SALU/wave {g('SQ_INSTS_SALU')/w:6.0f} SMEM/wave {g('SQ_INSTS_SMEM')/w:5.1f} busy Mcycles {g('SQ_BUSY_CYCLES')/32e6:6.3f} waitcnt share {g('SQ_WAIT_INST_ANY')/g('SQ_WAVE_CYCLES'):5.2f}")
PY
