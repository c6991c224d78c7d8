# tools/gpu_kstats_libs.sh lib... -- per-kernel average durations (rocprofv3 kernel trace, cfg2 in order) for the in-tree build and the given ones, same box
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for lib in moleculekit_amd/csrc/libmkamd.so "$@"; do
  tag=$(basename $lib .so)
  rm -rf $R/gpurun_out/ks_$tag
  (MKAMD_LIB=$R/$lib timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ks_$tag -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra --no-pipeline ${KS_ARGS:-} > $R/gpurun_out/ks_$tag.log 2>&1)
done
cd $R
python - "$@" <<'PY'
import csv, glob, collections, sys, os
for lib in ["moleculekit_amd/csrc/libmkamd.so"]+sys.argv[1:]:
    tag=os.path.basename(lib)[:-3]
    fs=sorted(glob.glob(f'gpurun_out/ks_{tag}/*/*kernel_trace.csv'), key=os.path.getmtime)
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        n=r['Kernel_Name']
        if 'mkamd::' not in n: continue
        acc[n.split('(')[0][:60]].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
    print(tag)
    for k,v in sorted(acc.items(), key=lambda kv:-sum(kv[1])):
        v=v[len(v)//4:]           # drop the warm-up quarter
        print(f"   {k:62s} n {len(v):4d} avg {sum(v)/len(v)/1e3:9.1f} us  min {min(v)/1e3:9.1f}")
PY
