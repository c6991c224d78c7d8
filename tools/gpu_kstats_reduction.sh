# tools/gpu_kstats_reduction.sh lib... -- k_dist_reduction's duration (rocprofv3 kernel trace of tools/bench_reduction.py) for the in-tree build and the given ones
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for lib in moleculekit_amd/csrc/libmkamd.so "$@"; do
  tag=$(basename $lib .so)
  rm -rf $R/gpurun_out/kr_$tag
  (MKAMD_LIB=$R/$lib timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kr_$tag -- python $R/tools/bench_reduction.py > $R/gpurun_out/kr_$tag.log 2>&1)
  tail -1 $R/gpurun_out/kr_$tag.log
done
cd $R
python - "$@" <<'PY'
import csv, glob, collections, sys, os
for lib in ["moleculekit_amd/csrc/libmkamd.so"]+sys.argv[1:]:
    tag=os.path.basename(lib)[:-3]
    fs=sorted(glob.glob(f'gpurun_out/kr_{tag}/*/*kernel_trace.csv'), key=os.path.getmtime)
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        n=r['Kernel_Name']
        if 'mkamd::' not in n: continue
        acc[n.split('(')[0][:60]].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
    print(tag)
    for k,v in sorted(acc.items(), key=lambda kv:-sum(kv[1])):
        print(f"   {k:62s} n {len(v):4d} avg {sum(v)/len(v)/1e3:9.1f} us  min {min(v)/1e3:9.1f}")
PY
