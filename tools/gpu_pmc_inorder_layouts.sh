# tools/gpu_pmc_inorder_layouts.sh -- instructions per tile of the in-order tile kernel on the two record layouts: the chain's
# column runs (MKAMD_DIRECT=0) and the one-pass binning's cell runs (automatic), same box, one PMC pass each
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
PROF="--no-cpu-baseline --no-extra --no-single --min-seconds 0 --steps 8 --warmup 2 --no-pipeline"
for d in 0 -1; do
  tag=layout_direct$d
  rm -rf $R/gpurun_out/pmc_$tag
  (cd /tmp && MKAMD_DIRECT=$d timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_$tag -- python $R/bench.py $PROF > $R/gpurun_out/pmc_$tag.log 2>&1)
done
python - <<'PY'
import csv, glob, collections, os
for d in ("0", "-1"):
    fs = sorted(glob.glob(f'gpurun_out/pmc_layout_direct{d}/*/*counter_collection.csv'), key=os.path.getmtime)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[-1])):
        k = r['Kernel_Name'].split('(')[0]
        if 'mkamd::' in k: acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    print("MKAMD_DIRECT=" + d + ("  (chain: column runs)" if d == "0" else "  (automatic: k_bin_direct, cell runs)"))
    for k, c in acc.items():
        w = c['SQ_WAVES'][-1]; i = -1          # the LAST launch: the first call of a context has no class table, its direct pass fails over to the chain
        if w < 1000: continue
        g = lambda n: c[n][i]
        print(f"   {k[-44:]:44s} waves {w:8.0f}  per wave: VALU {g('SQ_INSTS_VALU')/w:7.0f} SALU {g('SQ_INSTS_SALU')/w:7.0f} LDS {g('SQ_INSTS_LDS')/w:6.0f} VMEM {g('SQ_INSTS_VMEM')/w:6.1f}")
PY
