mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
SQ2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_IFETCH"
dpmc() { name=$1; mode=$2; shift; shift; rm -rf $R/gpurun_out/pmc_dist_${mode}_$name; (cd /tmp && MKAMD_DIST_ONLY=$mode timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_dist_${mode}_$name -- python $R/bench.py --workload dist --no-cpu-baseline --steps 8 --warmup 2 > $R/gpurun_out/pmc_dist_${mode}_$name.log 2>&1); }
for mode in periodic nonperiodic; do
  dpmc sq1 $mode $SQ1
  dpmc sq2 $mode $SQ2
done
python - <<'PY'
import csv, glob, collections
for mode in ("periodic","nonperiodic"):
    acc=collections.defaultdict(list)
    for p in ("sq1","sq2"):
        fs=sorted(glob.glob(f'gpurun_out/pmc_dist_{mode}_{p}/*/*counter_collection.csv'))
        if not fs: continue
        for r in csv.DictReader(open(fs[-1])):
            if 'k_dist_pairs' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    v={k: sum(x)/len(x) for k,x in acc.items()}
    if not v: print(mode,'no data'); continue
    w=v['SQ_WAVES']
    print(mode, 'VALU/wave %.0f SALU/wave %.0f LDS/wave %.0f VMEM/wave %.0f'%(v['SQ_INSTS_VALU']/w, v['SQ_INSTS_SALU']/w, v['SQ_INSTS_LDS']/w, v.get('SQ_INSTS_VMEM',0)/w),
          'valu_busy %.3f'%(v['SQ_ACTIVE_INST_VALU']*4/1024/v['GRBM_GUI_ACTIVE']), 'cycles %.0f'%v['GRBM_GUI_ACTIVE'],
          'wave_cycles %.3g wait_any %.3g wait_inst %.3g active_any %.3g lds_conf %.3g'%(v['SQ_WAVE_CYCLES'], v.get('SQ_WAIT_ANY',0), v.get('SQ_WAIT_INST_ANY',0), v.get('SQ_ACTIVE_INST_ANY',0), v.get('SQ_LDS_BANK_CONFLICT',0)))
PY
