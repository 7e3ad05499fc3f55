cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03h; mkdir -p $OUT
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
rm -rf $OUT/p
timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p -o s -- python $R/tools/wgrad_ablate.py > $OUT/log.txt 2>&1
c=$(find $OUT/p -name "*counter_collection.csv" | head -1)
python - "$c" <<'PY'
import csv,sys,collections
d=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if 'spconv_wgrad_block_kernel' in r['Kernel_Name']:
        d[r['Counter_Name']]['v'].append(float(r['Counter_Value']))
for k,v in d.items():
    x=sorted(v['v']); print(k, 'n',len(x),'median %.4g max %.4g'%(x[len(x)//2], x[-1]))
PY
done
rm -rf $OUT/p
