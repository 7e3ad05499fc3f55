# rocprofv3 --pmc passes over one bench workload (one counter set per pass, kernel
# trace only -- gpurun refuses --pmc together with the hip/hsa trace domains), merged
# into $OUT/pmc_summary.json by tools/pmc_to_json.py.
#   bash tools/pmc_collect.sh <lc|transfusion_l> <outdir>      (on the GPU box)
# MSMD_PREFETCH=0: counters are per launch and PMC mode serialises kernels anyway; the
# inline schedule needs fewer untimed settle steps.
WL=${1:-lc}
R=$GRAFT_REPO_ROOT
OUT=${2:-$R/gpurun_out/pmc_$WL}
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o s -- \
    env MSMD_PREFETCH=0 python $R/bench.py --workload $WL --no-also --steps 3 --warmup 2 \
    --no-cpu-baseline --no-profile > $OUT/g$i.log 2>&1
  tail -1 $OUT/g$i.log | cut -c1-120
done
python $R/tools/pmc_to_json.py $OUT $OUT/pmc_summary.json $WL
# gpurun merges at most 64 MiB back: the per-dispatch CSVs have served their purpose
rm -rf $OUT/g[0-9]
