# rocprofv3 --pmc passes over one bench workload (one counter set per pass, kernel
# trace only -- gpurun refuses --pmc together with the hip/hsa trace domains), merged
# into $OUT/pmc_summary.json by tools/pmc_to_json.py.
#   bash tools/pmc_collect.sh <lc|transfusion_l> <outdir>      (on the GPU box)
# The bench's DEFAULT schedule (index pass a step ahead on its own stream: MSMD_PREFETCH=1),
# so that the per-launch traffic belongs to the launches bench.py times.  PMC mode still
# serialises kernels: these runs are for counters, never for time.  $3 = counter groups to
# collect (default: the HBM bytes + MFMA busy ones bench.py's roofline reads; "all" adds the
# wave-cycle and L2 hit groups).
WL=${1:-lc}
R=$GRAFT_REPO_ROOT
OUT=${2:-$R/gpurun_out/pmc_$WL}
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
i=0
PMC_GROUPS=("FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" \
        "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum")
[ "${3:-}" = "all" ] && PMC_GROUPS+=("SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY")
for grp in "${PMC_GROUPS[@]}"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o s -- \
    env MSMD_BENCH_SETTLE_S=0.3 python $R/bench.py --workload $WL --no-also --steps 3 --warmup 2 \
    --no-cpu-baseline --no-profile > $OUT/g$i.log 2>&1
  tail -1 $OUT/g$i.log | cut -c1-120
done
python $R/tools/pmc_to_json.py $OUT $OUT/pmc_summary.json $WL
# gpurun merges at most 64 MiB back: the per-dispatch CSVs have served their purpose
rm -rf $OUT/g[0-9]
