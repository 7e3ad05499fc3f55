# rocprofv3 kernel stats of the default bench (25 steps incl. warmup); summary to stdout
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-prof_cur}
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python $R/bench.py ${2:-} --no-cpu-baseline --no-profile > $OUT/bench.log 2>&1
rm -f $OUT/*kernel_trace.csv      # (per-dispatch rows: tens of MB; the stats are what is kept)
tail -1 $OUT/bench.log | cut -c1-200
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python $R/tools/prof_summary.py $f $([ "${2:-}" = "--workload lc" ] && echo 42 || echo 36) 45
