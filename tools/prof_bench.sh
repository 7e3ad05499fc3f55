# rocprofv3 kernel trace of one bench workload -> per-kernel stats CSV + per-queue, per-step
# summary.  The number of steps the trace covers is READ FROM THE RUN (bench.py prints
# `[bench] workload=.. steps_total=N` on stderr: settle + warm-up + timed steps, whatever the
# time-based settle phase took), never assumed.
#   bash tools/prof_bench.sh <outdir-tag> [lc|transfusion_l]        (on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=${2:-lc}
OUT=$R/gpurun_out/${1:-prof_cur}/prof_$WL
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- \
  python $R/bench.py --workload $WL --no-also --no-cpu-baseline --no-profile > $OUT/bench.json 2> $OUT/bench.err
tail -c 300 $OUT/bench.json; echo
t=$(find $OUT -name "*kernel_trace.csv" | head -1)
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python $R/tools/stream_prof.py $t $OUT/bench.err 45 > $OUT/stream_summary.txt
steps=$(grep -o "steps_total=[0-9]*" $OUT/bench.err | tail -1 | cut -d= -f2)
python $R/tools/prof_summary.py $f $((steps + 1)) 45 > $OUT/summary.txt
cp $f $OUT/kernel_stats.csv
rm -rf $OUT/*/ $OUT/*kernel_trace.csv   # per-dispatch rows: tens of MB; the stats are what is kept
head -30 $OUT/stream_summary.txt
