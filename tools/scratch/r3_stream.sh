cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03j; mkdir -p $OUT
rm -rf $OUT/p
timeout 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/p -o s -- python $R/bench.py --workload lc --no-also --no-cpu-baseline --no-profile > $OUT/bench.json 2> $OUT/bench.err
t=$(find $OUT/p -name "*kernel_trace.csv" | head -1)
head -1 $t
python $R/tools/stream_prof.py $t $OUT/bench.err 45 > $OUT/stream_lc.txt
head -100 $OUT/stream_lc.txt
rm -rf $OUT/p
tail -c 400 $OUT/bench.json
