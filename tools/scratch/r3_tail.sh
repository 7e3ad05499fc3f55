cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
MSMD_BENCH_LAYERS=1 timeout 300 python bench.py --workload lc_tail --no-also --no-cpu-baseline > gpurun_out/tail.json 2> gpurun_out/tail.err
grep "^\[layer\]" gpurun_out/tail.err | head -400 > gpurun_out/tail_layers.txt
wc -l gpurun_out/tail_layers.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -o s -- python $GRAFT_REPO_ROOT/bench.py --workload lc_tail --no-also --no-cpu-baseline --no-profile > /tmp/pt.json 2> /tmp/pt.err
python $GRAFT_REPO_ROOT/tools/stream_prof.py $(find /tmp/pt -name "*kernel_trace.csv" | head -1) /tmp/pt.err 40 > $GRAFT_REPO_ROOT/gpurun_out/tail_stream.txt
head -60 $GRAFT_REPO_ROOT/gpurun_out/tail_stream.txt
