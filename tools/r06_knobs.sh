R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06j; mkdir -p $OUT; cd $R
run() { tag=$1; shift; for i in 1 2; do env "$@" timeout 300 python bench.py --no-also --no-cpu-baseline --no-profile > $OUT/$tag.$i.json 2>/dev/null; python -c "
import json; d=json.load(open('$OUT/$tag.$i.json')); print('$tag: %.1f samples/s %.3f ms' % (d['value'], d['ms_per_step']))"; done; }
run base X=1
run idxprio0 MSMD_INDEX_PRIORITY=0
run nnprio0 MSMD_NN_PRIORITY=0
run nn1 MSMD_NN_STREAMS=1
run nn4 MSMD_NN_STREAMS=4
run depth3 MSMD_PREFETCH_DEPTH=3
run hwq4 GPU_MAX_HW_QUEUES=4
