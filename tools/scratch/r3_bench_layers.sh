cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03d; mkdir -p $OUT
MSMD_BENCH_LAYERS=1 timeout 300 python bench.py --no-cpu-baseline --no-also > $OUT/lc_block2.json 2> $OUT/lc_block2.err
grep "^\[layer\]" $OUT/lc_block2.err | sort | uniq -c | sort -k3 | head -80
tail -c 300 $OUT/lc_block2.json
