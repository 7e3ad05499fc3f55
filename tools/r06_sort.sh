R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06k; mkdir -p $OUT; cd $R
timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_kernels.py tests/test_gpu_modules.py tests/test_gpu_fusion.py -k "plan or tiling or fusion or prefetch" 2>&1 | grep -E "passed|failed" | tail -2
for m in 0 1 0 1; do
  MSMD_SORT_ONESWEEP=$m timeout 300 python bench.py --no-also --no-cpu-baseline --no-profile > $OUT/s$m.json 2>/dev/null
  python -c "
import json; d=json.load(open('$OUT/s$m.json')); print('onesweep=$m: %.1f samples/s %.3f ms' % (d['value'], d['ms_per_step']))"
done
bash tools/prof_bench.sh r06k lc > $OUT/prof.log 2>&1
grep "^queue" $OUT/prof_lc/stream_summary.txt; grep -i "rocprim\|onesweep\|radix" $OUT/prof_lc/stream_summary.txt | cut -c1-150 | head -12
