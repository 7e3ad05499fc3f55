R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06h; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
timeout 300 python tools/rulebook_bench.py 2>/dev/null | cut -c1-200
for i in 1 2; do
  timeout 300 python bench.py --no-also --no-cpu-baseline --no-profile > $OUT/b$i.json 2>/dev/null
  python -c "
import json; d=json.load(open('$OUT/b$i.json')); print('LC: %.1f samples/s %.3f ms' % (d['value'], d['ms_per_step']))"
done
