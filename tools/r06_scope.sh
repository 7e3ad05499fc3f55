R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06i; mkdir -p $OUT; cd $R
for sc in call stage all call stage all; do
  MSMD_PLAN_SCOPE=$sc timeout 300 python bench.py --no-also --no-cpu-baseline --no-profile > $OUT/$sc.json 2>/dev/null
  python -c "
import json; d=json.load(open('$OUT/$sc.json')); print('scope $sc: %.1f samples/s %.3f ms' % (d['value'], d['ms_per_step']))"
done
