R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06f; mkdir -p $OUT; cd $R
for p in 3 1 3 1; do
  MSMD_CONV_PLANES=$p timeout 300 python bench.py --no-also --no-cpu-baseline --no-profile > $OUT/p$p.json 2> $OUT/p$p.err
  python -c "
import json; d=json.load(open('$OUT/p$p.json')); print('planes $p: %.1f samples/s %.3f ms' % (d['value'], d['ms_per_step']))"
done
