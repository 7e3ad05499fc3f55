cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03i; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_production.py -q -x -k "wgrad or production or conv" 2>&1 | tail -3
timeout 120 python tools/wgrad_ablate.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee $OUT/ab.txt
MSMD_BENCH_LAYERS=1 timeout 300 python bench.py --no-cpu-baseline --no-also > $OUT/lc.json 2> $OUT/lc.err
grep "^\[layer\]" $OUT/lc.err | grep wgrad | sort -k4,4 -k7,7n | awk 'NR%2==1' | head -20
python -c "
import json
d=json.loads(open('$OUT/lc.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['frac'], {k:v for k,v in d['roofline']['all_conv_kernels'].items() if 'wgrad' in k})"
