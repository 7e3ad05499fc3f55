R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06g; mkdir -p $OUT; cd $R
timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_kernels.py -k "voxel" 2>&1 | tail -5
timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_fusion.py tests/test_gpu_production.py tests/test_gpu_integration.py tests/test_gpu_bench_workloads.py 2>&1 | grep -E "passed|failed" | tail -2
for m in 0 1; do echo "MSMD_VOXELIZE_MANY=$m"; MSMD_VOXELIZE_MANY=$m timeout 300 python tools/rulebook_bench.py 2>/dev/null | grep hard_voxelize | cut -c1-230; done
for m in 0 1 0 1; do
  MSMD_VOXELIZE_MANY=$m timeout 300 python bench.py --no-also --no-cpu-baseline --no-profile > $OUT/v$m.json 2>/dev/null
  python -c "
import json; d=json.load(open('$OUT/v$m.json')); print('many=$m: %.1f samples/s %.3f ms' % (d['value'], d['ms_per_step']))"
done
