cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03k; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_fusion.py -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl\|amdgpu.ids" | tail -4
for f in 1 0; do
MSMD_FUSED_ASSEMBLY=$f timeout 300 python bench.py --no-cpu-baseline --no-also --no-profile > $OUT/lc$f.json 2> $OUT/lc$f.err
python -c "
import json
d=json.loads(open('$OUT/lc$f.json').read().strip().splitlines()[-1])
print('fused=$f', d['value'], d['ms_per_step'])"
done
