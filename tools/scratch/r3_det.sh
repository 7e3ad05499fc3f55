cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_detector.py tests/test_gpu_fusion.py -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl\|amdgpu.ids" | tail -15
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-also --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['host'])"
done
MSMD_BLOCKING_SYNC=1 timeout 300 python bench.py --no-cpu-baseline --no-also --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('blocking', d['value'], d['ms_per_step'], d['host'])"
