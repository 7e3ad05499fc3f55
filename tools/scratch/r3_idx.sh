cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl\|amdgpu.ids" | tail -6
for i in 1 2 3; do
timeout 300 python bench.py --no-cpu-baseline --no-also --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done
