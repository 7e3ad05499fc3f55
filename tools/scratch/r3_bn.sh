cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "RCCL\|amdgpu\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
for i in 1 2 3; do timeout 300 python bench.py --no-also --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['frac'], r['avg_launch_us'], r['launches'], r['sampling'][-40:])"; done
timeout 300 python bench.py --workload transfusion_l --no-also --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['frac'], r['avg_launch_us'], r['launches'], r['sampling'][-40:])"
