cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 2>&1 | grep -v "RCCL\|HIP version\|ROCm\|Hostname\|Librccl\|amdgpu.ids" | tail -25
