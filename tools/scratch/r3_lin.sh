cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_fusion.py -x -q 2>&1 | grep -v "RCCL\|amdgpu\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -5
bash tools/prof_bench.sh r03d lc > /dev/null 2>&1
grep -n "rows_linear\|Cijk" gpurun_out/r03d/prof_lc/stream_summary.txt | head
head -4 gpurun_out/r03d/prof_lc/stream_summary.txt
