cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03f; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "wgrad" 2>&1 | tail -3
timeout 120 python tools/wgrad_ablate.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee $OUT/ab.txt
timeout 120 python tools/wgrad_ablate.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $OUT/ab.txt
