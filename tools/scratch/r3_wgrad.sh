# round-3 wgrad A/B: parity tests, then timings of the block kernel against the slab kernel
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03a; mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -x -k "wgrad" 2>&1 | tail -15 > $OUT/t_wgrad.log; tail -5 $OUT/t_wgrad.log
timeout 120 python tools/wgrad_ablate.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee $OUT/ab_block.txt
MSMD_WGRAD=var timeout 120 python tools/wgrad_ablate.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee $OUT/ab_var.txt
MSMD_WGRAD_DBG=1 timeout 120 python tools/wgrad_ablate.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee $OUT/ab_block_dbg1.txt
MSMD_WGRAD_DBG=2 timeout 120 python tools/wgrad_ablate.py 2>&1 | grep -v amdgpu.ids | tail -1 | tee $OUT/ab_block_dbg2.txt
timeout 300 python -m pytest tests/test_gpu_production.py -q -x 2>&1 | tail -5 | tee $OUT/t_prod.log
