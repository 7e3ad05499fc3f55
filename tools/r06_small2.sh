R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06b; mkdir -p $OUT; cd $R
timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_reference_quirks.py tests/test_gpu_kernels.py -k "quirk or repeated or rulebook or conv_chain or strided or float_key or detector_switch or gma_stage" 2>&1 | tail -6
timeout 300 python tools/rulebook_bench.py 2>/dev/null | grep -i "conv3d\|strided" | cut -c1-260
