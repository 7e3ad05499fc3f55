# round 6: the tests touched by the small items (review items 4, 9 + the advisor's findings)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06b; mkdir -p $OUT; cd $R
timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_reference_quirks.py tests/test_gpu_image_glue.py \
  tests/test_gpu_rccl_two_ranks.py "tests/test_gpu_modules.py::test_packed_parameter_images_follow_the_optimizer" \
  "tests/test_gpu_kernels.py::test_rows_where_eq" tests/test_boundary.py 2>&1 | tail -15 > $OUT/tests.log
cat $OUT/tests.log
MSMD_CHECK_COUNTS=1 timeout 600 python -m pytest -m gpu -x -q tests/test_gpu_fusion.py 2>&1 | tail -3
timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_bench_workloads.py 2>&1 | tail -5
cat gpurun_out/r06_workload_parity.txt 2>/dev/null | tail -30
