cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
echo "== fixed library"
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "scans_do_not_depend" 2>&1 | tail -3
cp msmdfusion_amd/libmsmd_hip.so /tmp/good.so
cp msmdfusion_amd/libmsmd_hip_bug.so msmdfusion_amd/libmsmd_hip.so
echo "== library without the barrier"
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "scans_do_not_depend" 2>&1 | tail -6 | cut -c1-200
cp /tmp/good.so msmdfusion_amd/libmsmd_hip.so
