cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "split_conv or tile_prefix or tiling" 2>&1 | tail -2
cp msmdfusion_amd/libmsmd_hip_dbg.so msmdfusion_amd/libmsmd_hip.so
MSMD_FWD=split timeout 100 python tools/fwd_ablate.py 2>&1 | grep "FWD="
for d in ${DBGS:-0 16 48 15 47 63 1 2 4 8 32}; do
  MSMD_FWD_DBG=$d timeout 100 python tools/fwd_ablate.py 2>&1 | grep "FWD="
done
