cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "split_conv or tile_prefix or tiling" 2>&1 | tail -8
for mode in split block4 block8; do
  case $mode in
    split) export MSMD_FWD=split;;
    block4) export MSMD_FWD=block MSMD_FWD_PW=4;;
    block8) export MSMD_FWD=block MSMD_FWD_PW=8;;
  esac
  echo "== $mode"
  timeout 200 python tools/split_bench.py --check 2>&1 | grep "fwd" | cut -c1-200
  timeout 200 python tools/split_bench.py --lc 2>&1 | grep "fwd" | cut -c1-160
done
