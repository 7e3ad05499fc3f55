cd /root/repo
MSMD_FWD_WAVES=8 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "split_conv or tile_prefix" 2>&1 | tail -3
for U in 2 1; do echo "== WAVES=8 UB8=$U"; MSMD_FWD_UB8=$U MSMD_FWD_WAVES=8 python tools/split_bench.py 2>&1 | grep fwd | sed 's/| fp32.*| split3/| split3/' | cut -c1-150; MSMD_FWD_UB8=$U MSMD_FWD_WAVES=8 python tools/split_bench.py --lc 2>&1 | grep fwd| sed 's/| fp32.*| split3/| split3/' | cut -c1-150; done
