cd /root/repo
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "planes" 2>&1 | tail -3
for D in 0 2 4; do echo "== DBG=$D"; MSMD_DBG=$D python tools/wgrad_planes_bench.py 2>&1 | grep "stage" | sed 's/+ split pass.*//' ; done
