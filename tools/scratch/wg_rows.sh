for D in 2 18 6; do echo "== DBG=$D"; MSMD_DBG=$D python tools/wgrad_planes_bench.py 2>&1 | grep "stage 3 128->128\|stage 2  64"; done
