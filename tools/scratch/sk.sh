cd /root/repo
for z in 1 0; do for c in 3 12; do echo "== ZIGZAG=$z C1=$c"; MSMD_TILE_ZIGZAG=$z MSMD_SK_C1=$c python tools/split_bench.py 2>&1 | grep "fwd" | sed 's/| fp32.*| split3/| split3/' | cut -c1-150; MSMD_TILE_ZIGZAG=$z MSMD_SK_C1=$c python tools/split_bench.py --lc 2>&1 | grep "fwd" | sed 's/| fp32.*| split3/| split3/' | cut -c1-150; done; done
