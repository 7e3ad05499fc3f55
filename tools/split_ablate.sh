for d in 1 3 5; do echo "dbg=$d"; MSMD_DBG=$d python tools/split_bench.py --planes 3 2>&1 | grep -E "128->128|64-> 64|32-> 32" | sed 's/.*| split3/split3/'; done
