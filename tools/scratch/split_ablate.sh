python tools/split_bench.py --planes 3 --check 2>&1 | grep -v amdgpu.ids | tail -5 | sed 's/n=.*| split3/| split3/'
for d in 2 4 6; do echo "dbg=$d"; MSMD_DBG=$d python tools/split_bench.py --planes 3 2>&1 | grep -E "128->128|64-> 64|32-> 32" | sed 's/.*| split3/split3/'; done
python tools/split_bench.py --planes 2 --check 2>&1 | grep -v amdgpu.ids | tail -5| sed 's/n=.*| split2/split2/'
