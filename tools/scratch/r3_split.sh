cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03l; mkdir -p $OUT
OMP_NUM_THREADS=8 timeout 300 python tools/lc_split_time.py > $OUT/split_full.txt 2>&1
grep -n "tottime" $OUT/split_full.txt | head
