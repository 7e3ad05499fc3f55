cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03e; mkdir -p $OUT
for d in 62 126 190; do
MSMD_WGRAD_DBG=$d timeout 120 python tools/wgrad_ablate.py 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-150 | tee -a $OUT/ab_block_dbg2.txt
done
