# ablation of the ping-pong kernel (timing only: MSMD_DBG bits give wrong results by design):
# 2 = no gathers (out-of-range offsets), 4 = no MFMAs, 6 = neither, 8 = gathers folded onto 4096 rows
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_abl; mkdir -p $O; cd $R
for pp in 97 999; do for dbg in 0 2 4 6 8; do
  echo "== MSMD_FWD_PP_MIN=$pp MSMD_DBG=$dbg" >> $O/abl.txt
  MSMD_FWD_PP_MIN=$pp MSMD_DBG=$dbg python tools/split_bench.py --lc 2>/dev/null | grep fwd | sed 's/| fp32[^|]*|//' | cut -c1-140 >> $O/abl.txt
done; done
cat $O/abl.txt
