# weight DMA pieces moved into the multiply segment (builds with -DMSMD_PP_WM=1 / 2) against the shipped order
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_wm; mkdir -p $O; cd $R
for lib in libmsmd_hip libmsmd_hip_wm1 libmsmd_hip_wm2; do
  echo "== $lib" >> $O/wm.txt
  MSMD_LIB=$R/msmdfusion_amd/$lib.so MSMD_FWD_PP_MIN=65 python tools/split_bench.py --lc --check 2>/dev/null | grep fwd | sed 's/| fp32[^|]*|//' | cut -c1-170 >> $O/wm.txt
done
cat $O/wm.txt
