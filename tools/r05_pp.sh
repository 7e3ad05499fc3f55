# round 5: the ping-pong forward kernel -- GPU tests, layer A/B against the 4-wave kernel
# (MSMD_FWD_PP=0), phase profile, LC line A/B.  One gpurun call.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_pp2; mkdir -p $O; cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/tests.log
for pp in 1 0; do
  MSMD_FWD_PP=$pp python tools/split_bench.py --lc --check > $O/layers_lc_pp$pp.txt 2>&1
done
MSMD_FWD_NT12=0 python tools/split_bench.py --lc > $O/layers_lc_pp1_nt12off.txt 2>&1
MSMD_LIB=$R/msmdfusion_amd/libmsmd_hip_prof.so python tools/kprof.py --lc > $O/kprof_pp.txt 2>&1
for pp in 1 0 1 0; do
  MSMD_FWD_PP=$pp python bench.py --no-also --no-cpu-baseline > $O/bench_pp${pp}_$RANDOM.json 2>> $O/bench.err
done
tail -6 $O/tests.log; cat $O/layers_lc_pp1.txt $O/layers_lc_pp0.txt | grep -v amdgpu | cut -c1-200
