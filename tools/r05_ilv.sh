# round 5: interleaved load segment of the ping-pong kernel (MSMD_FWD_ILV) -- tests, layer A/B, LC A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_ilv2; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_production.py -q 2>&1 | tail -6 > $O/tests.log
for ilv in 1 0; do
  MSMD_FWD_ILV=$ilv MSMD_FWD_PP_MIN=97 python tools/split_bench.py --lc --check > $O/layers_lc_ilv$ilv.txt 2>&1
done
MSMD_LIB=$R/msmdfusion_amd/libmsmd_hip_prof.so MSMD_FWD_PP_MIN=97 python tools/kprof.py --lc > $O/kprof_ilv1.txt 2>&1
for v in "1 161" "0 161" "1 97" "1 161" "0 161" "1 97"; do set -- $v
  MSMD_FWD_ILV=$1 MSMD_FWD_PP_MIN=$2 python bench.py --no-also --no-cpu-baseline > $O/bench_ilv$1_min$2_$RANDOM.json 2>> $O/bench.err
done
tail -3 $O/tests.log; grep -h "fwd" $O/layers_lc_ilv1.txt $O/layers_lc_ilv0.txt | cut -c1-150; grep -v amdgpu $O/kprof_ilv1.txt
for f in $O/bench_ilv*.json; do python -c "
import json,sys;d=json.load(open('$f'));r=d['roofline'];print('$f'.split('/')[-1], d['value'], d['ms_per_step'], {k.split('<')[1][:10]:(v['ms'],v['tflops']) for k,v in r['all_conv_kernels'].items() if 'split_kernel' in k})"; done
