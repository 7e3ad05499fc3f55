# one gpurun call: the GPU suite + the default bench line (tag = $1)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${1:-r05_check}; mkdir -p $O; cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/tests.log
python bench.py ${2:-} > $O/bench.json 2> $O/bench.err
tail -5 $O/tests.log; python -c "
import json;d=json.load(open('$O/bench.json'));r=d['roofline']
print(d['value'],d['ms_per_step'],r['kernel'],r['frac'],r['bound'],r['frac_mfma'],r.get('step_frac'))
print({k:(v.get('value'),v.get('ms_per_step')) for k,v in d.get('also',{}).items()})
print({k.split('<')[1][:16] if '<' in k else k:(v['ms'],v['tflops'],v['launches']) for k,v in r['all_conv_kernels'].items()})"
