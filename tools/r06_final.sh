# round 6 closing evidence in one gpurun call: GPU suite, smoke, the default bench line (all legs),
# rocprofv3 kernel stats + per-queue summary of both workloads, integer-kernel rooflines, FPS,
# PMC passes under the pipelined schedule
R=$GRAFT_REPO_ROOT; T=${1:-r06_final}; O=$R/gpurun_out/$T; mkdir -p $O; cd $R
python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/tests.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1
python bench.py > $O/bench_default.json 2> $O/bench_default.err
for wl in lc transfusion_l; do bash tools/prof_bench.sh $T $wl > $O/prof_$wl.log 2>&1; done
timeout 300 python tools/rulebook_bench.py 2>/dev/null > $O/rulebook_voxelize_roofline.jsonl
timeout 120 python tools/fps_bench.py 2>/dev/null | grep -v amdgpu.ids > $O/fps.txt
bash tools/pmc_collect.sh lc $O/pmc_lc > $O/pmc.log 2>&1
tail -2 $O/tests.log; tail -1 $O/smoke.log; python -c "
import json;d=json.load(open('$O/bench_default.json'));r=d['roofline']
print(d['value'],d['ms_per_step'],r['kernel'],r['bound'],r['frac'],'step_frac',r.get('step_frac'),'traffic/algo',r.get('traffic_over_algorithmic'))
print({k:(v.get('value'),v.get('ms_per_step')) for k,v in d.get('also',{}).items()})
print(d['cpu_baseline']['value'])"
