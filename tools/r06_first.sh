# round 6, first call: the B=2 alone / in-step table, the default LC line on this box, the
# reference_quirks line.   gpurun --timeout 1500 -- 'bash tools/r06_first.sh'
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06a; mkdir -p $OUT; cd $R
timeout 600 python tools/split_bench.py --lc-b2 > $OUT/lc_b2.txt 2> $OUT/lc_b2.err; tail -12 $OUT/lc_b2.txt; tail -3 $OUT/lc_b2.err
timeout 300 python bench.py --no-also --no-cpu-baseline > $OUT/bench_lc.json 2> $OUT/bench_lc.err
python - <<PY
import json
d = json.load(open("$OUT/bench_lc.json")); r = d["roofline"]
print("LC", d["value"], d["ms_per_step"], r["kernel"], r["frac"], r["step_frac"], r["step_conv_ms"])
PY
timeout 300 python bench.py --workload lc_quirks --no-also --no-cpu-baseline > $OUT/bench_quirks.json 2> $OUT/bench_quirks.err
tail -c 1500 $OUT/bench_quirks.json; tail -3 $OUT/bench_quirks.err
