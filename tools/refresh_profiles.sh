# copy the judged summaries from gpurun_out/ (written by tools/prof_bench.sh, tools/pmc_collect.sh
# and the plain bench runs) into profiles/
set -e
cd "$(dirname "$0")/.."
cp gpurun_out/prof_final/s_kernel_stats.csv profiles/r01_transfusion_l_kernel_stats.csv
cp gpurun_out/prof_final_lc/s_kernel_stats.csv profiles/r01_lc_kernel_stats.csv
cp gpurun_out/pmc/pmc_summary.json profiles/r01_pmc_summary.json
[ -s gpurun_out/rulebook_bench.jsonl ] && grep "^{" gpurun_out/rulebook_bench.jsonl > profiles/r01_rulebook_voxelize_roofline.jsonl
grep "^{" gpurun_out/bench_default.json | tail -1 > profiles/r01_bench_default.json
grep "^{" gpurun_out/bench_lc.json | tail -1 > profiles/r01_bench_lc.json
grep "^{" gpurun_out/prof_final/bench.log | tail -1 > profiles/r01_transfusion_l_bench_under_rocprof.json
python - <<'PY'
import json
for f in ('profiles/r01_bench_default.json', 'profiles/r01_bench_lc.json'):
    d = json.loads(open(f).read()); r = d.get('roofline') or {}
    print(f, d['value'], d['ms_per_step'], r.get('kernel'), r.get('achieved'), r.get('frac'),
          r.get('traffic'), r.get('mfma_pipe_busy_frac_pmc'), r.get('avg_launch_us'))
PY
