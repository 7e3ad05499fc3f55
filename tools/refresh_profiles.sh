# copy the judged summaries of one gpurun round (tools/gpu_round.sh <tag> ...) from gpurun_out/<tag>/
# into profiles/ under this round's prefix:   bash tools/refresh_profiles.sh r03a r03
set -e
cd "$(dirname "$0")/.."
TAG=${1:?gpurun_out tag}; R=${2:?profiles prefix, e.g. r03}
G=gpurun_out/$TAG
for wl in lc transfusion_l; do
  P=$G/prof_$wl
  [ -d $P ] || continue
  cp $P/kernel_stats.csv profiles/${R}_${wl}_kernel_stats.csv
  # the commit the profiled tree was at (bench.py's roofline.rocprof_head; the GPU box has no
  # .git): the tree as it is NOW -- commit before profiling, copy before editing
  echo "$(git rev-parse --short=12 HEAD)$(git diff --quiet HEAD -- msmdfusion_amd bench.py || echo -dirty)" \
    > profiles/${R}_${wl}_kernel_stats.head
  cp $P/summary.txt profiles/${R}_${wl}_kernel_summary.txt
  cp $P/stream_summary.txt profiles/${R}_${wl}_stream_summary.txt
  grep "^{" $P/bench.json | tail -1 > profiles/${R}_${wl}_bench_under_rocprof.json
done
[ -s $G/bench_default.json ] && grep "^{" $G/bench_default.json | tail -1 > profiles/${R}_bench_default.json
[ -s $G/rulebook_voxelize_roofline.jsonl ] && grep "^{" $G/rulebook_voxelize_roofline.jsonl > profiles/${R}_rulebook_voxelize_roofline.jsonl
[ -s $G/fps.txt ] && cp $G/fps.txt profiles/${R}_fps.txt
for n in 2cpu 4cpu; do
  [ -s $G/bench_$n.json ] && grep "^{" $G/bench_$n.json | tail -1 > profiles/${R}_bench_$n.json
done
for wl in lc transfusion_l; do
  [ -s $G/pmc_$wl/pmc_summary.json ] && cp $G/pmc_$wl/pmc_summary.json profiles/${R}_pmc_summary_$wl.json
done
ls -la profiles | grep " ${R}_"
