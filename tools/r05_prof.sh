# round 5 evidence: the image-glue leg alone, PMC passes (HBM bytes, MFMA busy, L2) under the
# pipelined schedule.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05f; mkdir -p $O; cd $R
timeout 400 python bench.py --workload lc_img --no-also --no-cpu-baseline > $O/bench_lc_img.json 2> $O/bench_lc_img.err
tail -c 1200 $O/bench_lc_img.json; tail -5 $O/bench_lc_img.err
bash tools/pmc_collect.sh lc $O/pmc_lc > $O/pmc.log 2>&1
tail -3 $O/pmc.log
