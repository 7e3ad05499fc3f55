# stream-K ranges tapered by arrival order (MSMD_SK_TAPER, per mille): parity + LC A/B
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06e; mkdir -p $OUT; cd $R
for t in 0 200; do
  echo "== tests with MSMD_SK_TAPER=$t"
  MSMD_SK_TAPER=$t timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_kernels.py -k "split or production or contention" 2>&1 | tail -3
done
MSMD_SK_TAPER=200 timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_production.py 2>&1 | tail -2
for t in 0 100 200 300 0 200; do
  MSMD_SK_TAPER=$t timeout 300 python bench.py --no-also --no-cpu-baseline --no-profile > $OUT/t$t.json 2> $OUT/t$t.err
  python - <<PY
import json
try:
    d = json.load(open("$OUT/t$t.json")); print("taper $t: %.1f samples/s %.3f ms" % (d["value"], d["ms_per_step"]))
except Exception as e:
    print("taper $t failed", e); print(open("$OUT/t$t.err").read()[-500:])
PY
done
