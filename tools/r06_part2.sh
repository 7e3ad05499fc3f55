R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06d; mkdir -p $OUT; cd $R
run() {  # tag, env...
  tag=$1; shift
  for i in 1 2; do
    env "$@" timeout 300 python bench.py --no-also --no-cpu-baseline --no-profile > $OUT/$tag.$i.json 2> $OUT/$tag.$i.err
    python - <<PY
import json
try:
    d = json.load(open("$OUT/$tag.$i.json")); print("$tag run $i: %.1f samples/s %.3f ms" % (d["value"], d["ms_per_step"]))
except Exception as e:
    print("$tag run $i failed:", e); print(open("$OUT/$tag.$i.err").read()[-600:])
PY
  done
}
run base X=1
run idx8 MSMD_CU_PARTITION=8,0 MSMD_RESERVE_CUS=8
run idx16 MSMD_CU_PARTITION=16,0 MSMD_RESERVE_CUS=16
run idx16pp MSMD_CU_PARTITION=16,0 MSMD_RESERVE_CUS=16 MSMD_FWD_PP_MIN=97
run idx48 MSMD_CU_PARTITION=48,0 MSMD_RESERVE_CUS=48
run basepp MSMD_FWD_PP_MIN=97
