# CU partition experiment: index / search streams on CUs of their own (stream CU masks), conv
# grids sized to the rest -- LC samples/s, two runs each, one call
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
run nn4 MSMD_CU_PARTITION=0,4 MSMD_RESERVE_CUS=4
run nn8 MSMD_CU_PARTITION=0,8 MSMD_RESERVE_CUS=8
run idx32 MSMD_CU_PARTITION=32,0 MSMD_RESERVE_CUS=32
run idx24nn8 MSMD_CU_PARTITION=24,8 MSMD_RESERVE_CUS=32
run idx56nn8 MSMD_CU_PARTITION=56,8 MSMD_RESERVE_CUS=64
run idx24nn8pp MSMD_CU_PARTITION=24,8 MSMD_RESERVE_CUS=32 MSMD_FWD_PP_MIN=97
