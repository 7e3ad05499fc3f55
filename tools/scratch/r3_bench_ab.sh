cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03d; mkdir -p $OUT
timeout 300 python bench.py --no-cpu-baseline --no-also > $OUT/lc_block.json 2> $OUT/lc_block.err
MSMD_WGRAD=var timeout 300 python bench.py --no-cpu-baseline --no-also > $OUT/lc_var.json 2> $OUT/lc_var.err
python - <<'PY'
import json
for n in ("block","var"):
    try:
        d=json.loads(open("gpurun_out/r03d/lc_%s.json"%n).read().strip().splitlines()[-1])
        k=d["roofline"]["all_conv_kernels"]
        print(n, d["value"], d["ms_per_step"], {x:(v["ms"],v["tflops"],v["launches"]) for x,v in k.items() if "wgrad" in x})
    except Exception as e:
        print(n, "ERR", e)
PY
