R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06c; mkdir -p $OUT; cd $R
timeout 600 python -m pytest -m gpu -x -q tests/test_gpu_image_glue.py tests/test_gpu_detector.py 2>&1 | tail -12
for f in 0 1; do
  MSMD_FG_FUSED=$f timeout 400 python bench.py --workload lc_img --no-also --no-cpu-baseline --no-profile > $OUT/imgf_$f.json 2> $OUT/imgf_$f.err
  python - <<PY
import json
d = json.load(open("$OUT/imgf_$f.json")); g = d["image_glue"]
print("MSMD_FG_FUSED=$f", d["value"], "samples/s", d["ms_per_step"], "ms; glue", g["virtual_points_from_images_ms"], "ms")
PY
done
