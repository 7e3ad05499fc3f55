# image glue: NCHW maps / one conversion pass / channels-last convs -- glue ms + gather GB/s + step
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r06c; mkdir -p $OUT; cd $R
for m in 0 1 2; do
  MSMD_GLUE_NHWC=$m timeout 400 python bench.py --workload lc_img --no-also --no-cpu-baseline --no-profile > $OUT/img_$m.json 2> $OUT/img_$m.err
  python - <<PY
import json
d = json.load(open("$OUT/img_$m.json")); g = d["image_glue"]
print("MSMD_GLUE_NHWC=$m", d["value"], "samples/s", d["ms_per_step"], "ms; glue", g["virtual_points_from_images_ms"], "ms;",
      [(x["map"][2:], x["us"], x["frac_hbm"]) for x in g["fg_gather"]])
PY
done
timeout 600 python -m pytest -m gpu -x -q tests/test_gpu_image_glue.py tests/test_gpu_detector.py 2>&1 | tail -3
