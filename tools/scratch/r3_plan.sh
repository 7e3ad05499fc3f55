cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2; do
for f in 1 0; do
echo "fused=$f: $(MSMD_TILE_FUSED=$f timeout 300 python bench.py --no-also --no-cpu-baseline 2>/dev/null | tail -1 | cut -c100-200)"
done; done
for f in 1 0; do
echo "TL fused=$f: $(MSMD_TILE_FUSED=$f timeout 300 python bench.py --workload transfusion_l --no-also --no-cpu-baseline 2>/dev/null | tail -1 | cut -c80-180)"
done
