cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
for f in 1 0; do
echo "bn_from_conv=$f: $(MSMD_BN_FROM_CONV=$f timeout 300 python bench.py --no-also --no-cpu-baseline 2>/dev/null | tail -1 | cut -c100-200)"
done; done
for rep in 1 2; do
for f in 1 0; do
echo "TL bn_from_conv=$f: $(MSMD_BN_FROM_CONV=$f timeout 300 python bench.py --workload transfusion_l --no-also --no-cpu-baseline 2>/dev/null | tail -1 | cut -c80-180)"
done; done
