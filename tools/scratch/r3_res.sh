cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for rep in 1 2; do
for r in 0 4 8 16; do
echo "reserve=$r: $(MSMD_CU_RESERVE=$r timeout 300 python bench.py --no-also --no-cpu-baseline 2>/dev/null | tail -1 | cut -c100-200)"
done; done
for r in 0 8; do
echo "TL reserve=$r: $(MSMD_CU_RESERVE=$r timeout 300 python bench.py --workload transfusion_l --no-also --no-cpu-baseline 2>/dev/null | tail -1 | cut -c80-180)"
done
