cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout 600 python tools/scratch/seq_repro.py 3:lc 3:transfusion_l 3:lc_tail 3:lc_full 3:lc_b4 1:lc 1:lc_b4 2:transfusion_l 2:lc_b4 2:lc_b4 2:lc 2>&1 | grep -v "amdgpu.ids\|steps_total" | tail -20
