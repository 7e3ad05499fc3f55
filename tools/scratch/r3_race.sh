cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
L="2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:lc_b4 2:lc_b4 2:lc_b4 2:lc_b4 2:lc_b4 2:lc_b4"
timeout 600 python -X faulthandler tools/scratch/seq_repro.py $L 2>&1 | grep -v "amdgpu.ids\|steps_total" > /tmp/out.txt
echo "legs completed of 22: $(grep -c '^2:[a-z_0-9]* [0-9]' /tmp/out.txt)"
grep -v '^2:[a-z_0-9]* [0-9]' /tmp/out.txt | head -5 | cut -c1-200
