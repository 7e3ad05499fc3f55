cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
L="3:lc 3:transfusion_l 3:lc_b4 2:lc_b4 2:transfusion_l 1:lc 3:lc_tail 3:lc 2:lc 3:transfusion_l 3:lc_b4 2:lc_b4 3:lc_full 3:lc"
timeout 900 python -X faulthandler tools/scratch/seq_repro.py $L 2>&1 | grep -v "amdgpu.ids\|steps_total" > /tmp/out.txt
echo "legs completed of 14: $(grep -c '^[123]:[a-z_0-9]* [0-9]' /tmp/out.txt)"
grep -v '^[123]:[a-z_0-9]* [0-9]' /tmp/out.txt | head -5 | cut -c1-300
cat /tmp/out.txt | grep '^[123]:' | tr '\n' ' '
