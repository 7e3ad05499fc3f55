L="2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l 2:transfusion_l"
for c in $COMMITS; do
  cd $GRAFT_REPO_ROOT/_bisect/$c
  export PYTHONPATH=$GRAFT_REPO_ROOT/_bisect/$c
  echo "== $c: legs completed of 16:"
  timeout 300 python -X faulthandler tools/scratch/seq_repro.py $L 2>&1 | grep -v "amdgpu.ids\|steps_total" > /tmp/out_$c.txt
  grep -c "^2:transfusion_l [0-9]" /tmp/out_$c.txt
  grep -v "^2:transfusion_l [0-9]" /tmp/out_$c.txt | head -4 | cut -c1-200
done
