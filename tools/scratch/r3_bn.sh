cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
cp msmdfusion_amd/libmsmd_hip.so /tmp/new.so
for rep in 1 2 3; do
for v in new prev; do
if [ $v = new ]; then cp /tmp/new.so msmdfusion_amd/libmsmd_hip.so; else cp msmdfusion_amd/libmsmd_hip_prev.so msmdfusion_amd/libmsmd_hip.so; fi
echo "$v: $(timeout 300 python bench.py --no-also --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | cut -c100-200)"
done; done
for v in new prev; do
if [ $v = new ]; then cp /tmp/new.so msmdfusion_amd/libmsmd_hip.so; else cp msmdfusion_amd/libmsmd_hip_prev.so msmdfusion_amd/libmsmd_hip.so; fi
echo "TL $v: $(timeout 300 python bench.py --workload transfusion_l --no-also --no-cpu-baseline --no-profile 2>/dev/null | tail -1 | cut -c80-180)"
done
