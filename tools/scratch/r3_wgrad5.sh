cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r03g; mkdir -p $OUT
for d in 0 126; do
rm -rf $OUT/p
MSMD_WGRAD_DBG=$d timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/p -o s -- python $R/tools/wgrad_ablate.py > $OUT/log.txt 2>&1
t=$(find $OUT/p -name "*kernel_trace.csv" | head -1)
python - "$t" $d <<'PY'
import csv,sys,collections
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'wgrad_block' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
d=collections.defaultdict(list)
for a,b in zip(rows,rows[1:]):
    if 'reduce' not in a['Kernel_Name'] and 'reduce' in b['Kernel_Name']:
        d[b['Grid_Size_X']].append(((int(a['End_Timestamp'])-int(a['Start_Timestamp']))/1e3,(int(b['End_Timestamp'])-int(b['Start_Timestamp']))/1e3))
names={'4096':'64x64','16384':'128x128','9216':'96x96 or 192x192','6400':'80x80','7680':'80x96','12288':'96x128', '24576':'128x192','36864':'192x192'}
for k,v in d.items():
    m=sorted(x[0] for x in v); r=sorted(x[1] for x in v)
    print('DBG',sys.argv[2],names.get(k,k), len(v), 'main median %.1f us (min %.1f max %.1f)'%(m[len(m)//2],m[0],m[-1]), 'reduce %.1f'%r[len(r)//2])
PY
done
rm -rf $OUT/p
