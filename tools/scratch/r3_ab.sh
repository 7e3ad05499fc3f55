# usage: r3_ab.sh ENVVAR val1 val2 [reps]  -- LC bench A/B on one box, alternating
cd $GRAFT_REPO_ROOT
V=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq $N); do for x in $A $B; do
env $V=$x timeout 300 python bench.py --no-cpu-baseline --no-also --no-profile 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$x', d['value'], d['ms_per_step'])"
done; done
