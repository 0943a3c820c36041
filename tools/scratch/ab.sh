cd $GRAFT_REPO_ROOT
for st in 0 2000; do
python bench.py --steps 100 --warmup 20 --cpu-steps 0 --min-seconds 0 --settled-after 0 --settle $st ${AB_ARGS} 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('settle',$st,j['ms_per_step'],j.get('breakdown_ms'))"
done
