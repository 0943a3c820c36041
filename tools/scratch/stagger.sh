cd $GRAFT_REPO_ROOT
for st in 0 2000; do for m in 0 $(( (5<<24)|(1<<20) )) $(( (5<<24)|(2<<20) )) $(( (5<<24)|(4<<20) )) $(( (5<<24)|(8<<20) )) $(( (10<<24)|(3<<20) )); do
python bench.py --steps 100 --warmup 20 --cpu-steps 0 --min-seconds 0 --settled-after 0 --settle $st --ablate-mask $m 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('settle',$st,'mask',hex($m),j['ms_per_step'],j.get('breakdown_ms'))"
done; done
