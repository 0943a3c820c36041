R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/valu_split; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for st in 0 2000; do
for m in 0 7 4 16 32 1; do
  B="python $R/bench.py --steps 5 --warmup 20 --cpu-steps 0 --min-seconds 0 --settled-after 0 --settle $st --ablate-mask $m"
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/s${st}_m$m -o p -- $B > $OUT/s${st}_m$m.log 2>&1
done; done
python3 - <<'P'
import csv,glob,os,collections
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/valu_split'
for d in sorted(glob.glob(out+'/s*_m*/')):
    f=glob.glob(d+'/**/*counter_collection.csv',recursive=True)
    if not f: print(d,'no csv'); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k=r['Kernel_Name']
        if 'gather_brick' not in k: continue
        acc[k[:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        print(os.path.basename(d.rstrip('/')),k,{c:round(sum(x[-5:])/len(x[-5:])) for c,x in v.items()})
P
