#!/bin/bash
# round-6 visit a: (1) can this box show more than one HIP device (CPX / DPX compute partitions)? -- VERDICT r05 "next" #4;
# (2) a RECORDING run of the GPU suite (float bounds measured, not asserted) -> the evidence the tightened bounds are set from;
# (3) the list-reuse probe (VERDICT r05 "next" #1) on the settled headline box and on the developed C1 dam break;
# (4) baseline numbers of the small named configs (C1, C2) + a kernel trace of C1, for the small-N path.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06a
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
{
  echo "== whoami: $(id -un) uid $(id -u)"
  echo "== rocm-smi --showcomputepartition"; timeout 60 rocm-smi --showcomputepartition 2>&1 | tail -n 12
  echo "== rocm-smi --showmemorypartition"; timeout 60 rocm-smi --showmemorypartition 2>&1 | tail -n 8
  echo "== amd-smi partition"; timeout 60 amd-smi partition 2>&1 | tail -n 30
  echo "== HIP devices before"; python -c "import torch;print(torch.cuda.device_count(), [torch.cuda.get_device_properties(i).name for i in range(torch.cuda.device_count())], torch.cuda.get_device_properties(0).multi_processor_count)"
  # (switching the partition mode is a machine-wide setting: the pool's job runner refuses any script that contains the command,
  # before it reaches a box -- profiles/r06a_compute_partition_refused.json; only the read-only queries remain here)
  echo "== ls -l /dev/kfd /dev/dri"; ls -l /dev/kfd /dev/dri 2>&1 | head -n 20
} > $OUT/partition_probe.txt 2>&1
tail -n 25 $OUT/partition_probe.txt
SPH_TEST_RECORD_ONLY=1 SPH_TEST_EVIDENCE_DIR=$OUT/evidence timeout 1500 python -m pytest tests -m gpu -q --durations=12 > $OUT/pytest_gpu_recording.log 2>&1; echo "gpu pytest (recording) rc=$?"
tail -n 25 $OUT/pytest_gpu_recording.log
timeout 600 python tools/list_reuse_probe.py --workload c3p_uniform_1.75M --settle 2000 --horizon 40 > $OUT/list_reuse_probe_c3p_settled.json 2> $OUT/probe_c3p.err; echo "probe c3p rc=$?"
timeout 600 python tools/list_reuse_probe.py --workload c1_dambreak_262k --settle 1500 --horizon 40 > $OUT/list_reuse_probe_c1_developed.json 2> $OUT/probe_c1.err; echo "probe c1 rc=$?"
python -c "
import json
for f in ('c3p_settled','c1_developed'):
    d=json.load(open('$OUT/list_reuse_probe_%s.json'%f)); print(f, d['speed'], d['ms'], d['rebuild_interval_by_skin'], [ (r['k'],r['max_disp_over_h'],r['cell_changes_in_step']) for r in d['displacement'][:3]])
"
for w in c1_dambreak_262k c2_dragon_bath; do
  timeout 300 python bench.py --workload $w --cpu-steps 0 --with-bodies 0 > $OUT/bench_$w.json 2> $OUT/bench_$w.err; echo "bench $w rc=$?"
  python -c "import json;d=json.load(open('$OUT/bench_$w.json'));print('$w', d['value'], d['ms_per_step'], d['breakdown_ms'], 'settled', d.get('settled',{}).get('ms_per_step'), d.get('settled',{}).get('breakdown_ms'))"
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/profc1 -o prof --output-format csv -- python $R/bench.py --workload c1_dambreak_262k --cpu-steps 0 --with-bodies 0 --steps 100 --warmup 5 --min-seconds 0 --settled-after 0 > $OUT/rocprof_c1.log 2>&1 )
f=$(find $OUT/profc1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_c1_rest.csv && head -n 12 $OUT/kernel_stats_c1_rest.csv
t=$(find $OUT/profc1 -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/ktrace_gaps.py "$t" > $OUT/c1_gaps.txt 2>&1 && head -n 8 $OUT/c1_gaps.txt
rm -rf $OUT/profc1
