#!/bin/bash
# One parameterised GPU visit (replaces the per-call scripts of rounds 1-2).  Usage: bash tools/gpu_round.sh <tag> [parts]
#   parts (default "tests pmc kstats bench variants"):
#     tests    : pytest -m gpu
#     pmc      : rocprofv3 PMC passes, from rest and settled -> gpurun_out/<tag>/pmc_traffic.json (+ summaries)
#     kstats   : rocprofv3 kernel-trace stats of the default bench line's two states
#     bench    : the default bench line + the other workloads
#     variants : A/B table  partition (adaptive / fixed bricks) x emission (group-sorted / baseline)
#     native   : kernel trace of the one-rank native RCCL worker + slab driver overhead at world = 1
#     fastmath : A/B of the fast-math choice (tools/fastmath_ab.py)
#     nativemp : only the multi-process native-exchange tests (tests/fake_rccl)
#     timeline : per-workgroup time line of the two brick sweeps (profiling build, tools/brick_timeline.py)
#     bodies   : the default bench line's with_bodies object alone
#     pmcdf    : FETCH_SIZE / WRITE_SIZE passes over the DFSPH line -> profiles/pmc_traffic_dfsph.json
#     ranks2   : bench.py --gpus 2 without a launcher on the one GPU (torch transport / NativeTransport through tests/fake_rccl)
#     dfgaps   : GPU idle time between the kernels of the DFSPH line
TAG=${1:-round}; shift
PARTS=${*:-tests pmc kstats bench variants}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
export TMPDIR=/tmp
has() { case " $PARTS " in *" $1 "*) return 0;; *) return 1;; esac; }
if has tests; then
  # (the tests leave their measured curves / error tables under $SPH_TEST_EVIDENCE_DIR only when it is set)
  SPH_TEST_EVIDENCE_DIR=$OUT timeout 1200 python -m pytest tests -m gpu -q --durations=15 ${PYTEST_ARGS:-} > $OUT/pytest_gpu.log 2>&1; echo "gpu pytest rc=$?"
  tail -n 30 $OUT/pytest_gpu.log
fi
if has pmc; then
  bash tools/gpu_pmc.sh ${TAG}_rest > $OUT/pmc_rest.log 2>&1; tail -n 2 $OUT/pmc_rest.log
  bash tools/gpu_pmc.sh ${TAG}_settled --settle 2000 > $OUT/pmc_settled.log 2>&1; tail -n 2 $OUT/pmc_settled.log
  python tools/refresh_pmc.py gpurun_out/pmc_${TAG}_rest $OUT/pmc_traffic.json --settled gpurun_out/pmc_${TAG}_settled --tail 20 > $OUT/pmc_brief.json 2> $OUT/refresh.err; echo "refresh rc=$?"
  rm -rf gpurun_out/pmc_${TAG}_rest gpurun_out/pmc_${TAG}_settled
  cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
  head -c 1500 $OUT/pmc_brief.json
fi
if has pmcdf; then   # HBM traffic of the DFSPH sweeps (FETCH_SIZE / WRITE_SIZE passes over bench.py --solver dfsph) -> profiles/pmc_traffic_dfsph.json
  PMC_PASSES="fetch write" bash tools/gpu_pmc.sh ${TAG}_dfsph --solver dfsph > $OUT/pmc_dfsph.log 2>&1; tail -n 2 $OUT/pmc_dfsph.log
  python tools/refresh_pmc.py gpurun_out/pmc_${TAG}_dfsph $OUT/pmc_traffic_dfsph.json --solver dfsph --tail 0 > $OUT/pmc_dfsph_brief.json 2> $OUT/refresh_dfsph.err; echo "refresh dfsph rc=$?"
  rm -rf gpurun_out/pmc_${TAG}_dfsph
  cp $OUT/pmc_traffic_dfsph.json profiles/pmc_traffic_dfsph.json
  head -c 1200 $OUT/pmc_dfsph_brief.json
fi
if has ranks2; then   # bench.py --gpus 2 WITHOUT a launcher, both ranks on this one GPU (gloo bootstrap): torch transport and NativeTransport through the librccl stand-in
  FAKE=$(python -c "import importlib.util,os;s=importlib.util.spec_from_file_location('b','tests/fake_rccl/build.py');m=importlib.util.module_from_spec(s);s.loader.exec_module(m);print(m.build())")
  SPH_DIST_BACKEND=gloo SPH_C4_SCALE=0.4 SPH_TRANSPORT=torch timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --settled-after 200 --preheat-ms 0 > $OUT/bench_2ranks_one_gpu_torch.json 2> $OUT/bench_2ranks_torch.err; echo "2 ranks torch rc=$?"
  SPH_DIST_BACKEND=gloo SPH_C4_SCALE=0.4 SPH_TRANSPORT=native SPH_RCCL_LIB=$FAKE FAKE_RCCL_SLOT_BYTES=1048576 FAKE_RCCL_SLOTS=3 FAKE_RCCL_TIMEOUT_S=25 timeout 600 python bench.py --gpus 2 --steps 20 --warmup 5 --settled-after 200 --preheat-ms 0 > $OUT/bench_2ranks_one_gpu_native.json 2> $OUT/bench_2ranks_native.err; echo "2 ranks native rc=$?"
  for f in torch native; do python -c "import json;d=json.load(open('$OUT/bench_2ranks_one_gpu_$f.json'));print('$f', d['value'], d['config']['transport'], d['config']['comm'], d['config']['particles_owned_per_rank'], 'c4', d['c4_dambreak'].get('from_rest',{}).get('value'), d['c4_dambreak'].get('conserved'))"; done
fi
if has dfgaps; then   # GPU idle time between the kernels of the DFSPH line (tools/ktrace_gaps.py)
  ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/profd -o prof --output-format csv -- python $R/bench.py --cpu-steps 0 --steps 30 --warmup 3 --solver dfsph > $OUT/rocprof_dfsph.log 2>&1 )
  f=$(find $OUT/profd -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_dfsph.csv
  t=$(find $OUT/profd -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python tools/ktrace_gaps.py "$t" > $OUT/dfsph_gaps.txt 2>&1 && head -n 6 $OUT/dfsph_gaps.txt
  rm -rf $OUT/profd
fi
if has kstats; then
  bash tools/gpu_kstats.sh $TAG "--steps 60 --warmup 5 --min-seconds 0 --settled-after 0" "--steps 60 --warmup 5 --settle 2000 --settled-after 0"
fi
if has bench; then
  timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 1200 $OUT/bench_default.json; echo
  timeout 200 python bench.py --steps 20 --warmup 5 --cpu-steps 0 --with-bodies 0 > $OUT/bench_driver_args.json 2>> $OUT/bench.err
  for w in c1_dambreak_262k c2_dragon_bath c3_armadillo_equiv; do
    timeout 200 python bench.py --steps 100 --warmup 10 --cpu-steps 0 --workload $w --settled-after 0 --min-seconds 0 > $OUT/bench_$w.json 2>> $OUT/bench.err
  done
  timeout 200 python bench.py --steps 100 --warmup 10 --cpu-steps 0 --workload c1_dambreak_262k --settle 2500 --settled-after 0 > $OUT/bench_c1_developed.json 2>> $OUT/bench.err
  timeout 300 python bench.py --steps 30 --warmup 3 --cpu-steps 0 --solver dfsph > $OUT/bench_dfsph_c3p.json 2>> $OUT/bench.err
  timeout 400 python bench.py --steps 50 --warmup 10 --cpu-steps 0 --workload c4_dambreak --settled-after 2500 --min-seconds 0 --max-reps 1 > $OUT/bench_c4_dambreak_one_gpu.json 2>> $OUT/bench.err
  for f in driver_args c1_dambreak_262k c2_dragon_bath c3_armadillo_equiv c1_developed dfsph_c3p c4_dambreak_one_gpu; do python -c "import json;d=json.load(open('$OUT/bench_$f.json'));print('$f',d['value'],d['ms_per_step'],d['breakdown_ms'])"; done
fi
if has variants; then
  timeout 600 python tools/variant_sweep.py --variants 25,0 --shapes 1,0 --steps 60 --settled-steps 80 --out $OUT/variants_partition_x_emission.json > $OUT/variants.log 2>&1
  grep -v "^$" $OUT/variants.log | grep -v amdgpu | tail -n 10
fi
if has native; then
  python - <<'P'
import json, sys
sys.path.insert(0, "tests")
import scenes
json.dump(scenes.fluid_with_rigid_bodies("/tmp/cube_native.obj"), open("/tmp/scene_native.json", "w"))
P
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_native -o prof --output-format csv -- python $R/tests/slab_worker.py native1 0 1 29611 /tmp/res_native.npz /tmp/scene_native.json 12 > $OUT/rocprof_native.log 2>&1 )
  f=$(find $OUT/prof_native -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/native_rccl_one_rank_kernel_stats.csv && grep -i "nccl\|rccl" "$f" | cut -c1-200
  rm -rf $OUT/prof_native
  timeout 300 python tools/slab_overhead.py > $OUT/slab_overhead_world1.txt 2>&1; grep -v amdgpu $OUT/slab_overhead_world1.txt
fi
if has fastmath; then   # A/B of the fast-math choice (SPH_OPT_EXACT_MATH): tools/fastmath_ab.py
  timeout 900 python tools/fastmath_ab.py $TAG > $OUT/fastmath_ab.log 2>&1; echo "fastmath rc=$?"; tail -n 3 $OUT/fastmath_ab.log | cut -c1-1500
fi
if has nativemp; then   # only the multi-process native-exchange tests (tests/fake_rccl)
  timeout 600 python -m pytest tests/test_distributed.py -m gpu -x -q -k "native_exchange" --durations=10 > $OUT/pytest_native_mp.log 2>&1; echo "native mp rc=$?"
  tail -n 25 $OUT/pytest_native_mp.log
fi
if has timeline; then   # per-workgroup time line of the two brick sweeps (profiling build)
  timeout 600 python tools/brick_timeline.py --out $OUT/brick_timeline.txt > $OUT/brick_timeline.log 2>&1; echo "timeline rc=$?"; grep -v amdgpu $OUT/brick_timeline.log | tail -n 40
fi
if has bodies; then     # the default line's with_bodies object alone (C3 with immersed bodies)
  timeout 300 python bench.py --steps 100 --warmup 10 --cpu-steps 0 --settled-after 0 --min-seconds 0 > $OUT/bench_with_bodies.json 2>> $OUT/bench.err
  python -c "import json;d=json.load(open('$OUT/bench_with_bodies.json'));print('with_bodies', json.dumps(d.get('with_bodies'))[:900])"
fi
