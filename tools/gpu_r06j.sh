#!/bin/bash
# round-6 visit j: robustness of the final tree -- `python bench.py --gpus 2 / 4 / 8` with NO launcher, all ranks on the one GPU (gloo bootstrap,
# NativeTransport through the librccl stand-in; the times mean nothing), and a 4,000-step soak of the bench workloads
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06j
mkdir -p $OUT
cd $R
FAKE=$(python -c "import importlib.util,os;s=importlib.util.spec_from_file_location('b','tests/fake_rccl/build.py');m=importlib.util.module_from_spec(s);s.loader.exec_module(m);print(m.build())")
for n in 2 4 8; do
  ( time SPH_DIST_BACKEND=gloo SPH_C4_SCALE=0.5 SPH_TRANSPORT=native SPH_RCCL_LIB=$FAKE FAKE_RCCL_SLOT_BYTES=1048576 FAKE_RCCL_SLOTS=3 FAKE_RCCL_TIMEOUT_S=60 timeout 900 python bench.py --gpus $n --steps 10 --warmup 3 --settled-after 100 --preheat-ms 0 > $OUT/bench_${n}ranks_one_gpu_native.json 2> $OUT/bench_${n}ranks.err ) 2>&1 | grep real
  echo "$n ranks rc=$?"
  python -c "import json;d=json.load(open('$OUT/bench_${n}ranks_one_gpu_native.json'));c=d['c4_dambreak'];print($n, d['value'], d['config']['transport'], d['config']['comm'], d['config']['particles_owned_per_rank'], 'c4:', c.get('error'), c.get('conserved'), c.get('owned_end'), c.get('comm'))"
done
timeout 1500 python tools/soak.py 4000 > $OUT/soak.txt 2>&1; echo "soak rc=$?"; grep -v amdgpu $OUT/soak.txt | tail -n 12
