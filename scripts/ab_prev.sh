#!/bin/bash
# A/B of two builds of librt_hip.so in ONE GPU call (boxes differ by 1-2 % between calls): the tree's build against
# raytracer_amd/lib/variants/prev.so (a copy of an earlier build: `cp raytracer_amd/lib/librt_hip.so raytracer_amd/lib/variants/prev.so`
# before the change), alternating, four times.  BENCH_ARGS as for knob_sweep.sh.
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
run() { timeout -k 10 200 python bench.py --cases none --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
for i in 1 2 3 4; do
  unset RT_HIP_LIB; a=$(run) || exit 1
  export RT_HIP_LIB=$REPO/raytracer_amd/lib/variants/prev.so; b=$(run) || exit 1
  echo "new $a prev $b"
done
