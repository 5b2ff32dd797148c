#!/bin/bash
# A/B of the tree's build against raytracer_amd/lib/variants/prev.so in ONE GPU call: radiance margins against the
# llvmpipe fixtures and the oracle for both, then the metric frame and room.obj alternating (scripts/ab_prev.sh with cases).
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
PREV=$REPO/raytracer_amd/lib/variants/prev.so
echo "== margins, new"; timeout -k 10 300 python scripts/radiance_margin.py || exit 1
echo "== margins, prev"; timeout -k 10 300 python scripts/radiance_margin.py $PREV || exit 1
run() { timeout -k 10 200 python bench.py --cases room --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['cases']['room']
print(d['ms_per_step'], d['latency_ms_one_frame'], r['ms_per_frame'], r['latency_ms_one_frame'])"; }
for i in 1 2 3; do
  unset RT_HIP_LIB; a=$(run) || exit 1
  export RT_HIP_LIB=$PREV; b=$(run) || exit 1
  echo "new $a | prev $b"
done
