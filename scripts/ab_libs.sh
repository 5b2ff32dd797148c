#!/bin/bash
# Several builds of librt_hip.so against each other in ONE GPU call (boxes differ by 1-2 % between calls), alternating,
# ROUNDS times: usage scripts/ab_libs.sh name=path ...   (paths relative to the repository; BENCH_ARGS as for ab_prev.sh)
# prints per build: metric frame ms (in flight), its latency alone, room.obj ms (in flight), its latency alone
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO; mkdir -p gpurun_out
run() { timeout -k 10 200 python bench.py --cases ${CASES:-room} --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d.get('cases') or {}
print(d['ms_per_step'], d['latency_ms_one_frame'], *[f\"{k} {v['ms_per_frame']} {v['latency_ms_one_frame']}\" for k,v in c.items()])"; }
for i in $(seq ${ROUNDS:-3}); do
  for nv in "$@"; do
    export RT_HIP_LIB=$REPO/${nv#*=}; r=$(run) || exit 1
    echo "$i ${nv%%=*}: $r"
  done
done
