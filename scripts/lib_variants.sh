#!/bin/bash
# bench.py (metric frame, then room.obj) under other builds of librt_hip.so (raytracer_amd/lib/variants/*.so, RT_HIP_LIB)
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$REPO"
for lib in "" raytracer_amd/lib/variants/*.so; do
  export RT_HIP_LIB=${lib:+$REPO/$lib}
  [ -z "$lib" ] && unset RT_HIP_LIB
  a=$(timeout -k 10 200 python bench.py --cases none --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])") || exit 1
  b=$(timeout -k 10 200 python bench.py --cases none --no-cpu-baseline --scene room --steps 10 --warmup 3 2>/dev/null | python3 -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])") || exit 1
  echo "[${lib:-default}] scene $a room $b"
done
