#!/bin/bash
# repeat the forced-shard rehearsal until it fails once (intermittent SIGABRT after the line, round 5); usage: loop_bench.sh [runs] [extra bench args]
runs=${1:-30}; shift
for i in $(seq 1 $runs); do
  python bench.py --gpus 1 --force-shard --plan interleave --exchange kv --workload tiny --steps 2 --warmup 1 "$@" > /tmp/b.out 2> /tmp/b.err
  rc=$?
  n=$(grep -c '^{"metric"' /tmp/b.out)
  echo "run $i rc=$rc lines=$n"
  if [ $rc -ne 0 ] || [ $n -ne 1 ]; then echo "---- stderr"; grep -v "amdgpu.ids\|^frame #\|^  File\|^    " /tmp/b.err | head -60 | cut -c1-600; break; fi
done
