#!/bin/bash
# The head cell of the search step launch by launch under two builds of the library (A = senas_amd/libsenas_hip_old.so, kept
# from before a kernel change; B = the current one), plus the replayed step time of each:
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/lib_ab.sh <tag> [cell ...]'
set -e -o pipefail
tag=${1:-ab}; shift || true
cells=${@:-head}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cp $root/senas_amd/libsenas_hip.so /tmp/lib_new.so
for v in old new; do
  if [ $v = old ]; then cp $root/senas_amd/libsenas_hip_old.so $root/senas_amd/libsenas_hip.so; else cp /tmp/lib_new.so $root/senas_amd/libsenas_hip.so; fi
  rocprofv3 --kernel-trace -d $out/trace_$v -o cells --output-format csv -- python3 $root/tools/lane_timeline.py search --serial --order $out/order_$v.json > $out/timeline_$v.log 2>&1
  (cd $root && python3 tools/cell_kernels.py $out/trace_$v $out/order_$v.json --ordered $cells > $out/cells_$v.txt)
  rm -rf $out/trace_$v
  (cd $root && python3 bench.py --steps 40 --lp-steps 0 --search-steps 40 --no-cpu-baseline > $out/bench_$v.log 2> $out/bench_$v.err) || true
  tail -1 $out/bench_$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); s=d.get('search_step',{}); print('$v', 'train ms/step', d.get('ms_per_step'), 'search ms/step', s.get('ms_per_step'))" || true
done
cp /tmp/lib_new.so $root/senas_amd/libsenas_hip.so
echo done
