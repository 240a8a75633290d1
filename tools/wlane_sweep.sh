#!/bin/bash
# Step times against the number of chains the lane scheduler may use (train:search), one short bench per setting; MODES is kept
# for builds that take SENAS_WLANE_MODE (a round-4 prototype, profiles/r4_wlane_modes.txt; the shipped library ignores it).
# Stops at the first failure.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'LANESETS="4:5 4:6 4:7" bash tools/wlane_sweep.sh gpurun_out/lanes.txt'
out=${1:-gpurun_out/wlane.txt}
mkdir -p $(dirname $out)
: > $out
for lanes in ${LANESETS:-4:5 4:6 3:4}; do
  lt=${lanes%%:*}; ls=${lanes##*:}
  for mode in ${MODES:-0}; do
    SENAS_WLANE_MODE=$mode SENAS_MAX_LANES=$lt SENAS_SEARCH_LANES=$ls timeout -k 10 200 python3 bench.py --steps ${STEPS:-40} --search-steps ${STEPS:-40} --lp-steps 0 --no-cpu-baseline > /tmp/wl.json 2> /tmp/wl.err || { echo "mode $mode lanes $lt/$ls: FAILED" >> $out; tail -3 /tmp/wl.err >> $out; exit 1; }
    python3 -c "
import json
d = json.loads(open('/tmp/wl.json').read().strip().splitlines()[-1])
print('wlane mode $mode  lanes train $lt search $ls:  train %.3f ms  search %.3f ms' % (d['ms_per_step'], d['search_step']['ms_per_step']))" >> $out
  done
done
cat $out
