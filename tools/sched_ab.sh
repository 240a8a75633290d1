#!/bin/bash
# Lane-scheduler policy A/B on one box: search step / derived train step under replay (tools/lanes_probe.py time N, lanes on).
#   bash tools/sched_ab.sh [steps] > profiles/r5_sched_policies.txt
steps=${1:-30}
run() {   # label, env...
    label=$1; shift
    out=$(env "$@" LANES_ONLY=1 timeout -k 10 300 python3 tools/lanes_probe.py time $steps 2>/tmp/ab.err | tr '\n' ' ')
    echo "$label  $out"
}
echo "# tools/sched_ab.sh $steps: lanes on, search step / derived train step under replay"
run "critical, 2 refinements (default)      " A=1
run "critical, no refinement                " SENAS_SCHED_REFINE=0
run "critical, 4 refinements                " SENAS_SCHED_REFINE=4
run "critical + stream priorities + classes " SENAS_SCHED_PRIORITY=1
run "critical, node counts instead of times " SENAS_SCHED_NO_TIMING=1
run "chain (round 4 policy), down lane      " SENAS_SCHED_POLICY=chain
run "chain, down cells on the origin stream " SENAS_SCHED_POLICY=chain SENAS_DOWN_LANE=0
