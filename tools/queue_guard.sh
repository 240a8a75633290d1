#!/bin/bash
# The lane scheduler's queue guard (csrc/sched.hip: lanes_allowed): search / train step time (tools/lanes_probe.py time N, lanes on) with
# GPU_MAX_HW_QUEUES at its default and overridden, the guard in force and bypassed (SENAS_SCHED_TRUST_QUEUES=1).
#   bash tools/queue_guard.sh [steps] > profiles/r5_queue_guard.txt
steps=${1:-20}
run() {   # label, env...
    label=$1; shift
    out=$(env "$@" LANES_ONLY=1 timeout -k 10 240 python3 tools/lanes_probe.py time $steps 2>/tmp/qg.err | tr '\n' ' ')
    note=$(grep -m1 "senas sched" /tmp/qg.err)
    echo "$label  $out  ${note}"
}
echo "# tools/queue_guard.sh $steps: lanes on, search step / derived train step under replay"
run "default queues, guard on              " A=1
run "GPU_MAX_HW_QUEUES=2, guard on          " GPU_MAX_HW_QUEUES=2
run "GPU_MAX_HW_QUEUES=2, guard bypassed   " GPU_MAX_HW_QUEUES=2 SENAS_SCHED_TRUST_QUEUES=1
run "GPU_MAX_HW_QUEUES=3, guard bypassed   " GPU_MAX_HW_QUEUES=3 SENAS_SCHED_TRUST_QUEUES=1
run "GPU_MAX_HW_QUEUES=6, guard on (serial)" GPU_MAX_HW_QUEUES=6
run "GPU_MAX_HW_QUEUES=6, guard bypassed   " GPU_MAX_HW_QUEUES=6 SENAS_SCHED_TRUST_QUEUES=1
