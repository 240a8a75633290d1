#!/bin/bash
# Round-5 kernel-side changes A/B on one box (tools/lanes_probe.py time N, lanes on): planar stacked outputs, the one-launch wide node.
#   bash tools/merge_ab.sh [steps] > profiles/r5_planar_wide_ab.txt
steps=${1:-30}
run() { label=$1; shift; out=$(env "$@" LANES_ONLY=1 timeout -k 10 300 python3 tools/lanes_probe.py time $steps 2>/tmp/ab.err | tr '\n' ' '); echo "$label  $out"; }
echo "# tools/merge_ab.sh $steps: search step / derived train step under replay"
run "default (planar parts, wide node)   " A=1
run "interleaved stacked outputs         " SENAS_PLANAR=0
run "two-launch node on small maps       " SENAS_NODE_WIDE=0
run "both off                            " SENAS_PLANAR=0 SENAS_NODE_WIDE=0
run "default again                       " A=1
