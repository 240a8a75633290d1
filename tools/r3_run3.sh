#!/bin/bash
# round 3: whole GPU suite, then the search step's steady-state trace (foreign kernels listed)
set -e -o pipefail
root=$(pwd)
out=$root/gpurun_out/r3e
mkdir -p $out
if ! python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; then tail -60 $out/pytest.log; exit 1; fi
tail -3 $out/pytest.log
cd /tmp && export TMPDIR=/tmp
SENAS_TRACE_MARKER=1 rocprofv3 --kernel-trace --output-format csv -d $out/tr -o tr -- python3 $root/tools/search_profile.py 5 > $out/tr.log 2>&1
cd $root
python tools/aten_in_step.py $out/tr 5 > $out/aten.txt
python tools/trace_by_grid.py $out/tr --steady 5 > $out/search_steady.txt
rm -rf $out/tr
tail -2 $out/tr.log
head -40 $out/aten.txt
