#!/bin/bash
# Every example script of the reference under BOTH readings of Taichi -- this repo's DSL shim (the product) and tests/refexec with the
# reference's own Scene (run.py) -- on the same random stream and elementary functions; the authored grids must be identical.
# Build container only (reads /root/reference); ~7 minutes on 8 cores.      bash tools/refexec_examples/check_all.sh [outdir]
OUT=${1:-/tmp/refexec_examples}; mkdir -p $OUT
HERE=$(cd "$(dirname "$0")" && pwd)
SCRIPTS="main.py example1.py example2.py example3.py example4.py example5.py example6.py example7.py example8.py example9.py example10.py"
for s in $SCRIPTS; do (cd /tmp && REFEXEC_EXAMPLES_OUT=$OUT python $HERE/run.py $s > $OUT/$s.log 2>&1) & done
wait
rc=0
for s in $SCRIPTS; do (cd /tmp && REFEXEC_EXAMPLES_OUT=$OUT python $HERE/cmp.py $s) || rc=1; done
[ $rc == 0 ] && echo "all eleven scripts author identical grids under both readings" || echo "DIFFERENCES (above)"
exit $rc
