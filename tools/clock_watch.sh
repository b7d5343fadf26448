#!/bin/bash
# samples the GPU's clocks / power while a command runs: clock_watch.sh <out> -- cmd...
out=$1; shift; shift
( while true; do echo "$(date +%s.%N) $(rocm-smi --showclocks --showpower --showtemp --csv 2>/dev/null | tail -n +2 | head -1)"; sleep 0.2; done ) > $out 2>&1 &
W=$!
"$@"
kill $W
