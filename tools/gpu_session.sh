#!/bin/bash
# Run GPU steps one after another on the gpurun box; a step that TIMES OUT or is
# killed stops the session (no further GPU work after a hang), an ordinary
# failure does not.  Usage: tools/gpu_session.sh "<secs> <name> <command>" ...
mkdir -p gpurun_out
for spec in "$@"; do
    secs=${spec%% *}; rest=${spec#* }; name=${rest%% *}; cmd=${rest#* }
    echo "=== [$name] (limit ${secs}s): $cmd"
    start=$(date +%s)
    timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
    rc=$?
    echo "=== [$name] exit $rc after $(( $(date +%s) - start ))s"
    tail -n 25 "gpurun_out/$name.log"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then
        echo "=== [$name] timed out / killed: stopping the session"
        exit $rc
    fi
done
exit 0
