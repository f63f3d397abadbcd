#!/bin/bash
# gpurun_wait.sh TIMEOUT 'command' -- call gpurun; while it answers "no box or slot free right now" (exit status 3: nothing ran, nothing
# was charged) wait a minute and ask again.  Any other status (the command ran, or was refused) is final.
t="$1"; shift
for i in $(seq 1 40); do
    /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
    rc=$?
    [ $rc -ne 3 ] && exit $rc
    sleep 60
done
exit 3
