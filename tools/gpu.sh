#!/bin/bash
# usage: tools/gpu.sh LOGFILE TIMEOUT 'command' — gpurun with retries while no GPU slot is free (exit code 3)
LOG=$1; TMO=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $TMO -- "$@" > $LOG 2>&1; rc=$?
  if [ $rc -ne 3 ]; then break; fi
  sleep 60
done
echo "gpurun rc=$rc"
