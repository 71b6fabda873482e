#!/bin/bash
# retry.sh <timeout> <script on the box>: gpurun with retries while every slot of the pod is busy (exit code 3)
t=$1; shift
for i in 1 2 3 4 5 6 7 8 9 10; do
  /usr/local/graft/bin/gpurun --timeout $t -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 120
done
exit 3
