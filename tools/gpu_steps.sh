#!/bin/bash
# Runs a list of GPU steps on the box, each under its own timeout; stops at the first step that timed out or was killed
# (no GPU work after a hang), carries on after ordinary failures.  Usage: bash tools/gpu_steps.sh <tag> "<secs>|<name>|<cmd>" ...
TAG=$1; shift
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$OLDPWD"
for step in "$@"; do
  secs=${step%%|*}; rest=${step#*|}; name=${rest%%|*}; cmd=${rest#*|}
  echo "=== $name (limit ${secs}s): $cmd"
  timeout -k 10 "$secs" bash -c "$cmd" > "$OUT/$name.log" 2>&1
  rc=$?
  echo "=== $name exit $rc"; tail -n 6 "$OUT/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out: stopping"; exit 1; fi
done
exit 0
