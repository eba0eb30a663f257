#!/bin/bash
# socket power (rocm-smi, read only) every ~0.35 s while a command runs: tools/power_watch.sh <seconds> <command...>
secs=$1; shift
"$@" > /tmp/power_watch_cmd.log 2>&1 &
pid=$!
t0=$(date +%s.%N)
n=$(python3 -c "print(int($secs / 0.35))")
for i in $(seq 1 $n); do
  p=$(/opt/rocm/bin/rocm-smi --showpower 2>/dev/null | grep -o '(W): [0-9.]*' | head -1)
  echo "$(python3 -c "import time;print(round(time.time()-$t0,2))") $p"
  kill -0 $pid 2>/dev/null || break
done
wait $pid
cat /tmp/power_watch_cmd.log
