#!/bin/bash
# samples the shader clock and the socket power (rocm-smi, read only) while one bench run is in flight: is the run power- or clock-limited?
root=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
python3 $root/bench.py --cpu-baseline 0 --extras 0 --steps 20 > $root/gpurun_out/clock_watch_bench.json 2>/dev/null &
pid=$!
sleep 4
for i in $(seq 1 40); do
  /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|Temperature \(Sensor (junction|edge)" | grep -o 'S: ([0-9]*Mhz)\|(W): [0-9.]*\|(C): [0-9.]*' | tr '\n' ' '
  echo
  sleep 0.4
done
wait $pid
python3 -c "import json;print(json.load(open('$root/gpurun_out/clock_watch_bench.json'))['value'])"
