#!/bin/bash
# clocks and power while bench.py runs (four chains, then --depth 1): rocm-smi sampled once a second
mkdir -p gpurun_out
for depth in 4 1; do
  timeout -k 10 200 python bench.py --steps 12000 --warmup 40 --depth $depth --no-cpu-baseline --no-extras > gpurun_out/clk_bench_$depth.json 2> gpurun_out/clk_bench.err &
  BP=$!
  sleep 22        # import, build, tune
  echo "== depth $depth"
  for i in 1 2 3 4 5 6; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power" | tr -s ' ' | tr '\n' ';'; echo
    sleep 1
  done
  wait $BP
  python -c "
import json; d=json.loads(open('gpurun_out/clk_bench_$depth.json').read().strip().splitlines()[-1]); print('frames/s', d['value'], 'ms/tick', d['ms_per_step'])"
done
