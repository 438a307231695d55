for mode in "graph:" "nograph:--no-graph" ; do
  for ser in 0 1; do
    name=${mode%%:*}; flag=${mode#*:}
    RVA_SERIAL_HEADS=$ser python bench.py --steps 300 --warmup 20 --no-extras --no-cpu-baseline $flag > gpurun_out/r2_bcmp_${name}_$ser.log 2>/dev/null
    python -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/r2_bcmp_${name}_$ser.log') if l.startswith('{')][-1])
print('$name serial_heads=$ser', d['value'], d['ms_per_step'], d['stages_ms']['detector'], d['p99_latency_ms'])"
  done
done
