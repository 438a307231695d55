# same-box A/B with the switch set to 0 / unset: tools/ab_env2.sh VAR
var=$1
for rep in 1 2 3 4; do
  for val in unset 0; do
    if [ $val = 0 ]; then export $var=0; else unset $var; fi
    python bench.py --steps 400 --warmup 40 --no-extras --no-cpu-baseline > gpurun_out/ab2_$val.log 2>/dev/null
    python -c "
import json
d=json.loads([l for l in open('gpurun_out/ab2_$val.log') if l.startswith('{')][-1])
print('$var=$val', d['value'], d['ms_per_step'], d['stages_ms']['detector'], d['p99_latency_ms'])"
  done
done
