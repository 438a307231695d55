# same-box A/B of an environment switch: tools/ab_env.sh VAR  -> bench with VAR unset / VAR=1, twice each
var=$1
for rep in 1 2; do
  for val in 0 1; do
    if [ $val = 1 ]; then export $var=1; else unset $var; fi
    python bench.py --steps 500 --warmup 50 --no-extras --no-cpu-baseline > gpurun_out/ab_$val.log 2>/dev/null
    python -c "
import json
d=json.loads([l for l in open('gpurun_out/ab_$val.log') if l.startswith('{')][-1])
print('$var=$val', d['value'], d['ms_per_step'], d['stages_ms']['detector'], d['p99_latency_ms'])"
  done
done
