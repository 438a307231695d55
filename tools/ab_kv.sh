# same-box A/B of two environment settings: tools/ab_kv.sh "A=1 B=2" "A=3"  (each a space-separated list of VAR=value; alternating runs)
for rep in 1 2 3; do
  for cfg in "$1" "$2"; do
    env $cfg python bench.py --steps 400 --warmup 40 --no-extras --no-cpu-baseline > gpurun_out/abkv.log 2>/dev/null
    python -c "
import json
d=json.loads([l for l in open('gpurun_out/abkv.log') if l.startswith('{')][-1])
print('[$cfg]', d['value'], d['ms_per_step'], d['stages_ms']['detector'], d['p99_latency_ms'])"
  done
done
