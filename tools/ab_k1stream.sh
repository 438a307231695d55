for rep in 1 2; do
for m in same normal prio; do
RVA_K1_STREAM=$m python bench.py --steps 400 --warmup 40 --no-extras --no-cpu-baseline > gpurun_out/abk.log 2>/dev/null
python -c "
import json
d=json.loads([l for l in open('gpurun_out/abk.log') if l.startswith('{')][-1])
print('$m', d['value'], d['ms_per_step'], d['p99_latency_ms'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done; done
