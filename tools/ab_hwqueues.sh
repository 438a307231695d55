# throughput of the whole pipeline against the number of HIP hardware queues (GPU_MAX_HW_QUEUES; the runtime multiplexes streams onto them)
for rep in 1; do
for q in default 3 6 8; do
if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
python bench.py --steps 400 --warmup 40 --no-extras --no-cpu-baseline > gpurun_out/abk.log 2>/dev/null
python -c "
import json
d=json.loads([l for l in open('gpurun_out/abk.log') if l.startswith('{')][-1])
print('queues=$q', d['value'], d['ms_per_step'], d['p99_latency_ms'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done; done
