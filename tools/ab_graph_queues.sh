for q in 4 8; do for g in "" "--no-graph"; do
GPU_MAX_HW_QUEUES=$q python bench.py --steps 400 --warmup 40 --no-extras --no-cpu-baseline $g > gpurun_out/abk.log 2>/dev/null
python -c "
import json
d=json.loads([l for l in open('gpurun_out/abk.log') if l.startswith('{')][-1])
print('queues=$q $g', d['value'], d['ms_per_step'], d['p99_latency_ms'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
done; done
