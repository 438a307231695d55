# same-box A/B of the detect-branch concurrency: serial plan / one lane per level / box and class sub-branches split
for cfg in "serial:RVA_SERIAL_HEADS=1" "levels:RVA_HEAD_SPLIT=0" "split:RVA_HEAD_SPLIT=1" "levels2:RVA_HEAD_SPLIT=0" "split2:RVA_HEAD_SPLIT=1"; do
  name=${cfg%%:*}; var=${cfg#*:}
  env $var python bench.py --steps 400 --warmup 20 --no-extras --no-cpu-baseline > gpurun_out/r2_bcmp_$name.log 2>/dev/null
  python -c "
import json
d=json.loads([l for l in open('gpurun_out/r2_bcmp_$name.log') if l.startswith('{')][-1])
print('$name', d['value'], d['ms_per_step'], d['stages_ms']['detector'], d['p99_latency_ms'])"
done
