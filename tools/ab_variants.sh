# same-box A/B of kernel variants in the whole pipeline: tools/ab_variants.sh "64 65" -> bench with all variants / with these skipped, twice each
skip="$1"
for rep in 1 2; do
  for mode in all skip; do
    if [ $mode = skip ]; then export RVA_SKIP_VARIANTS="$skip"; else unset RVA_SKIP_VARIANTS; fi
    python bench.py --steps 500 --warmup 50 --no-extras --no-cpu-baseline > gpurun_out/abv_$mode.log 2>/dev/null
    python -c "
import json
d=json.loads([l for l in open('gpurun_out/abv_$mode.log') if l.startswith('{')][-1])
print('$mode (skip=$skip)', d['value'], d['ms_per_step'], d['stages_ms']['detector'], d['p99_latency_ms'])"
  done
done
