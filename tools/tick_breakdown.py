#!/usr/bin/env python3
"""Per-tick kernel breakdown of the timed region of a `rocprofv3 --kernel-trace` run of bench.py.

The timed region is found from the K1 launches (one per tick): the last `ticks` of them.  Prints, per kernel name, launches
per tick, average duration and microseconds per tick; plus wall time per tick and the GPU-busy fraction (union of kernel
intervals over all streams / wall)."""
import csv, glob, sys
src, ticks = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 90
f = sorted(glob.glob(f"{src}/**/*kernel_trace.csv", recursive=True))[0]
rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: r[1])
k1 = [r for r in rows if "k1_ratio" in r[0]]
t0 = k1[-ticks - 1][1]
t1 = k1[-1][1]
sel = [r for r in rows if t0 <= r[1] < t1]
wall = (t1 - t0) / ticks / 1e3
iv = sorted((s, e) for _, s, e in sel)
busy, cur_s, cur_e = 0, *iv[0]
for s, e in iv[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
agg = {}
for n, s, e in sel:
    a = agg.setdefault(n, [0, 0])
    a[0] += 1; a[1] += e - s
print(f"# timed region of the kernel trace: {ticks} ticks; wall per tick {wall:.1f} us; GPU busy (union of kernel intervals) {busy / (t1 - t0):.3f}")
print("# sum of kernel durations per tick %.1f us (exceeds wall where kernels of different streams overlap)" % (sum(a[1] for a in agg.values()) / ticks / 1e3))
print("kernel,launches_per_tick,avg_us,us_per_tick")
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"\"{n[:100]}\",{c / ticks:.2f},{d / c / 1e3:.2f},{d / ticks / 1e3:.1f}")
