#!/usr/bin/env python3
"""Turn a rocprofv3 `--kernel-trace --stats --output-format csv` directory into the short summary
committed under profiles/ (top kernels + every librva kernel, per-launch averages)."""
import csv
import glob
import sys


def main(src, out, note=""):
    f = sorted(glob.glob(f"{src}/**/*kernel_stats.csv", recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    own = ("k1_", "k2_", "k3_", "k4_", "k_zero", "k_bias")
    with open(out, "w") as o:
        o.write(f"# rocprofv3 --kernel-trace --stats summary ({note})\n# source: {f}\n")
        o.write(f"# total kernel time {tot / 1e6:.3f} ms over {sum(int(r['Calls']) for r in rows)} launches\n")
        o.write("name,calls,total_ms,avg_us,min_us,max_us,pct\n")
        for i, r in enumerate(rows):
            if i < 25 or any(k in r["Name"] for k in own):
                o.write(f"\"{r['Name'][:110]}\",{r['Calls']},{float(r['TotalDurationNs']) / 1e6:.3f},"
                        f"{float(r['AverageNs']) / 1e3:.2f},{float(r['MinNs']) / 1e3:.2f},{float(r['MaxNs']) / 1e3:.2f},"
                        f"{float(r['Percentage']):.2f}\n")
    print(open(out).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
