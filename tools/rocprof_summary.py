"""Summarise a rocprofv3 run (rocpd .db or *_kernel_stats.csv) into a small text table for profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_r01/bench_results.db profiles/r01_bench_kernel_stats.txt
"""
import csv
import sqlite3
import sys


def short(name, n=110):
    name = name.replace("void ", "")
    return name if len(name) <= n else name[:n - 3] + "..."


def from_db(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    return [(r[0], int(r[1]), float(r[2]), float(r[3]), float(r[4])) for r in rows]


def from_csv(path):
    out = []
    for r in csv.DictReader(open(path)):
        out.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3,
                    float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    return out


def main(src, dst):
    rows = from_db(src) if src.endswith(".db") else from_csv(src)
    rows.sort(key=lambda r: -r[2])
    with open(dst, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary of {src}\n")
        f.write(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>6}  kernel\n")
        for name, calls, tot, avg, pct in rows[:40]:
            f.write(f"{calls:>6} {tot:>12.1f} {avg:>10.1f} {pct:>6.2f}  {short(name)}\n")
    print(open(dst).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
