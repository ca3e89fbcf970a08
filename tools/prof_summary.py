"""Kernel-time summary of a rocprofv3 --kernel-trace run (rocpd .db) as CSV: name, calls, total_ms, avg_ms, percent, ms per step
(the top_kernels view reports microseconds).
Usage: python tools/prof_summary.py results.db [out.csv] [steps]"""
import csv
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    out = open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 and sys.argv[2] != "-" else sys.stdout
    w = csv.writer(out)
    w.writerow(["kernel", "calls", "total_ms", "avg_ms", "percent", "ms_per_step"])
    for name, calls, total, avg, pct in rows:
        short = name if len(name) < 200 else name[:197] + "..."
        w.writerow([short, calls, "%.1f" % (total / 1e3), "%.4f" % (avg / 1e3), "%.2f" % pct, "%.3f" % (total / 1e3 / steps)])
    w.writerow(["TOTAL", sum(r[1] for r in rows), "%.1f" % (sum(r[2] for r in rows) / 1e3), "", "100", "%.1f" % (sum(r[2] for r in rows) / 1e3 / steps)])


if __name__ == "__main__":
    main()
