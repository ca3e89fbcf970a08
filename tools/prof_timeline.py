"""One steady-state step of a rocprofv3 --kernel-trace run (rocpd .db) as a timeline: every kernel between the ends of the last two
launches of a marker kernel (default: amsgrad), with its start offset, duration and queue / stream -- what runs beside what, where the
chip idles, which chain is the critical one.
Usage: python tools/prof_timeline.py results.db out.tsv [marker substring [marker launches to skip at the end]]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    marker = sys.argv[3] if len(sys.argv) > 3 else "amsgrad"
    cols = [r[1] for r in db.execute("PRAGMA table_info(kernels)")]
    lane = [c for c in ("stream_id", "queue_id", "queue", "stream", "tid") if c in cols]
    sel = "name, start, end" + "".join(", " + c for c in lane)
    rows = db.execute("select %s from kernels order by start" % sel).fetchall()
    ends = [r[2] for r in rows if marker in r[0]]
    if len(ends) < 3:
        raise SystemExit("fewer than three %r launches in the trace" % marker)
    skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0       # (e.g. the steps of a second leg timed behind the one of interest)
    if len(ends) < 3 + skip:
        raise SystemExit("fewer than %d %r launches in the trace" % (3 + skip, marker))
    t0, t1 = ends[-3 - skip], ends[-2 - skip]                # (the last step may be followed by teardown work: take the one before)
    step = [r for r in rows if t0 <= r[1] < t1]
    busy, cur_end = 0, t0
    with open(sys.argv[2], "w") as f:
        f.write("# step of %.3f ms, %d kernels; columns: start_us dur_us gap_before_us %s kernel\n" % ((t1 - t0) / 1e6, len(step), " ".join(lane)))
        for r in step:
            gap = max(0, r[1] - cur_end)
            busy += max(0, r[2] - max(r[1], cur_end))
            cur_end = max(cur_end, r[2])
            f.write("%9.1f %8.1f %7.1f %s %s\n" % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, gap / 1e3, " ".join(str(v) for v in r[3:]), r[0][:110]))
        f.write("# chip busy (union of kernel intervals) %.3f ms of %.3f ms; sum of kernel durations %.3f ms\n"
                % (busy / 1e6, (t1 - t0) / 1e6, sum(r[2] - r[1] for r in step) / 1e6))


if __name__ == "__main__":
    main()
