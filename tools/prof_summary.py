"""Turns a rocprofv3 rocpd database (rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd) into the per-kernel
text summary kept under profiles/: calls, total, average, min, max duration and share of GPU time."""
import sqlite3
import sys


def main(db, out, title):
    c = sqlite3.connect(db)
    rows = c.execute(
        "SELECT S.display_name, COUNT(*), SUM(K.end - K.start), MIN(K.end - K.start), MAX(K.end - K.start) "
        "FROM rocpd_kernel_dispatch K JOIN rocpd_info_kernel_symbol S ON S.id = K.kernel_id AND S.guid = K.guid "
        "GROUP BY S.display_name ORDER BY 3 DESC").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = [title, "source: rocprofv3 --kernel-trace --stats (rocpd database), durations in microseconds", "",
             "%10s %12s %10s %10s %10s %7s  %s" % ("calls", "total_us", "avg_us", "min_us", "max_us", "pct", "kernel")]
    for name, n, tot, mn, mx in rows:
        short = name.split("(")[0].replace("void ", "")
        lines.append("%10d %12.1f %10.2f %10.2f %10.2f %6.2f%%  %s" % (n, tot / 1e3, tot / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total, short))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "kernel summary")
