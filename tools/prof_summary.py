"""Turns a rocprofv3 rocpd database (rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd) into the per-kernel
text summary kept under profiles/: calls, total, average, min, max duration and share of GPU time, and the same per
render batch (every k_generate dispatch starts one), so that the batches of the timed steps can be told from the warm-up,
the instrumented pass and the small comparison render that bench.py also makes."""
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
    disp = c.execute(
        "SELECT S.display_name, K.start, K.end - K.start FROM rocpd_kernel_dispatch K JOIN rocpd_info_kernel_symbol S "
        "ON S.id = K.kernel_id AND S.guid = K.guid ORDER BY K.start").fetchall()
    batches = []
    for name, _, dur in disp:
        short = name.split("(")[0].replace("void ", "")
        if short.startswith("rtd::k_generate"):
            batches.append({})
        if not batches or not (short.startswith("rtd::k_extend") or short.startswith("rtd::k_shade")):
            continue
        b = batches[-1].setdefault(short, [0, 0.0])
        b[0] += 1
        b[1] += dur / 1e3
    if batches:
        lines += ["", "per render batch, in time order (a k_generate dispatch starts a batch): launches, total_us, avg_us"]
        for k, b in enumerate(batches):
            for short, (n, tot) in sorted(b.items()):
                lines.append("  batch %2d %6d %12.1f %10.2f  %s" % (k, n, tot, tot / n, short))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "kernel summary")
