"""Sums rocprofv3 --pmc counters per kernel from a rocpd database and prints derived ratios."""
import sqlite3
import sys
from collections import defaultdict


def load(db):
    c = sqlite3.connect(db)
    rows = c.execute(
        "SELECT S.display_name, P.name, SUM(E.value), COUNT(DISTINCT K.id) FROM rocpd_pmc_event E "
        "JOIN rocpd_info_pmc P ON P.id = E.pmc_id AND P.guid = E.guid "
        "JOIN rocpd_kernel_dispatch K ON K.event_id = E.event_id AND K.guid = E.guid "
        "JOIN rocpd_info_kernel_symbol S ON S.id = K.kernel_id AND S.guid = K.guid GROUP BY 1, 2").fetchall()
    out = defaultdict(dict)
    for k, name, v, n in rows:
        short = k.split("(")[0].replace("void ", "")
        out[short][name] = v
        out[short]["_dispatches"] = n
    return out


def main(db, out_path=None, title=""):
    data = load(db)
    lines = [title, "source: rocprofv3 --pmc (rocpd database); counters summed over all dispatches of each kernel", ""]
    for k, v in sorted(data.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("FETCH_SIZE", 0))):
        lines.append("%s  (%d dispatches)" % (k, v["_dispatches"]))
        for name in sorted(x for x in v if not x.startswith("_")):
            lines.append("    %-28s %.6g" % (name, v[name]))
        if "SQ_THREAD_CYCLES_VALU" in v and v.get("SQ_ACTIVE_INST_VALU"):
            lines.append("    %-28s %.3f   (SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64))" % ("active-lane fraction", v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64.0)))
        if "SQ_WAVE_CYCLES" in v and v.get("SQ_WAVE_CYCLES"):
            wc = v["SQ_WAVE_CYCLES"]
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if n in v:
                    lines.append("    %-28s %.3f" % (n + " / WAVE_CYCLES", v[n] / wc))
        if "SQ_ACTIVE_INST_VALU" in v and v.get("SQ_BUSY_CYCLES"):
            pass
        lines.append("")
    text = "\n".join(lines)
    print(text)
    if out_path:
        open(out_path, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, sys.argv[3] if len(sys.argv) > 3 else "")
