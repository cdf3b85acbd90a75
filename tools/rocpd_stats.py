"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a per-kernel table
(`rocprofv3 --kernel-trace --stats` equivalent).  Usage: python tools/rocpd_stats.py results.db [out.md]"""
import re
import sqlite3
import sys


def main(path, out=None, skip_first_frac=0.0):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    kcols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "kernel_name" if "kernel_name" in kcols else "display_name"
    rows = cur.execute(f"select s.{name_col}, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id").fetchall()
    agg = {}
    for name, st, en in rows:
        name = re.sub(r"\(.*", "", name)[:110]
        a = agg.setdefault(name, [0, 0])
        a[0] += 1
        a[1] += en - st
    total = sum(a[1] for a in agg.values())
    lines = ["| kernel | calls | total ms | avg us | % |", "|---|---:|---:|---:|---:|"]
    for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| `{name}` | {n} | {t/1e6:.3f} | {t/n/1e3:.1f} | {100*t/total:.1f} |")
    lines.append(f"| **total** | {sum(a[0] for a in agg.values())} | {total/1e6:.3f} | | 100 |")
    txt = "\n".join(lines)
    # idle time on the device: gaps between the end of one kernel and the start of the next, charged to the
    # kernel that starts late (shows where the host falls behind)
    rows.sort(key=lambda r: r[1])
    gaps = {}
    busy_end = rows[0][2]
    idle = 0
    for name, st, en in rows[1:]:
        if st > busy_end:
            g = st - busy_end
            if g < 50_000_000:          # ignore the multi-ms holes between phases (warm-up, sync points)
                name = re.sub(r"\(.*", "", name)[:110]
                a = gaps.setdefault(name, [0, 0])
                a[0] += 1
                a[1] += g
                idle += g
        busy_end = max(busy_end, en)
    lines2 = ["", f"device idle between kernels (gaps < 50 ms): {idle/1e6:.3f} ms total", "",
              "| kernel that starts after the gap | gaps | idle ms | avg us |", "|---|---:|---:|---:|"]
    for name, (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
        lines2.append(f"| `{name}` | {n} | {t/1e6:.3f} | {t/n/1e3:.1f} |")
    txt = txt + "\n" + "\n".join(lines2)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(*sys.argv[1:3])
