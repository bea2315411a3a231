"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) into a per-kernel stats table (markdown)."""
import sqlite3
import sys


def main(path, out=None, by_calls=None):
    db = sqlite3.connect(path)
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tables if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tables if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
    scols = [r[1] for r in db.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else "name")
    rows = db.execute(f"select s.{name_col}, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), "
                      f"max(d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} "
                      f"order by 3 desc").fetchall()
    total = sum(r[2] for r in rows)
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---:|---:|---:|---:|---:|---:|"]
    for name, calls, tot, avg, mn, mx in rows[:40]:
        short = name if len(name) < 100 else name[:97] + "..."
        lines.append(f"| `{short}` | {calls} | {tot / 1e6:.3f} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100 * tot / total:.1f} |")
    lines.append(f"\ntotal kernel time {total / 1e6:.2f} ms over {sum(r[1] for r in rows)} dispatches ({len(rows)} distinct kernels)")
    text = "\n".join(lines)
    if by_calls:  # every kernel, most launched first: where a step's dispatches are
        with open(by_calls, "w") as f:
            for name, calls, tot, avg, mn, mx in sorted(rows, key=lambda r: -r[1]):
                f.write(f"{calls:7d} {tot / 1e6:10.3f} ms  {name[:160]}\n")
    if out:
        open(out, "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, sys.argv[3] if len(sys.argv) > 3 else None)
