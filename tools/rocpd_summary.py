"""Summarise a rocprofv3 rocpd SQLite database (--kernel-trace) into a per-kernel table:
calls, total ms, average us, share.  Usage: python tools/rocpd_summary.py <results.db> [out.txt]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols_ks = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in cols_ks else "kernel_name"
    rows = list(cur.execute(
        f"select s.{name_col}, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    agg = {}
    for name, st, en in rows:
        name = name.replace("(anonymous namespace)::", "").replace("avsr_gemm_impl::", "")
        name = re.sub(r"\((?!.*<).*$", "", name)
        a = agg.setdefault(name, [0, 0.0])
        a[0] += 1
        a[1] += (en - st) * 1e-6
    tot = sum(v[1] for v in agg.values())
    lines = [f"{'calls':>7} {'total_ms':>10} {'avg_us':>9} {'share':>6}  kernel", "-" * 110]
    for name, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{n:7d} {ms:10.3f} {ms / n * 1e3:9.2f} {ms / tot * 100:5.1f}%  {name[:170]}")
    lines.append("-" * 110)
    lines.append(f"{sum(v[0] for v in agg.values()):7d} {tot:10.3f} total kernel time (ms); wall span {(rows[-1][2] - rows[0][1]) * 1e-6:.1f} ms")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
