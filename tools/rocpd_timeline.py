"""Timeline of ONE training step from a rocprofv3 rocpd database (--kernel-trace): every kernel between the last two
optimizer updates, with its start offset, duration, the idle gap since the previous kernel ended, and its grid.
Usage: python tools/rocpd_timeline.py <results.db> <out.txt> [steps_back]
The summary at the top gives launches, busy time, idle time and the per-kernel totals of that single step -- the place to
see what a graph replay really executes (framework kernels included) and where the device waits."""
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kname import clean  # noqa: E402


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols_ks = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    cols_kd = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    name_col = "display_name" if "display_name" in cols_ks else "kernel_name"
    gx = "d.grid_size_x" if "grid_size_x" in cols_kd else ("d.grid_x" if "grid_x" in cols_kd else "0")
    wx = "d.workgroup_size_x" if "workgroup_size_x" in cols_kd else "1"
    rows = list(cur.execute(
        f"select s.{name_col}, d.start, d.end, {gx}, {wx} from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    marks = [i for i, r in enumerate(rows) if "multi_adamw" in r[0]]
    assert len(marks) > back + 1, "need at least two optimizer updates in the trace"
    lo, hi = marks[-1 - back] + 1, marks[-back] + 1
    step = rows[lo:hi]
    t0 = step[0][1]
    lines, agg = [], {}
    busy = gaps = 0.0
    prev_end = None
    for i, (name, st, en, g, w) in enumerate(step):
        name = re.sub(r"at::native::", "", clean(name))
        dur = (en - st) * 1e-3
        gap = (st - prev_end) * 1e-3 if prev_end is not None else 0.0
        prev_end = max(en, prev_end or en)
        busy += dur
        gaps += max(gap, 0.0)
        a = agg.setdefault(name[:100], [0, 0.0, 0.0])
        a[0] += 1
        a[1] += dur
        a[2] += max(gap, 0.0)
        lines.append(f"{i:5d} {(st - t0) * 1e-3:10.1f} {dur:8.2f} {gap:7.2f} {int(g) // max(int(w), 1):7d}  {name[:110]}")
    span = (step[-1][2] - t0) * 1e-3
    head = [f"one step: {len(step)} launches, span {span:.1f} us, kernel time {busy:.1f} us, idle between kernels {gaps:.1f} us",
            # under hipGraph replay the profiler's begin timestamp of a dispatch is (about) the end of its predecessor: the listed
            # durations CONTAIN the dependent-launch boundary, which therefore never shows as idle time
            f"(graph replay: every duration includes its launch boundary -- MI355X guide: 1.45 - 1.9 us between dependent kernels, "
            f"i.e. {len(step) * 1.45e-3:.2f} - {len(step) * 1.9e-3:.2f} ms of this step are boundaries that no line below shows)",
            "", f"{'calls':>6} {'total_us':>9} {'avg_us':>8} {'gap_before_us':>13}  kernel", "-" * 100]
    for name, (n, us, gp) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        head.append(f"{n:6d} {us:9.1f} {us / n:8.2f} {gp:13.1f}  {name}")
    head += ["", f"{'#':>5} {'start_us':>10} {'dur_us':>8} {'gap_us':>7} {'blocks':>7}  kernel", "-" * 100]
    out = "\n".join(head + lines)
    open(sys.argv[2], "w").write(out + "\n")
    print("\n".join(head[:40]))


if __name__ == "__main__":
    main()
