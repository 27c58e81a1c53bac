"""List kernel dispatches (in launch order) whose name contains a substring: duration in us."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor(); pat = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
kd, ks = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")
out = []
for name, st, en in cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"):
    if pat in name: out.append(round((en - st) / 1e3, 1))
print(out)
