"""Print PMC counters per kernel dispatch from a rocprofv3 rocpd database."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
kd, ks, pe, ip = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
print([r[1] for r in cur.execute(f"pragma table_info({pe})")]); print([r[1] for r in cur.execute(f"pragma table_info({ip})")])
rows = list(cur.execute(f"select d.id, s.kernel_name, d.start, d.end, d.grid_size_x*d.grid_size_y*d.grid_size_z from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
names = dict(cur.execute(f"select id, name from {ip}"))
vals = {}
for ev, pid, v in cur.execute(f"select event_id, pmc_id, value from {pe}"):
    vals.setdefault(ev, {})
    vals[ev][names[pid]] = vals[ev].get(names[pid], 0) + v
# event_id -> dispatch mapping
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
evcol = "event_id" if "event_id" in cols else "id"
ev_of = dict(cur.execute(f"select id, {evcol} from {kd}"))
for did, name, st, en, grid in rows:
    if "gemm" not in name and "wgrad" not in name: continue
    v = vals.get(ev_of[did], {})
    print(re.sub(r"\(.*", "", name)[:60], f"{(en-st)/1e3:.1f}us", {k: int(x) for k, x in sorted(v.items())})
