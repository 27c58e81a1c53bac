"""Where do the waves of each kernel spend their cycles?  One rocprofv3 pass over tools/pmc_step.py with
  --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL
(guide, "rocprofv3 PMC slots": WAIT_ANY = wave parked on s_waitcnt / barrier; WAIT_INST_ANY = issue stall; the three
are disjoint shares of WAVE_CYCLES).    python tools/pmc_sq.py <dir or .db>[,<dir>...] [out.txt]   (tools/r6_pmc_sq.sh: two passes of four counters)"""
import sqlite3
import sys
import os
import re

import pmc_report as P

NAMES = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_ACTIVE_INST_LDS",
         "SQ_INST_CYCLES_VMEM", "SQ_LDS_UNALIGNED_STALL"]


def main():
    per = {}
    base = None
    dirs = sys.argv[1].split(",")  # (one directory per rocprofv3 pass: the counters are looked up in each)
    for nm in NAMES:
        rows = None
        for d in dirs:
            try:
                rows = P.load(d, nm)
                break
            except (AssertionError, Exception):
                continue
        if rows is None:
            continue
        _, step = P.split(rows)
        if base is None:
            base = step
        per[nm] = step
    agg, order = {}, []
    for i, (name, us, _, _) in enumerate(base):
        a = agg.get(name)
        if a is None:
            a = agg[name] = {"n": 0, "us": 0.0, **{k: 0.0 for k in per}}
            order.append(name)
        a["n"] += 1
        a["us"] += us
        for k, st in per.items():
            a[k] += st[i][2]
    L = [f"{'calls':>5} {'avg_us':>8} {'wait_any%':>9} {'wait_inst%':>10} {'active%':>8} {'lds_conf%':>9} {'lds_act%':>8} {'vmem%':>6} {'unal%':>6}  kernel",
         "-" * 130]
    for name in sorted(order, key=lambda n: -agg[n]["us"])[:40]:
        a = agg[name]
        wc = max(a.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        pct = lambda k: 100.0 * a.get(k, 0.0) / wc  # noqa: E731
        L.append(f"{a['n']:5d} {a['us'] / a['n']:8.2f} {pct('SQ_WAIT_ANY'):9.1f} {pct('SQ_WAIT_INST_ANY'):10.1f} {pct('SQ_ACTIVE_INST_ANY'):8.1f} "
                 f"{pct('SQ_LDS_BANK_CONFLICT'):9.1f} {pct('SQ_ACTIVE_INST_LDS'):8.1f} {pct('SQ_INST_CYCLES_VMEM'):6.1f} {pct('SQ_LDS_UNALIGNED_STALL'):6.1f}  {name[:90]}")
    text = "\n".join(L)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
