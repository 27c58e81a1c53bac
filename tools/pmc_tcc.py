"""L2 (TCC) counters per kernel family from rocprofv3 --pmc passes over tools/pmc_step.py: hit rate of the per-XCD L2 and the
requests it sent on to the fabric (Infinity Cache / HBM), for the measured step (between the two marker launches).

    python tools/pmc_tcc.py <dir/.db with TCC_HIT_sum + TCC_MISS_sum> [<dir/.db with TCC_EA0_RDREQ_sum (+ TCP_TCC_READ_REQ_sum)>] [out.txt]

Answers VERDICT r3 item 5a: is the 13-18 TB/s operand-delivery ceiling of the M = B*T GEMMs an L2 -> CU limit (high L2 hit rate: the
bytes come out of the 4 MiB L2 of the XCD) or a fabric limit (every XCD missing on the whole problem: low hit rate, EA read
requests ~ 8x the operand bytes)?"""
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kname import clean  # noqa: E402

FAMILIES = [("transformer GEMM fwd (f16 NT tiles)", r"^gemm_fast_kernel<\d+, \d+, \d+, 0, .*, 1>$"),
            ("transformer GEMM (bf16 NT tiles)", r"^gemm_fast_kernel<\d+, \d+, \d+, 0, .*, 0>$|^gemm_fast_kernel<\d+, \d+, \d+, 0, \d, \d, \d, \d>$"),
            ("paired dgrad + wgrad GEMMs", r"^gemm_pair_kernel"), ("TN wgrad GEMM", r"^gemm_tn_fast_kernel"),
            ("split-plane GEMM / conv (precise forward)", r"^gemm_split_kernel"),
            ("trunk conv fwd / dgrad (tiled)", r"^gemm_fast_kernel<\d+, \d+, \d+, [12],"), ("trunk conv fwd / dgrad (patch-staged)", r"^conv_patch_kernel"), ("conv3x3 c64", r"^conv3x3_c64_kernel"),
            ("3x3 wgrad", r"^conv3x3_wgrad_kernel|^wgrad_reduce"), ("video stem", r"^stem_"), ("attention", r"^attn_"), ("BatchNorm", r"^bn_"),
            ("LayerNorm", r"^layernorm"), ("depthwise conv", r"^dwconv"), ("optimizer", r"^multi_adamw|^multi_sumsq|^clip_coef")]


def load(path):
    if os.path.isdir(path):
        hits = [os.path.join(r, f) for r, _, fs in os.walk(path) for f in fs if f.endswith(".db")]
        assert hits, f"no .db under {path}"
        path = hits[0]
    cur = sqlite3.connect(path).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tabs if t.startswith(p)][0]  # noqa: E731
    kd, ks, pe, ip = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol"), T("rocpd_pmc_event"), T("rocpd_info_pmc")
    cols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in cols else "kernel_name"
    names = dict(cur.execute(f"select id, name from {ip}"))
    vals = {}
    for ev, pid, v in cur.execute(f"select event_id, pmc_id, value from {pe}"):
        d = vals.setdefault(ev, {})
        d[names[pid]] = d.get(names[pid], 0) + v
    rows = []
    for name, st, en, ev, gx in cur.execute(f"select s.{name_col}, d.start, d.end, d.event_id, d.grid_size_x from {kd} d "
                                            f"join {ks} s on d.kernel_id = s.id order by d.start"):
        rows.append((clean(name), st, en, vals.get(ev, {}), gx))
    # the measured step lies between the first marker launch (sum_scale_kernel) after the 1 GiB calibration stream and the last one
    big = [i for i, r in enumerate(rows) if r[0].startswith("scale_dropout_kernel<float, float>") and r[4] >= (1 << 20)]
    assert big, "calibration launches not found"
    marks = [i for i, r in enumerate(rows) if r[0].startswith("sum_scale_kernel") and i > big[-1]]
    assert len(marks) >= 2, "marker launches not found"
    return [r[:4] for r in rows[marks[0] + 1: marks[-1]]], sorted({n for n in names.values()})


def main():
    args = [a for a in sys.argv[1:] if not a.endswith(".txt")]
    out = [a for a in sys.argv[1:] if a.endswith(".txt")]
    agg = {}
    counters = []
    for path in args:
        rows, cs = load(path)
        counters += cs
        for name, st, en, v in rows:
            fam = next((f for f, pat in FAMILIES if re.search(pat, name)), "other")
            a = agg.setdefault(fam, {"calls": {}, "us": {}})
            a["calls"][path] = a["calls"].get(path, 0) + 1
            a["us"][path] = a["us"].get(path, 0.0) + (en - st) / 1e3
            for k, x in v.items():
                a[k] = a.get(k, 0) + x
    lines = [f"# L2 (TCC) counters of ONE eager training step (tools/pmc_step.py), per kernel family; counters: {sorted(set(counters))}",
             f"{'family':44s} {'calls':>6s} {'ms':>7s} {'L2 hit %':>9s} {'L2 req (M)':>11s} {'EA rd req (M)':>14s} {'EA rd GB (x64 B / x128 B)':>26s} {'TCP->TCC rd req (M)':>20s}"]
    for fam, a in sorted(agg.items(), key=lambda kv: -max(kv[1]["us"].values())):
        hit, miss = a.get("TCC_HIT_sum", 0), a.get("TCC_MISS_sum", 0)
        ea = a.get("TCC_EA0_RDREQ_sum", a.get("TCC_EA_RDREQ_sum", 0))
        tcp = a.get("TCP_TCC_READ_REQ_sum", 0)
        lines.append(f"{fam:44s} {max(a['calls'].values()):6d} {max(a['us'].values()) / 1e3:7.2f} "
                     f"{(100.0 * hit / (hit + miss) if hit + miss else float('nan')):9.1f} {(hit + miss) / 1e6:11.1f} {ea / 1e6:14.1f} "
                     f"{ea * 64 / 1e9:12.2f} / {ea * 128 / 1e9:<11.2f} {tcp / 1e6:20.1f}")
    text = "\n".join(lines)
    print(text)
    if out:
        open(out[0], "w").write(text + "\n")


if __name__ == "__main__":
    main()
