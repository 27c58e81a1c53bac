"""HBM-side traffic per kernel from two rocprofv3 PMC passes over tools/pmc_step.py (one with FETCH_SIZE, one with
WRITE_SIZE; see that file for the commands).

    python tools/pmc_report.py <dir or .db of the FETCH_SIZE pass> <... WRITE_SIZE pass> [out.txt] [step.json]

Counter -> bytes.  rocprofv3 reports both counters in KiB-like units that the guide (MI355X_MICROARCH.md, "HBM") says are
NOT bytes-accurate on gfx950 (FETCH_SIZE tallies 128-B requests at 64 B; WRITE_SIZE uncalibrated).  The workload therefore
starts with calibration launches of known traffic (scale_dropout_kernel<float, float> over 2^28 floats: 1 GiB read,
1 GiB written, 4x the Infinity Cache): bytes_per_count = 2^30 / counter value of those launches, applied to every kernel
of the step.  The measured step is the dispatch range between the two marker launches (sum_scale_kernel) that bracket it.
Output: one line per kernel name (time-ordered aggregate): launches, mean duration, counter bytes read / written per step,
achieved HBM-side TB/s = (read + written) / total duration, plus -- for the families listed in ALGO -- the algorithmic
byte count of the same launches (a lower bound: every operand read once, every result written once) and the ratio."""
import json
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kname import clean  # noqa: E402


def load(path, counter):
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
    pmc_ids = [r[0] for r in cur.execute(f"select id from {ip} where name = ?", (counter,))]
    assert pmc_ids, f"{counter} not in {path}"
    vals = {}
    q = f"select event_id, sum(value) from {pe} where pmc_id in ({','.join(map(str, pmc_ids))}) group by event_id"
    for ev, v in cur.execute(q):
        vals[ev] = v
    rows = []
    for name, st, en, ev, gx, gy, gz in cur.execute(
            f"select s.{name_col}, d.start, d.end, d.event_id, d.grid_size_x, d.grid_size_y, d.grid_size_z from {kd} d "
            f"join {ks} s on d.kernel_id = s.id order by d.start"):
        name = clean(name)
        rows.append((name, (en - st) * 1e-3, vals.get(ev, 0.0), gx * gy * gz))
    return rows


# kernel families: (label, regex over rocprofv3 kernel names, C-ABI entry points whose launches they are)
FAMILIES = [
    ("transformer GEMMs (Linear fwd NT on f16 / bf16 / split planes, dgrad NT, wgrad TN, paired launches)",
     r"^(gemm_fast_kernel<\d+, \d+, \d+, 0,|gemm_pair_kernel|gemm_tn_fast_kernel<3, 0>|gemm_split_kernel<\d+, \d+, \d+, 0,)",
     ("avsr_gemm_bf16_nt", "avsr_gemm_bf16_tn", "avsr_gemm_h16_nt", "avsr_gemm_f32s_nt")),
    ("ResNet conv fwd / dgrad (implicit GEMM on f16 / bf16 / split planes + the patch-staged kernels)",
     r"^(gemm_fast_kernel<\d+, \d+, \d+, [12],|conv3x3_c64_kernel|conv_patch_kernel|gemm_split_kernel<\d+, \d+, \d+, 1,)", ("avsr_conv2d_bf16", "avsr_conv2d_h16", "avsr_conv2d_f32s")),
    ("ResNet 3x3 conv wgrad", r"^(conv3x3_wgrad_kernel|wgrad_reduce_kernel)", ("avsr_conv3x3_wgrad_bf16",)),
    ("BatchNorm passes", r"^bn_", ("avsr_bn_stats", "avsr_bn_stats_finalize", "avsr_bn_act_fwd", "avsr_bn_bwd_reduce", "avsr_bn_bwd_apply",
                                  "avsr_bn_act_pool_fwd", "avsr_bn_pool_bwd_reduce", "avsr_bn_pool_bwd_apply", "avsr_bn_small_fwd",
                                  "avsr_bn_small_bwd", "avsr_bn_act_fwd2", "avsr_bn_act_fwd_h16", "avsr_bn_small_fwd2", "avsr_bn_small_fwd_h16")),
    ("LayerNorm fwd / bwd", r"^layernorm_", ("avsr_layernorm_fwd", "avsr_layernorm_bwd", "avsr_layernorm_fwd2", "avsr_layernorm_fwd_h16")),
    ("optimizer (clip + AdamW + bf16 weight copies)", r"^(multi_adamw|multi_sumsq|clip_coef)", ("avsr_adamw_step", "avsr_adamw_cast_step")),
    ("video stem conv fwd / wgrad", r"^stem_", ("avsr_stem357_fwd", "avsr_stem357_wgrad", "avsr_stem357_fwd_f32s")),
    ("depthwise conv fwd / dgrad / wgrad", r"^dwconv_", ("avsr_dwconv_fwd", "avsr_dwconv_wgrad", "avsr_dwconv_fwd2", "avsr_dwconv_fwd_h16")),
    ("max-pool fwd / bwd", r"^maxpool_", ("avsr_maxpool2d_fwd", "avsr_maxpool2d_bwd")),
]


def split(rows):
    """(calibration rows, step rows)."""
    big = [i for i, r in enumerate(rows) if r[0].startswith("scale_dropout_kernel<float, float>") and r[3] >= (1 << 20)]
    assert len(big) >= 3, "calibration launches not found"
    marks = [i for i, r in enumerate(rows) if r[0].startswith("sum_scale_kernel") and i > big[-1]]
    assert len(marks) >= 2, "step markers not found"
    return [rows[i] for i in big[-3:]], rows[marks[0] + 1: marks[-1]]


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out_path = sys.argv[3] if len(sys.argv) > 3 else None
    step_info = json.load(open(sys.argv[4])) if len(sys.argv) > 4 else {}
    cal_f, step_f = split(fetch)
    cal_w, step_w = split(write)
    GiB = float(1 << 30)
    kf = GiB / (sum(r[2] for r in cal_f) / len(cal_f))
    kw = GiB / (sum(r[2] for r in cal_w) / len(cal_w))
    cal_us = sum(r[1] for r in cal_f) / len(cal_f)
    assert [r[0] for r in step_f] == [r[0] for r in step_w], "the two passes did not run the same launch sequence"
    agg, order = {}, []
    for (name, us_f, cf, _), (_, us_w, cw, _) in zip(step_f, step_w):
        a = agg.get(name)
        if a is None:
            a = agg[name] = [0, 0.0, 0.0, 0.0]
            order.append(name)
        a[0] += 1
        a[1] += 0.5 * (us_f + us_w)
        a[2] += cf * kf
        a[3] += cw * kw
    algo = step_info.get("algo_bytes", {})
    tot_us = sum(a[1] for a in agg.values())
    tot_rd, tot_wr = sum(a[2] for a in agg.values()), sum(a[3] for a in agg.values())
    L = []
    L.append(f"HBM-side traffic of ONE eager training step ({step_info.get('shape', 'bench workload, middle batch')}); "
             f"{sum(a[0] for a in agg.values())} kernel launches, {tot_us / 1e3:.2f} ms of kernel time")
    L.append(f"calibration: 1 GiB streamed read+write in {cal_us:.1f} us = {2 * GiB / cal_us / 1e6:.2f} TB/s; "
             f"FETCH_SIZE count = {kf:.1f} B (guide: nominal 1024 B, x2 under-report on gfx950), WRITE_SIZE count = {kw:.1f} B")
    L.append(f"step total: read {tot_rd / 1e9:.2f} GB, written {tot_wr / 1e9:.2f} GB -> {(tot_rd + tot_wr) / tot_us / 1e6:.2f} TB/s "
             f"averaged over kernel time (HBM peak 8 TB/s)")
    entries = step_info.get("entries", {})
    if entries:
        L.append("")
        L.append("per kernel family: algorithmic bytes (every operand read once + every result written once, from the binding layer) "
                 "vs counter bytes")
        L.append(f"{'launches':>8} {'tot_ms':>7} {'algo_MB':>9} {'counter_MB':>10} {'cnt/algo':>8} {'algo TB/s':>9} {'cnt TB/s':>8} {'TFLOP/s':>8}  family")
        L.append("-" * 140)
        for label, pat, ents in FAMILIES:
            ks = [n for n in agg if re.search(pat, n)]
            if not ks:
                continue
            n = sum(agg[k][0] for k in ks)
            us = sum(agg[k][1] for k in ks)
            cnt = sum(agg[k][2] + agg[k][3] for k in ks)
            al = sum(entries[e]["bytes"] for e in ents if e in entries)
            fl = sum(entries[e]["flops"] for e in ents if e in entries)
            L.append(f"{n:8d} {us / 1e3:7.3f} {al / 1e6:9.1f} {cnt / 1e6:10.1f} {(cnt / al if al else float('nan')):8.2f} "
                     f"{al / us / 1e6:9.2f} {cnt / us / 1e6:8.2f} {fl / us / 1e6:8.1f}  {label}")
    L.append("")
    L.append(f"{'calls':>5} {'avg_us':>8} {'tot_ms':>7} {'rd_MB':>9} {'wr_MB':>9} {'TB/s':>6} {'algo_MB':>9} {'cnt/algo':>8}  kernel")
    L.append("-" * 140)
    for name in sorted(order, key=lambda n: -agg[n][1]):
        n, us, rd, wr = agg[name]
        al = algo.get(name)
        L.append(f"{n:5d} {us / n:8.2f} {us / 1e3:7.3f} {rd / 1e6:9.1f} {wr / 1e6:9.1f} {(rd + wr) / us / 1e6:6.2f} "
                 f"{(al / 1e6 if al else float('nan')):9.1f} {((rd + wr) / al if al else float('nan')):8.2f}  {name[:110]}")
    text = "\n".join(L)
    print(text)
    if out_path:
        open(out_path, "w").write(text + "\n")
        summary = {"shape": step_info.get("shape", "bench workload, middle batch"), "launches": sum(a[0] for a in agg.values()),
                   "kernel_time_ms": tot_us / 1e3, "read_bytes": tot_rd, "written_bytes": tot_wr,
                   "calibration": {"bytes_per_FETCH_SIZE_count": kf, "bytes_per_WRITE_SIZE_count": kw,
                                   "stream_TBps": 2 * GiB / cal_us / 1e6},
                   "kernels": {n: {"calls": a[0], "avg_us": a[1] / a[0], "rd_bytes": a[2], "wr_bytes": a[3]}
                               for n, a in agg.items()}}
        json.dump(summary, open(os.path.splitext(out_path)[0] + ".json", "w"), indent=1)
    return agg, kf, kw


if __name__ == "__main__":
    main()
