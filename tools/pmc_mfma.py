"""MFMA-pipe utilisation per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES pass over tools/pmc_step.py.

    python tools/pmc_mfma.py <dir or .db> [out.txt]

SQ_VALU_MFMA_BUSY_CYCLES counts, per SIMD, the cycles its MFMA pipe is busy (32 per v_mfma_f32_32x32x16_bf16; guide:
MI355X_MICROARCH.md "Per-instruction cycle constants"); rocprofv3 reports the sum over the 1024 SIMDs.
utilisation = counter / (1024 SIMDs x kernel duration x 2.4 GHz) -- 100 % would be the 2.5 PFLOP/s dense bf16 peak.
Only the measured step (between the two marker launches of pmc_step.py) is summarised."""
import sys

import pmc_report as P


def main():
    rows = P.load(sys.argv[1], "SQ_VALU_MFMA_BUSY_CYCLES")
    _, step = P.split(rows)
    agg, order = {}, []
    for name, us, c, _ in step:
        a = agg.get(name)
        if a is None:
            a = agg[name] = [0, 0.0, 0.0]
            order.append(name)
        a[0] += 1
        a[1] += us
        a[2] += c
    tot_us = sum(a[1] for a in agg.values())
    tot_c = sum(a[2] for a in agg.values())
    L = [f"MFMA-pipe utilisation of ONE eager training step: {sum(a[0] for a in agg.values())} launches, {tot_us / 1e3:.2f} ms of kernel "
         f"time; whole step {tot_c / (1024 * tot_us * 2400) * 100:.1f} % of the dense bf16 MFMA peak",
         "", f"{'calls':>5} {'avg_us':>8} {'tot_ms':>7} {'mfma_busy%':>10}  kernel", "-" * 120]
    for name in sorted(order, key=lambda n: -agg[n][1]):
        n, us, c = agg[name]
        if c <= 0:
            continue
        L.append(f"{n:5d} {us / n:8.2f} {us / 1e3:7.3f} {c / (1024 * us * 2400) * 100:10.1f}  {name[:100]}")
    text = "\n".join(L)
    print(text)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")


if __name__ == "__main__":
    main()
