"""Per-kernel resource report from the compiler's own metadata (no GPU needed): hipcc -S of every csrc/*.hip for gfx950,
then VGPRs / AGPRs / SGPRs / static LDS / scratch / spills per kernel and the occupancy cliff each one sits at
(512 VGPRs per SIMD lane: <= 128 -> 4 waves per SIMD, <= 168 -> 3, <= 256 -> 2, above -> 1).
Usage: python tools/isa_report.py [out.txt]     (about a minute)
Why: a kernel that silently crosses 256 VGPRs loses its second resident block per CU -- the relative-position attention
backward did exactly that when a register prefetch was added (54 -> 62 us), and only the metadata shows it."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "auto_avsr_amd", "csrc")


def demangle(names):
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            out = subprocess.run([tool], input="\n".join(names) + "\n", capture_output=True, text=True).stdout.strip().split("\n")
            if len(out) == len(names):
                return out
        except Exception:
            pass
    return names


def main():
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for f in sorted(os.listdir(CSRC)):
            if not f.endswith(".hip"):
                continue
            asm = os.path.join(tmp, f + ".s")
            cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", CSRC, "-I", os.path.join(ROOT, "include"),
                   "-ffp-contract=fast", "--cuda-device-only", "-S", "-o", asm, os.path.join(CSRC, f)]
            flags = [ln for ln in open(os.path.join(CSRC, f)) if "AVSR_CXXFLAGS:" in ln]
            if flags:
                cmd[1:1] = flags[0].split("AVSR_CXXFLAGS:")[1].split()
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                print("compile failed:", f, r.stderr[-400:])
                continue
            txt = open(asm).read()
            for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?"
                                 r"\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?"
                                 r"\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", txt, flags=re.S):
                agpr, lds, name, scratch, sgpr, sspill, vgpr, vspill = m.groups()
                rows.append((f, name, int(vgpr), int(agpr), int(sgpr), int(lds), int(scratch), int(vspill), int(sspill)))
    names = demangle([r[1] for r in rows])
    lines = [f"{'file':<20} {'vgpr':>5} {'agpr':>5} {'sgpr':>5} {'lds_B':>7} {'scratch':>8} {'v_spill':>7} {'s_spill':>7} {'waves/SIMD':>10}  kernel",
             "(vgpr = unified VGPR + AGPR count; s_spill = SGPRs parked in VGPR lanes, harmless; scratch / v_spill > 0 would be a defect)",
             "-" * 150]
    for (f, _, v, a, s, lds, scr, sp, ssp), nm in sorted(zip(rows, names), key=lambda t: -t[0][2]):
        tot = max(v, 1)  # unified register file: vgpr_count already includes the AGPRs on gfx90a+
        waves = 512 // ((tot + 7) // 8 * 8)
        waves = min(max(waves, 1), 8)
        nm = re.sub(r"\(anonymous namespace\)::", "", nm)
        nm = re.sub(r"\(.*$", "", nm)
        flag = "  <-- SCRATCH / VGPR SPILLS" if (scr or sp) else ""
        lines.append(f"{f:<20} {v:5d} {a:5d} {s:5d} {lds:7d} {scr:8d} {sp:7d} {ssp:7d} {waves:10d}  {nm[:110]}{flag}")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 1:
        open(sys.argv[1], "w").write(out + "\n")


if __name__ == "__main__":
    main()
