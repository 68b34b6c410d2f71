"""Derived table for tools/pmc_ab.sh's summary.txt (per-launch averages of the SQ passes): MFMA utilisation, wait shares and the
instruction mix per MFMA of every kernel that issues MFMAs.

    python tools/pmc_sq_derive.py gpurun_out/<session>/sq/summary.txt "<header note>" > profiles/rN_pmc_sq_hot_path.txt

SQ_BUSY_CYCLES counts 32 SQ instances (8 XCDs x 4 shader engines), SQ_VALU_MFMA_BUSY_CYCLES 1024 SIMDs:
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES)."""
import re
import sys


def main():
    txt = open(sys.argv[1]).read()
    note = sys.argv[2] if len(sys.argv) > 2 else ""
    sec, cur = {}, None
    for ln in txt.splitlines():
        if ln.startswith("== "):
            cur = ln[3:].split("/")[0]
            sec[cur] = {}
            continue
        m = re.match(r"\s+(.*?)\s+n=(\d+)\s+(.*)", ln)
        if m and cur:
            d = {k: float(v) for k, v in re.findall(r"(\w+)=([\d.e+\-]+)", m.group(3))}
            sec[cur][m.group(1).strip()] = d
    out = ["# " + note,
           "# tools/pmc_ab.sh = separate rocprofv3 --pmc passes (--kernel-trace only) over `python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-alt",
           "# --no-one-call` at the bench's 32768-ray tile (sqf: --mode fwd); counters are per-launch averages.  Derived by tools/pmc_sq_derive.py:",
           "#   mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES)   wait_any = SQ_WAIT_ANY / SQ_WAVE_CYCLES   issue_wait = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES",
           "# %-52s %9s %9s %10s %9s %8s %9s %9s" % ("kernel (training step)", "mfma_busy", "wait_any", "issue_wait", "valu/mfma", "lds/mfma", "vmrd/mfma", "vmwr/mfma")]

    def rows(s1, s2):
        r = []
        for k, d in sec.get(s1, {}).items():
            if d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0:
                continue
            e = sec.get(s2, {}).get(k, {})
            mf = e.get("SQ_INSTS_MFMA", 0)
            f = lambda x: ("%9.2f" % (e.get(x, 0) / mf)) if mf else "        -"
            r.append("# %-52s %9.3f %9.3f %10.3f %s %s %s %s" % (k[-52:], d["SQ_VALU_MFMA_BUSY_CYCLES"] / (32 * d["SQ_BUSY_CYCLES"]),
                                                                d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"],
                                                                f("SQ_INSTS_VALU"), f("SQ_INSTS_LDS"), f("SQ_INSTS_VMEM_RD"), f("SQ_INSTS_VMEM_WR")))
        return r
    out += rows("sq", "sq2")
    out.append("# inference forward (--mode fwd):")
    out += rows("sqf", "none")
    print("\n".join(out))
    print(txt, end="")


if __name__ == "__main__":
    main()
