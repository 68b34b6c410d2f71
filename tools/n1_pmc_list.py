"""Per-dispatch counters of the GEMM launches of the last iteration of tools/n1_trace.py (rocprofv3 --pmc CSVs of
tools/n1_pmc.sh: sq/, fetch/, write/).  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs);
FETCH_SIZE / WRITE_SIZE are KB: bytes = counter x 1024 (MI355X_MICROARCH.md, HBM section); FETCH_SIZE is printed as
counted -- on gfx950 it tallies the 128-byte requests of 16-byte-per-lane streaming reads at 64 bytes (double it for
those), and Infinity-Cache hits are included."""
import collections
import csv
import glob
import os
import sys


def load(d):
    fs = sorted(glob.glob(os.path.join(d, "*counter_collection.csv")) + glob.glob(os.path.join(d, "*", "*counter_collection.csv")))
    if not fs:
        return None
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        e = disp.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "grid": r.get("Grid_Size", "?")})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return disp


def last_iter(disp):
    ids = sorted(disp)
    names = [disp[i]["name"] for i in ids]
    rgb = [k for k, n in enumerate(names) if "rgb_conv" in n]        # rgb_conv_kernel / rgb_conv_split_kernel: first launch of a forward
    grids = {k: int(disp[ids[k]]["grid"]) for k in rgb}
    start = [k for k in rgb if grids[k] == min(grids.values())][-1] if rgb else 0
    return [disp[ids[k]] for k in range(start, len(ids))]


root = sys.argv[1]
sq, fe, wr = (load(os.path.join(root, x)) for x in ("sq", "fetch", "write"))
rows = last_iter(sq)
fr = last_iter(fe) if fe else [None] * len(rows)
wrr = last_iter(wr) if wr else [None] * len(rows)
for d, f, w in zip(rows, fr, wrr):
    if "at::native" in d["name"] or "rocclr" in d["name"]:
        continue
    n = d["name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("gnr::", "").split("(")[0]
    wc = d.get("SQ_WAVE_CYCLES", 0)
    gui = d.get("GRBM_GUI_ACTIVE", 0)
    print("%-34s grid %-9s waves %-6d us@2.4GHz %-7.1f mfma_busy %.3f  wait_any %.3f  wait_inst %.3f  fetch %8.1f MB  write %8.1f MB" % (
        n[:34], d["grid"], d.get("SQ_WAVES", 0), gui / 8 / 2400.0, d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(gui * 128.0, 1),
        d.get("SQ_WAIT_ANY", 0) / max(wc, 1), d.get("SQ_WAIT_INST_ANY", 0) / max(wc, 1),
        (f or {}).get("FETCH_SIZE", 0) * 1024 / 1e6, (w or {}).get("WRITE_SIZE", 0) * 1024 / 1e6))
