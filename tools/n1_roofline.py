"""Per-launch roofline of the upsampler (N1): algorithmic FLOPs and bytes of every launch of one B-image forward+backward
against the fp32-MFMA peak (157.3 TFLOP/s) and the HBM roof (8.0 TB/s spec, 6.29 TB/s measured copy), from the launch list
tools/n1_trace.sh writes (profiles/r5_n1_launches.txt) -- and, when given, the per-launch FETCH_SIZE / WRITE_SIZE of
tools/n1_pmc.sh (profiles/r5_n1_pmc.txt).

    python tools/n1_roofline.py profiles/r5_n1_launches.txt [profiles/r5_n1_pmc.txt] > profiles/r5_n1_roofline.txt

Algorithmic bytes = every input of the launch read once + every output written once (fp32; the one-byte sign maps as bytes);
split-K partial tiles, weight re-layouts and the batched reductions have no algorithmic bytes: their launches are listed as
overhead.  The roles follow the program order of gnr_upsample_fwd / gnr_upsample_bwd (gazenerf_amd/csrc/gnr_upsample.hip) for
the reference's NeuralRenderer (258 -> 129 -> 64 -> 32 channels, 64 x 64 -> 512 x 512: models/neural_renderer.py:57-113).
"""
import re
import sys

PEAK_TF, HBM_SPEC, HBM_COPY = 157.3e12, 8.0e12, 6.29e12


def expected(B, ch=(258, 129, 64, 32), S0=64):
    """[(role, kernel-name prefix(es), flops, bytes)] in launch order (reductions excluded: they float)."""
    nb = len(ch) - 1
    side = [S0 << i for i in range(nb + 1)]
    out = []
    f4 = 4.0
    P0 = side[0] ** 2
    out.append(("fwd rgb0 conv", ("rgb_conv",), 2 * 3 * ch[0] * B * P0, f4 * B * P0 * (ch[0] + 3)))
    out.append(("fwd rgb up 0", ("bilinear_blur",), 0, f4 * 3 * B * P0 * 5))
    out.append(("fwd weight re-layout", ("conv16_pack",), 0, 0))
    for i in range(nb):
        C, Cn, P = ch[i], ch[i + 1], side[i] ** 2
        px = B * P
        c1 = (2 * 2 * C * C * px, f4 * px * (C + 2 * C))
        c2 = (2 * 4 * C * 2 * C * px, f4 * px * (2 * C + C + 4 * C) + px * C)
        if C % 64 == 0:
            out.append(("fwd b%d layer_1+layer_2 (chain)" % i, ("upchain",), c1[0] + c2[0], f4 * px * (C + 2 * C + 4 * C) + px * C))
        else:
            out.append(("fwd b%d layer_1" % i, ("conv16_kernel",), *c1))
            out.append(("fwd b%d layer_2 + shuffle" % i, ("conv16_kernel",), *c2))
        out.append(("fwd b%d feat_layers (blur fused, RGB rider)" % i, ("conv16_kernel", "conv16_blur_lds"), 2 * Cn * C * 4 * px,
                    f4 * 4 * px * (C + Cn + 3 + 3 + (3 if i == nb - 1 else 0))))
        if i < nb - 1:
            out.append(("fwd rgb up %d" % (i + 1), ("bilinear_blur",), 0, f4 * 3 * B * 4 * P * 5))
    Pn = side[nb] ** 2
    out.append(("bwd sigmoid'", ("sigmoid_bwd",), 0, f4 * 3 * B * Pn * 3))
    out.append(("bwd weight re-layout", ("conv16_pack",), 0, 0))
    for i in range(nb - 1, -1, -1):
        C, Cn, P = ch[i], ch[i + 1], side[i] ** 2
        px = B * P
        if i < nb - 1:
            out.append(("bwd rgb up^T %d" % (i + 1), ("blur_bilinear_adj",), 0, f4 * 3 * B * 4 * P * 5))
        has_next = i < nb - 1
        out.append(("bwd b%d RGB branch + lrelu' + blur^T" % i, ("rgb_bwd_blur", "rgb_bwd_fused"), 2 * 2 * 3 * Cn * 4 * px,
                    f4 * 4 * px * (3 + Cn + (Cn if has_next else 0) + Cn)))
        out.append(("bwd b%d dW feat_layers" % i, ("wgrad",), 2 * Cn * C * 4 * px, f4 * 4 * px * (Cn + C)))
        out.append(("bwd b%d du = Wf^T g + un-shuffle" % i, ("conv16_unshuffle",), 2 * C * Cn * 4 * px,
                    f4 * px * (4 * Cn + 4 * C + (C if C % 4 == 0 else 0)) + px * C))
        out.append(("bwd b%d dW layer_2" % i, ("wgrad",), 2 * 4 * C * 2 * C * px, f4 * px * (4 * C + 2 * C)))
        out.append(("bwd b%d dpre1 = W2^T dpre2" % i, ("conv16_kernel",), 2 * 4 * C * 2 * C * px, f4 * px * (4 * C + 2 * C + 2 * C)))
        out.append(("bwd b%d dW layer_1" % i, ("wgrad",), 2 * 2 * C * C * px, f4 * px * (2 * C + C)))
        out.append(("bwd b%d dnet = W1^T dpre1 (+ repeat^T)" % i, ("conv16_kernel",), 2 * 2 * C * C * px,
                    f4 * px * (2 * C + C + (4 * C if C % 4 else C)) + (px * C if C % 4 else 0)))
    out.append(("bwd rgb up^T 0", ("blur_bilinear_adj",), 0, f4 * 3 * B * P0 * 5))
    out.append(("bwd rgb0 conv", ("rgb_bwd_fused",), 2 * 2 * 3 * ch[0] * B * P0, f4 * B * P0 * (3 + ch[0] + 2 * ch[0])))
    out.append(("bwd RGB weight sums", ("rgb_wsum",), 0, 0))
    return out


def parse_launches(path, section="== b7"):
    rows, on, batch = [], False, 7
    for ln in open(path):
        if ln.startswith("== "):
            on = ln.strip() == section
            continue
        if not on:
            continue
        m = re.match(r"N1 B=(\d+)", ln)
        if m:
            batch = int(m.group(1))
            wall = ln.strip()
            continue
        m = re.match(r"(\S.*?)\s+grid\s+(\S+)\s+([\d.]+) us", ln)
        if m:
            rows.append((m.group(1).strip(), float(m.group(3))))
    return batch, wall, rows


def parse_pmc(path):
    """name -> list of (FETCH bytes, WRITE bytes) in launch order, from tools/n1_pmc.sh's table (columns named in its header)."""
    out = []
    hdr = None
    for ln in open(path):
        if ln.startswith("#") or not ln.strip():
            continue
        parts = ln.split()
        if hdr is None and "FETCH" in ln.upper():
            hdr = ln
            continue
        out.append(ln.rstrip("\n"))
    return hdr, out


def main():
    batch, wall, rows = parse_launches(sys.argv[1])
    exp = expected(batch)
    print("# tools/n1_roofline.py %s" % " ".join(sys.argv[1:]))
    print("# %s" % wall)
    print("# per launch: algorithmic GFLOP and MB (inputs once + outputs once), the time both roofs allow (157.3 TFLOP/s fp32 MFMA; 8.0 TB/s HBM3E),")
    print("# measured us, measured / roof;  bound = which roof is the longer one for the launch")
    print("%-46s %-34s %8s %8s %8s %8s %6s %5s" % ("role", "kernel", "GFLOP", "MB", "roof us", "meas us", "frac", "bound"))
    j = 0
    tot = dict(meas=0.0, roof=0.0, roof_copy=0.0, over=0.0, flop=0.0, byte=0.0, mfma_t=0.0, hbm_t=0.0)
    for name, us in rows:
        if "wgrad_reduce" in name or "copyBuffer" in name:
            print("%-46s %-34s %8s %8s %8s %8.1f %6s %5s" % ("overhead: split-K reduction / copy", name[:34], "-", "-", "-", us, "-", "-"))
            tot["over"] += us
            tot["meas"] += us
            continue
        if j >= len(exp):
            print("?? unexpected launch", name)
            continue
        role, prefixes, fl, by = exp[j]
        if not any(name.startswith(p) for p in prefixes):
            print("?? %s: expected one of %s, found %s" % (role, prefixes, name))
        j += 1
        t_m, t_h = fl / PEAK_TF * 1e6, by / HBM_SPEC * 1e6
        roof = max(t_m, t_h)
        tot["meas"] += us
        if roof == 0.0:
            tot["over"] += us
            print("%-46s %-34s %8s %8s %8s %8.1f %6s %5s" % ("overhead: " + role, name[:34], "-", "-", "-", us, "-", "-"))
            continue
        tot["roof"] += roof
        tot["roof_copy"] += max(t_m, by / HBM_COPY * 1e6)
        tot["flop"] += fl
        tot["byte"] += by
        print("%-46s %-34s %8.2f %8.1f %8.1f %8.1f %6.2f %5s" % (role, name[:34], fl / 1e9, by / 1e6, roof, us, roof / us, "mfma" if t_m >= t_h else "hbm"))
    print()
    print("sum of measured kernel time %.1f us; of which launches without algorithmic work (re-layouts, reductions) %.1f us" % (tot["meas"], tot["over"]))
    print("sum of per-launch roofs %.1f us (HBM at the 8.0 TB/s spec) / %.1f us (HBM at the 6.29 TB/s a copy reaches): the call runs at %.2f / %.2f of it"
          % (tot["roof"], tot["roof_copy"], tot["roof"] / tot["meas"], tot["roof_copy"] / tot["meas"]))
    print("algorithmic work of the call: %.1f GFLOP, %.2f GB -> %.1f us at the MFMA peak alone, %.1f us at the HBM spec alone"
          % (tot["flop"] / 1e9, tot["byte"] / 1e9, tot["flop"] / PEAK_TF * 1e6, tot["byte"] / HBM_SPEC * 1e6))


if __name__ == "__main__":
    main()
