#!/usr/bin/env python
"""Sample socket power / shader clock / temperature of GPU 0 while a command runs (GPU box only).

    python tools/smi_sample.py <out.csv> -- <command ...>

sysfs hwmon (power1_average | power1_input in uW, freq1_input in Hz, temp*_input in mC) at ~20 Hz of the card whose PCI
address is HIP device 0's (hipDeviceGetPCIBusId); without a match every card is sampled and the one with the highest peak
power is kept.  `rocm-smi -P -c -t --json` once at start, kept in <out>.smi.json with the hwmon paths.
Prints min / median / max of power and clock over the samples taken while the GPU was busy (power above the idle third)."""
import glob
import json
import os
import statistics
import subprocess
import sys
import threading
import time


def own_pci_bus():
    """PCI bus id of HIP device 0 of THIS container (a gpurun box can be one GPU of a larger node: card0 in sysfs is not
    necessarily it -- session 2 of round 4 sampled an idle neighbour)."""
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, 0) == 0:
            return buf.value.decode().lower()
    except OSError:
        pass
    return None


def hwmon_sets():
    """{card: {key: path}} for every amdgpu card with a hwmon directory."""
    out = {}
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        card = d.split("/")[4]
        files = {}
        for key, names in (("power_uW", ("power1_average", "power1_input")), ("sclk_Hz", ("freq1_input",)),
                           ("temp_mC", ("temp2_input", "temp1_input")), ("mclk_Hz", ("freq2_input",))):
            for n in names:
                p = os.path.join(d, n)
                if key not in files and os.path.exists(p):
                    files[key] = p
        if "power_uW" in files:
            try:
                files["pci"] = os.path.basename(os.path.realpath(os.path.join("/sys/class/drm", card, "device"))).lower()
            except OSError:
                files["pci"] = "?"
            out[card] = files
    return out


def hwmon_files():
    """The hwmon files of OUR GPU: the card whose PCI address is HIP device 0's; with no match, every card is sampled and
    the one that drew the most power during the run is reported (main())."""
    sets = hwmon_sets()
    bus = own_pci_bus()
    for card, files in sets.items():
        if bus and files.get("pci") == bus:
            return {k: v for k, v in files.items() if k != "pci"}, sets, card
    return None, sets, None


def smi_json():
    try:
        r = subprocess.run(["rocm-smi", "-P", "-c", "-t", "--json"], capture_output=True, text=True, timeout=20)
        return json.loads(r.stdout)
    except Exception as e:                                   # noqa: BLE001
        return {"error": str(e)}


def main():
    out = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    files, sets, card = hwmon_files()
    first = smi_json()
    with open(out + ".smi.json", "w") as f:
        json.dump({"own_pci_bus": own_pci_bus(), "matched_card": card, "hwmon": sets, "rocm_smi": first}, f, indent=1)
    watch = {card: files} if files else {c: {k: v for k, v in fs.items() if k != "pci"} for c, fs in sets.items()}
    rows, stop = {c: [] for c in watch}, threading.Event()

    def sample():
        t0 = time.time()
        while not stop.is_set():
            for c, fs in watch.items():
                row = {"t": time.time() - t0}
                for k, p in fs.items():
                    try:
                        row[k] = float(open(p).read().strip())
                    except (OSError, ValueError):
                        pass
                rows[c].append(row)
            time.sleep(0.05 if watch else 0.5)

    th = threading.Thread(target=sample, daemon=True)
    th.start()
    rc = subprocess.call(cmd)
    stop.set()
    th.join(5)
    if not watch:
        print("smi %s: no hwmon power file visible (see %s.smi.json)" % (os.path.basename(out), out), flush=True)
        sys.exit(rc)
    # no PCI match: the card that drew the most during the run
    best = max(rows, key=lambda c: max([r.get("power_uW", 0) for r in rows[c]] or [0]))
    rr = rows[best]
    keys = ["t", "power_uW", "sclk_Hz", "mclk_Hz", "temp_mC"]
    with open(out, "w") as f:
        f.write(",".join(keys) + "\n")
        for r in rr:
            f.write(",".join("%g" % r[k] if k in r else "" for k in keys) + "\n")
    pw = [r["power_uW"] / 1e6 for r in rr if "power_uW" in r]
    thr = min(pw) + (max(pw) - min(pw)) / 3.0
    busy = [r for r in rr if r.get("power_uW", 0) / 1e6 >= thr]
    bp = [r["power_uW"] / 1e6 for r in busy]
    bc = [r["sclk_Hz"] / 1e6 for r in busy if "sclk_Hz" in r]
    line = "smi %s [%s%s]: %d samples (%d busy); power W min/med/max %.0f / %.0f / %.0f" % (
        os.path.basename(out), best, "" if files else ", by peak power of %d cards" % len(rows), len(rr), len(busy), min(bp),
        statistics.median(bp), max(bp))
    if bc:
        line += "; sclk MHz min/med/max %.0f / %.0f / %.0f" % (min(bc), statistics.median(bc), max(bc))
    print(line, flush=True)
    sys.exit(rc)


if __name__ == "__main__":
    main()
