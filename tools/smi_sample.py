#!/usr/bin/env python
"""Sample socket power / shader clock / temperature of GPU 0 while a command runs (GPU box only).

    python tools/smi_sample.py <out.csv> -- <command ...>

sysfs hwmon (power1_average | power1_input in uW, freq1_input in Hz, temp*_input in mC) at ~20 Hz when the files exist,
`rocm-smi -P -c -t --json` once a second otherwise (and always once at start, kept as <out>.smi.json for the field names).
Prints min / median / max of power and clock over the samples taken while the GPU was busy (power above the idle third)."""
import glob
import json
import os
import statistics
import subprocess
import sys
import threading
import time


def hwmon_files():
    out = {}
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        for key, names in (("power_uW", ("power1_average", "power1_input")), ("sclk_Hz", ("freq1_input",)),
                           ("temp_mC", ("temp2_input", "temp1_input")), ("mclk_Hz", ("freq2_input",))):
            for n in names:
                p = os.path.join(d, n)
                if key not in out and os.path.exists(p):
                    out[key] = p
        if out:
            break
    return out


def smi_json():
    try:
        r = subprocess.run(["rocm-smi", "-P", "-c", "-t", "--json"], capture_output=True, text=True, timeout=20)
        return json.loads(r.stdout)
    except Exception as e:                                   # noqa: BLE001
        return {"error": str(e)}


def main():
    out = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    files = hwmon_files()
    first = smi_json()
    with open(out + ".smi.json", "w") as f:
        json.dump({"hwmon": files, "rocm_smi": first}, f, indent=1)
    rows, stop = [], threading.Event()

    def sample():
        t0 = time.time()
        while not stop.is_set():
            row = {"t": time.time() - t0}
            if "power_uW" in files:
                for k, p in files.items():
                    try:
                        row[k] = float(open(p).read().strip())
                    except (OSError, ValueError):
                        pass
                rows.append(row)
                time.sleep(0.05)
            else:
                j = smi_json()
                card = next((v for k, v in j.items() if k.startswith("card")), {})
                for k, v in card.items():
                    kl = k.lower()
                    try:
                        if "power" in kl and "(w)" in kl:
                            row["power_uW"] = float(v) * 1e6
                        elif kl.startswith("sclk clock speed"):
                            row["sclk_Hz"] = float(str(v).strip("()Mhz ")) * 1e6
                        elif "temperature" in kl and "junction" in kl:
                            row["temp_mC"] = float(v) * 1e3
                    except ValueError:
                        pass
                rows.append(row)
                time.sleep(0.5)

    th = threading.Thread(target=sample, daemon=True)
    th.start()
    rc = subprocess.call(cmd)
    stop.set()
    th.join(5)
    keys = ["t", "power_uW", "sclk_Hz", "mclk_Hz", "temp_mC"]
    with open(out, "w") as f:
        f.write(",".join(keys) + "\n")
        for r in rows:
            f.write(",".join("%g" % r[k] if k in r else "" for k in keys) + "\n")
    pw = [r["power_uW"] / 1e6 for r in rows if "power_uW" in r]
    if pw:
        thr = min(pw) + (max(pw) - min(pw)) / 3.0
        busy = [r for r in rows if r.get("power_uW", 0) / 1e6 >= thr]
        bp = [r["power_uW"] / 1e6 for r in busy]
        bc = [r["sclk_Hz"] / 1e6 for r in busy if "sclk_Hz" in r]
        line = "smi %s: %d samples (%d busy); power W min/med/max %.0f / %.0f / %.0f" % (
            os.path.basename(out), len(rows), len(busy), min(bp), statistics.median(bp), max(bp))
        if bc:
            line += "; sclk MHz min/med/max %.0f / %.0f / %.0f" % (min(bc), statistics.median(bc), max(bc))
        print(line, flush=True)
    else:
        print("smi %s: no power samples (see %s.smi.json)" % (os.path.basename(out), out), flush=True)
    sys.exit(rc)


if __name__ == "__main__":
    main()
