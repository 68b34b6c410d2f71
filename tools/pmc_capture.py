#!/usr/bin/env python
"""HBM-side traffic of the bench's kernels from rocprofv3 PMC counters -> profiles/pmc_traffic.json.

Runs ON THE GPU BOX (via gpurun):

    python tools/pmc_capture.py <outdir-under-gpurun_out> [--mode fwdbwd|fwd] [bench args...]

Collection follows MI355X_MICROARCH.md (HBM / rocprofv3 sections): FETCH_SIZE and WRITE_SIZE in SEPARATE
`rocprofv3 --pmc <counter> --kernel-trace` passes (they do not fit one pass; no other trace domains), bytes =
counter [KB] * 1024, FETCH_SIZE doubled for 16-byte-per-lane coalesced streaming reads (this build's loads are all
dwordx4 / LDS-DMA dwordx4).  WRITE_SIZE is uncalibrated on gfx950; it is reported as counted and checked against the
known dump sizes in DESIGN.md.

The JSON is keyed by (kernel, rays per launch, samples per ray): bench.py reports `roofline.traffic` only from an
entry captured at its own launch size and refuses anything else.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_pass(counter, outdir, bench_args):
    d = os.path.join(outdir, counter.lower())
    os.makedirs(d, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--",
           sys.executable, os.path.join(ROOT, "bench.py")] + bench_args + ["--no-cpu-baseline", "--no-alt", "--no-one-call"]
    with open(os.path.join(d, "bench.log"), "w") as log:
        subprocess.run(["timeout", "600"] + cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, check=False)   # a crashed rocprofv3 can hang forever
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    out_name = sys.argv[1]
    args = sys.argv[2:]
    mode = "fwdbwd"
    if "--mode" in args:
        mode = args[args.index("--mode") + 1]
    side = int(args[args.index("--side") + 1]) if "--side" in args else 512
    samples = int(args[args.index("--samples") + 1]) if "--samples" in args else 64
    micro = int(args[args.index("--micro") + 1]) if "--micro" in args else 32768          # bench.py's default tile
    rays = min(micro, side * side) if mode == "fwdbwd" else side * side
    outdir = os.path.join(ROOT, "gpurun_out", out_name)
    os.makedirs(outdir, exist_ok=True)
    bench_args = args + (["--steps", "1", "--warmup", "0"] if "--steps" not in args else [])
    fetch = run_pass("FETCH_SIZE", outdir, bench_args)
    write = run_pass("WRITE_SIZE", outdir, bench_args)
    entries = []
    for name in sorted(set(fetch) | set(write)):
        if "gnr::" not in name:
            continue
        short = name.split("(")[0].replace("void ", "")
        f_kb = sum(fetch.get(name, [0.0])) / max(1, len(fetch.get(name, [])))
        w_kb = sum(write.get(name, [0.0])) / max(1, len(write.get(name, [])))
        entries.append({
            "kernel": short, "rays_per_launch": rays, "samples_per_ray": samples, "dispatches": len(fetch.get(name, [])),
            "FETCH_SIZE_KB_mean": f_kb, "WRITE_SIZE_KB_mean": w_kb,
            "hbm_bytes_per_launch": 2.0 * f_kb * 1024.0 + w_kb * 1024.0,
            "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) on `bench.py %s`; "
                      "bytes = KB x 1024, FETCH_SIZE x 2 (gfx950 counts 128-B requests of 16-B/lane streaming reads at 64 B), "
                      "mean per dispatch" % " ".join(bench_args)})
    path = os.path.join(outdir, "pmc_traffic_%s.json" % mode)
    with open(path, "w") as f:
        json.dump({"kernels": entries}, f, indent=1)
    for e in entries:
        print("%-60s n=%-4d fetch %.1f MB (x2) write %.1f MB -> %.1f MB per launch" % (
            e["kernel"][:60], e["dispatches"], e["FETCH_SIZE_KB_mean"] / 1024 * 2, e["WRITE_SIZE_KB_mean"] / 1024,
            e["hbm_bytes_per_launch"] / 1e6))
    print("wrote", path)


if __name__ == "__main__":
    main()
