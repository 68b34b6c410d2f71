"""GPU: the upsampler (N1) alone, forward and backward, at batch B -- the workload of tools/n1_trace.sh, which runs it
under `rocprofv3 --kernel-trace` and lists every launch of the LAST iteration in launch order with its duration.
Also prints wall times per call measured with HIP events (forward alone, forward+backward).

    python tools/n1_trace.py --batch 7 --iters 5
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gazenerf_amd as G                                         # noqa: E402
from gazenerf_amd.upsample import NeuralRendererAMD              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=7)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--fwd-only", action="store_true")
    ap.add_argument("--split", type=int, default=0, help="experiment: the batch as this many groups of images, each on its own stream")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    tile = os.environ.get("N1_CONV16_TILE")                     # tools/n1_sweep.sh: pin one GEMM instance ("MT,NT")
    if tile:
        from gazenerf_amd import _lib
        _lib.check(_lib.load().gnr_set_conv16_tile(*[int(v) for v in tile.split(",")]))
    net = NeuralRendererAMD(graph_inference=os.environ.get("N1_GRAPH", "1") == "1").to(dev)
    x = torch.randn(a.batch, 258, 64, 64, device=dev, requires_grad=not a.fwd_only)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]

    streams = [torch.cuda.Stream() for _ in range(a.split)]
    parts = list(torch.chunk(x.detach(), a.split)) if a.split else []
    for q in parts:
        q.requires_grad_(not a.fwd_only)

    def step_split(mark=False):
        cur = torch.cuda.current_stream()
        if mark: ev[0].record()
        ys = []
        for sidx, q in zip(streams, parts):
            sidx.wait_stream(cur)
            with torch.cuda.stream(sidx):
                if a.fwd_only:
                    with torch.no_grad():
                        ys.append(net(q))
                else:
                    ys.append(net(q))
        for sidx in streams:
            cur.wait_stream(sidx)
        if mark: ev[1].record()
        if not a.fwd_only:
            for sidx, y in zip(streams, ys):
                sidx.wait_stream(cur)
                with torch.cuda.stream(sidx):
                    y.backward(torch.ones_like(y))
            for sidx in streams:
                cur.wait_stream(sidx)
        if mark: ev[2].record()

    def step(mark=False):
        if a.split:
            return step_split(mark)
        if a.fwd_only:
            with torch.no_grad():
                if mark: ev[0].record()
                y = net(x)
                if mark: ev[1].record(); ev[2].record()
            return
        if mark: ev[0].record()
        y = net(x)
        if mark: ev[1].record()
        y.backward(torch.ones_like(y))
        if mark: ev[2].record()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    tf, tb = [], []
    for _ in range(a.iters):
        step(True)
        torch.cuda.synchronize()
        tf.append(ev[0].elapsed_time(ev[1])); tb.append(ev[0].elapsed_time(ev[2]))
    tf.sort(); tb.sort()
    print("N1 B=%d: forward %.3f ms, forward+backward %.3f ms (median of %d)" % (a.batch, tf[len(tf) // 2], tb[len(tb) // 2], a.iters))


if __name__ == "__main__":
    main()
