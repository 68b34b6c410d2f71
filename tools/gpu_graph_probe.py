"""Scratch: can the whole-network inference forward be captured in a HIP graph (torch.cuda.CUDAGraph)?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gazenerf_amd import synth, GazeNeRFNetAMD
dev = torch.device("cuda:0")
for prec in ("fp32", "bf16x3"):
    net = GazeNeRFNetAMD(precision=prec).to(dev).eval()
    p = {k: v.to(dev) for k, v in synth.synth_problem(64, batch=1, seed=1).items()}
    args = ("test", p["xy"], None, None, p["shape_code"], p["appea_code"], p["gaze"], p["R"], p["T"], p["Kinv"])
    with torch.no_grad():
        for _ in range(3):
            ref = net(*args)["coarse_dict"]["merge_img"].clone()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); net(*args); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        eager = sorted(ts)[2]
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            net(*args)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            out = net(*args)["coarse_dict"]["merge_img"]
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print("%s: eager %.2f ms, graph replay %.2f ms, identical: %s" % (prec, eager * 1e3, sorted(ts)[2] * 1e3, bool(torch.equal(out, ref))))
