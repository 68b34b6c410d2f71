"""GPU diagnostic (not collected by pytest): the whole network (hot path -> merge -> upsampler) on B images in one call against the
same images as two calls -- images equal, parameter gradients additive.  B = 12 stacks 3 B + 1 = 37 maps through the upsampler (more
(image, tile) pairs than one round of workgroups: the case the rider-share scratch of the weight-gradient GEMMs was too small for
until round 5).     python tests/diagnostics/network_split.py [B]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from gazenerf_amd import GazeNeRFNetAMD, synth        # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    net = GazeNeRFNetAMD(featmap_size=16, pred_img_size=64, num_sample_coarse=32).to(dev)
    p = {k: v.to(dev) for k, v in synth.synth_problem(16, batch=B, camera="4", seed=3).items()}
    t_rand = synth.synth_jitter(B, 256, 32, seed=1).to(dev)
    keys = ("merge_img_face", "merge_img_eyes", "merge_img")

    def run(lo, hi):
        net.zero_grad(set_to_none=True)
        q = {k: v[lo:hi].contiguous() for k, v in p.items()}
        res = net("train", q["xy"], None, None, q["shape_code"], q["appea_code"], q["gaze"], q["R"], q["T"], q["Kinv"],
                  t_rand=t_rand[lo:hi].contiguous())["coarse_dict"]
        w = torch.linspace(0.5, 1.5, res[keys[0]][0].numel(), device=dev).reshape(res[keys[0]][0].shape)
        loss = sum((res[k] * w).sum() for k in keys) + (res["bg_img"] ** 2).sum() * ((hi - lo) / B)   # bg_img has no batch axis
        loss.backward()
        return {k: res[k].detach().double().cpu() for k in keys}, {n: q_.grad.double().cpu() for n, q_ in net.named_parameters() if q_.grad is not None}

    full = run(0, B)
    a, b = run(0, B // 2), run(B // 2, B)
    rel = lambda x, y: float((x - y).norm() / y.norm().clamp_min(1e-30))
    e_img = max(rel(full[0][k], torch.cat([a[0][k], b[0][k]], 0)) for k in keys)
    worst = max((rel(full[1][n], a[1][n] + b[1][n]), n) for n in full[1])
    print("B = %d (%d stacked maps): images %.2e, worst parameter gradient %.2e (%s) over %d tensors" % (B, 3 * B + 1, e_img, worst[0], worst[1], len(full[1])))
    ok = e_img <= 1e-5 and worst[0] <= 1e-4
    print("ok" if ok else "FAIL")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
