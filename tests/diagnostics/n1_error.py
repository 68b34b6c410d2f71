"""GPU: the upsampler's actual distance from the reference fixtures g8 (the tests assert bounds; this prints the values).

    python tests/diagnostics/n1_error.py
"""
import os, sys
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE))); sys.path.insert(0, os.path.dirname(HERE))
import test_upsample as T
dev = torch.device("cuda:0")
for name in ("g8_upsampler_tiny", "g8_upsampler_full"):
    g, cfg, n_blocks, params, x = T._case(name)
    full = name.endswith("full")
    img, dx, dp = T._run_hip(x, params, n_blocks, cfg["min_feat"], dev)
    e = float((T._subsample("out_img", img.cpu(), full) - g["out_img"]).abs().max())
    gx = T._rel_l2(T._subsample("grad_x", dx.cpu(), full), g["grad_x"])
    gw = max(T._rel_l2(T._subsample(k, v.cpu(), full), g["gradw_" + k]) for k, v in dp.items())
    print(name, "img max-abs %.2e  grad_x rel-L2 %.2e  worst weight grad rel-L2 %.2e" % (e, gx, gw))
