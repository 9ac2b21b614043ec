#!/usr/bin/env python3
"""The reference-native shape alone (landscape.yaml: in 6 / out 3, 128x128x3 latents; bench.py's ref128_step leg): for rocprofv3 passes.
    python tools/bench_ref128.py [B] [precision]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
prec = sys.argv[2] if len(sys.argv) > 2 else "f16"
torch.set_grad_enabled(False)
print(json.dumps(bench.ref128_leg(torch.device("cuda:0"), prec, B)), flush=True)
