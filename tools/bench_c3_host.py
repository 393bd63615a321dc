#!/usr/bin/env python
"""bench.py's C3_host leg alone (fits from pinned host arrays, float64 and float32 stamps):
    python tools/bench_c3_host.py [nstamps] [steps]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
print(json.dumps(bench.run_c3_host(torch.device("cuda", 0), n=n, steps=steps), indent=1))
