#!/usr/bin/env python3
"""A few training steps of BASELINE config 4 on one rank (for profiling): python train_steps.py [H W N B steps native]."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gpu_configs import train  # noqa: E402

graph, coherent = "--graph" in sys.argv, "--coherent" in sys.argv
a = [int(v) for v in sys.argv[1:] if not v.startswith("--")]
H, W, N, B, steps = (a + [512, 640, 5, 2, 3][len(a):])[:5]
train(H, W, N, B, steps=steps, graph=graph, coherent=coherent)
