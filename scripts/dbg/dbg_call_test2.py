import sys, os
sys.path.insert(0, '.')
import tests.test_gpu_train as T
import torch
import mvster_amd.graph as G
from mvster_amd import ops
mode = sys.argv[1]
if mode == "nocapture":
    orig_init = G.GraphedTrainStep.__init__
    def init(self, *a, **k):
        k["capture"] = False
        orig_init(self, *a, **k)
    G.GraphedTrainStep.__init__ = init
if mode == "window":
    ops.SORTED_SCATTER = False
if mode == "nobn":
    pass
T.test_graphed_train_step_follows_the_eager_trajectory()
torch.cuda.synchronize()
print("ok", mode)
