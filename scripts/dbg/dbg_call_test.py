import sys
sys.path.insert(0, '.')
import tests.test_gpu_train as T
import torch
if len(sys.argv) > 1:
    T.note = lambda *a, **k: None
T.test_graphed_train_step_follows_the_eager_trajectory()
torch.cuda.synchronize()
print("ok")
