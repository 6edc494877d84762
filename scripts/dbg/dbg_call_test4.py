import sys, os, inspect
sys.path.insert(0, '.')
import tests.test_gpu_train as T
import torch
from mvster_amd import ops
src = inspect.getsource(T.test_graphed_train_step_follows_the_eager_trajectory)
b = src.index("    # new inputs go through the static buffers")
src = src[:b] + '''    from mvster_amd import ops as _o
    held = list(_o.DBG[-4:])
    def show(tag):
        torch.cuda.synchronize()
        print(tag, [t[-4:].tolist() for t in held], flush=True)
    show("before")
    before = step().item()
    show("after 4th")
''' + src[b:]
ns = dict(T.__dict__)
exec(src, ns)
ns["test_graphed_train_step_follows_the_eager_trajectory"]()
torch.cuda.synchronize()
print("ok")
