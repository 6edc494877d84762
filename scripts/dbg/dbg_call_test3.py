import sys, os, inspect, textwrap
sys.path.insert(0, '.')
import tests.test_gpu_train as T
import torch
src = inspect.getsource(T.test_graphed_train_step_follows_the_eager_trajectory)
mode = sys.argv[1]
a = src.index("    moved, worst, worst_name")
b = src.index("    # new inputs go through the static buffers")
if mode == "nostats":
    src = src[:a] + src[b:]
elif mode == "statsonly_m1":
    src = src[:a] + "    for k, pa in m1.named_parameters():\n        x = pa.detach() - sd[k].to(DEV)\n" + src[b:]
elif mode == "sync":
    src = src[:b] + "    torch.cuda.synchronize()\n    import gc; gc.collect()\n    torch.cuda.empty_cache()\n" + src[b:]
elif mode == "nom1":
    c = src.index("    m1, o1 = build()")
    d = src.index("    m2, o2 = build()")
    src = src[:c] + "    eager = [0]*6\n" + src[d:a] + src[b:]
    src = src.replace("    for a, b in zip(eager[3:], graphed):\n        assert abs(a - b) <= 2e-2 * abs(a), (eager, graphed)\n    assert graphed[-1] < eager[0]\n", "")
ns = dict(T.__dict__)
exec(src, ns)
ns["test_graphed_train_step_follows_the_eager_trajectory"]()
torch.cuda.synchronize()
print("ok", mode)
