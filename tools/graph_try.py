import os, sys, time, torch
sys.path.insert(0, "/root/repo")
import unet_convlstm_amd as U
from unet_convlstm_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(1)
def make(cap):
    torch.manual_seed(1)
    m = U.TemporalUNetDualView(1, 1, base_ch=64, use_skip_lstm=True).cuda().train()
    o = U.FusedAdamW(m.parameters(), lr=1e-3, weight_decay=1e-4, max_grad_norm=1.0, capturable=cap)
    return m, o
d = U.SyntheticSequences(B, 20, 64, 64, seed=2, kind="uniform")
m1, o1 = make(False)
m2, o2 = make(True)
if os.environ.get("SINGLE_STREAM"):
    ops.ASYNC_WGRAD = False
    ops.BN_RUNNING_ON_SIDE = False
    ops.PARAM_GRADS_ON_SIDE = False
    ops.PREPACK = False
for _ in range(2):
    l1, _ = U.train_step(m1, o1, d.x, d.y, None, False)
g = U.GraphedTrainStep(m2, o2, d.x, d.y, None, False, warmup=2)   # 2 eager steps, then the capture (which executes nothing)
for _ in range(4):
    l1, _ = U.train_step(m1, o1, d.x, d.y, None, False)
    l2, _ = g(d.x, d.y)
torch.cuda.synchronize()
pa, pb = o1.flat.flat_p, o2.flat.flat_p
print("after 6 steps: loss", float(l1), float(l2), "param rel diff", float((pa - pb).norm() / pa.norm()), "step", o1.step_count, float(o2.hyper[6]))
U.quiesce_host_gc()
for name, fn in (("eager", lambda: U.train_step(m1, o1, d.x, d.y, None, False)), ("graph", lambda: g(d.x, d.y))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"B={B} {name}: {dt / 20 * 1e3:.2f} ms/step (host enqueue {th / 20 * 1e3:.2f} ms/step)")
