import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_gpu_native import _ref_stage, _native_from, _rand
dev = torch.device("cuda", 0)

def run(units, shape):
    ref = _ref_stage(units, dev); nat = _native_from(ref, units, dev)
    x = _rand(*shape, dev=dev, seed=5)
    xr = x.clone().requires_grad_(True)
    xn = x.permute(0, 2, 3, 1).contiguous().clone().requires_grad_(True)
    ref.train(); nat.train()
    yr = ref(xr); yn = nat(xn)
    last = units[-1] == "head"
    yc = yn if last else yn.permute(0, 3, 1, 2)
    ferr = float((yc - yr).abs().max() / yr.abs().max())
    gy = _rand(*yr.shape, dev=dev, seed=6)
    yr.backward(gy); nat.arena.zero_grad()
    yn.backward(gy if last else gy.permute(0, 2, 3, 1).contiguous())
    gerr = float((xn.grad.permute(0, 3, 1, 2) - xr.grad).abs().max() / xr.grad.abs().max())
    # double-precision reference of the same thing
    ref64 = _ref_stage(units, dev).double(); ref64.load_state_dict({k: v.double() for k, v in ref.state_dict().items()})
    for m in ref64.modules():
        if isinstance(m, torch.nn.BatchNorm2d): pass
    ref64.train()
    x64 = x.double().requires_grad_(True)
    # NB: ref's buffers were already updated by one forward; irrelevant in train mode
    y64 = ref64(x64); y64.backward(gy.double())
    g32 = float((xr.grad.double() - x64.grad).abs().max() / x64.grad.abs().max())
    gnat = float((xn.grad.permute(0, 3, 1, 2).double() - x64.grad).abs().max() / x64.grad.abs().max())
    worst = 0
    rg = dict(ref.named_parameters())
    for name, p in nat.named_parameters():
        g = nat.ddl_export_tensor(name, p.grad); r = rg[name].grad
        worst = max(worst, float((g - r).abs().max() / r.abs().max().clamp_min(1e-6)))
    print(f"{units} {shape}: fwd relerr {ferr:.2e}  dx relerr(native vs torch32) {gerr:.2e}  torch32-vs-fp64 {g32:.2e}  native-vs-fp64 {gnat:.2e}  worst param grad {worst:.2e}", flush=True)

run(("denseblock3",), (2, 256, 4, 4))
run(("denseblock3",), (2, 256, 8, 8))
run(("denseblock2", "transition2"), (2, 128, 8, 8))
run(("transition2", "denseblock3"), (2, 512, 8, 8))
run(("denseblock2", "transition2", "denseblock3"), (2, 128, 8, 8))
run(("denseblock2", "transition2", "denseblock3"), (4, 128, 16, 16))
