"""Native P2P engine (csrc/p2p.cpp) on the real RCCL with a world of one: every group is a coalesced send + receive to
this rank itself (RCCL executes it as a local copy), so tickets, the dedicated stream, the compute-stream fences and
the timing source run on one GPU.  Run under a short timeout: an unsupported self-transfer would hang."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ddl import comm
from ddl.parallel.p2p import P2PChannel

ctx = comm.init("nccl", "auto", force=True)
dev = ctx.device
print("init ok", flush=True)
chan = P2PChannel(None, dev)
assert chan.native is not None, "native engine expected on RCCL"
x = torch.empty(30, 28, 28, 128, device=dev)
ys = [torch.zeros_like(x) for _ in range(4)]
tickets = []
for i in range(4):
    x.fill_(float(i + 1))                                   # payload produced on the compute stream right before the post
    tickets.append(chan.post(sends=[(x, 0)], recvs=[(ys[i], 0)]))
    chan.wait(tickets[-1])                                  # compute stream waits: x may be overwritten next round
    print("group", i, "ticket", tickets[-1], flush=True)
for i in range(4):
    assert float(ys[i].min()) == float(ys[i].max()) == float(i + 1), (i, float(ys[i].min()), float(ys[i].max()))
chan.drain()
torch.cuda.synchronize()
print("tickets", tickets, "groups", chan.groups, "outstanding", chan.native.outstanding, flush=True)
assert tickets == [0, 1, 2, 3] and chan.native.outstanding == 0

# timing source (what the step profiler reads as "p2p_ms") and a group that is NOT waited for before the drain
chan.native.set_timing(True)
z = torch.zeros_like(x)
x.fill_(7.0)
t = chan.post(sends=[(x, 0)], recvs=[(z, 0)])
chan.drain()                                               # host-side: every posted group has completed
torch.cuda.synchronize()
ms = chan.native.drain_ms()
assert float(z.min()) == 7.0 and ms >= 0.0, (float(z.min()), ms)
print(f"timed group: {ms:.3f} ms on the P2P stream", flush=True)

# --debug-sync form: the same groups issued on the compute stream itself
chan2 = P2PChannel(None, dev, debug_sync=True)
w = torch.zeros_like(x)
x.fill_(9.0)
chan2.wait(chan2.post(sends=[(x, 0)], recvs=[(w, 0)]))
chan2.drain()
torch.cuda.synchronize()
assert float(w.min()) == float(w.max()) == 9.0
print("P2P SELF OK", flush=True)
comm.destroy()
