"""P2P channel of the pipeline schedules: posted transfer groups + tickets.

On a GPU with an RCCL group this is a thin face of the native ``P2PEngine`` (csrc/p2p.cpp): transfers are issued
from dedicated P2P streams, a send group is fenced behind the compute stream at the point its payload was
produced, receives can be posted ahead of the compute that consumes them, and ``wait(ticket)`` makes the compute
stream (never the host) wait for exactly one group.  Over gloo - CPU plumbing tests, or several ranks sharing one
GPU - the same interface runs on ``isend`` / ``irecv`` with host staging for GPU tensors.

Transport (``rccl``: auto | direct | c10d).  ``direct`` - the default when the channel knows its ``neighbours`` -
gives every neighbour its OWN 2-rank RCCL communicator and its OWN HIP stream (csrc/rccl_comm.cpp): the grouped
``ncclSend`` / ``ncclRecv`` are enqueued on that stream itself, and a receive pre-posted toward the previous stage
can never sit in front of a send toward the next one (c10d runs every transfer of a process group FIFO on one
internal stream).  The communicators are created in one global order - pairs sorted by (lower, higher) group rank
- so no two ranks ever wait for each other in different ``ncclCommInitRank`` calls.

Reference: ``torch.distributed.pipelining`` issues ``batch_isend_irecv`` from the schedule loop on the compute
stream and waits right away (pp.py:138-150 -> torch schedules.py:505,758-800).
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .. import comm, ops
from ..utils.streams import new_stream, new_stream_beside

Transfer = Tuple[torch.Tensor, int]          # (tensor, GLOBAL rank of the peer)


class _HostRecv:
    """gloo cannot move GPU tensors: receive into host memory, copy up when waited for."""

    def __init__(self, work, host: torch.Tensor, dev: torch.Tensor) -> None:
        self.work, self.host, self.dev = work, host, dev

    def wait(self) -> None:
        self.work.wait()
        self.dev.copy_(self.host)


def link_pairs(me: int, neighbours: Sequence[int]) -> List[Tuple[int, int]]:
    """The (lower, higher) group-rank pairs this rank shares a link communicator with, in the ONE global creation
    order every rank follows (sorted).  `ncclCommInitRank` blocks until both members have joined, so two ranks must
    never wait for each other in different inits: with every rank walking its pairs in ascending (lower, higher) order
    the pair that is globally smallest among all unfinished ones always has both members at it
    (`tests/test_dist.py::test_link_creation_order_never_deadlocks` simulates chains and rings)."""
    return sorted({tuple(sorted((int(g), int(me)))) for g in neighbours if g is not None})


class P2PChannel:
    def __init__(self, group, device, debug_sync: bool = False, profiler=None,
                 neighbours: Optional[Sequence[int]] = None, rccl: str = "auto", test_hook: int = 0,
                 hook_spin_us: float = 0.0, skip_fence: int = 0) -> None:
        """``neighbours``: GLOBAL ranks this channel will ever talk to (a pipeline stage: previous / next stage).
        ``test_hook`` / ``skip_fence``: fence tests only (csrc/p2p.cpp)."""
        self.group = group
        self.device = torch.device(device)
        self.profiler = profiler
        self.native = None
        self.stream = None
        self.streams: List = []          # every stream transfers are issued from (lane streams must avoid them)
        self.transport = "host"
        self.groups = 0
        self._next = 0
        self._pending = {}
        if (self.device.type == "cuda" and comm.is_dist() and comm.backend_of(group) == "nccl" and ops.have_ext()
                and os.environ.get("DDL_NATIVE_P2P", "1") == "1"):
            # beside the compute stream: a P2P stream that shares its hardware queue would carry the transfers only in
            # the gaps between compute kernels (utils/streams.py)
            self.stream = new_stream_beside(self.device, [torch.cuda.current_stream(self.device)])
            self.streams = [self.stream]
            s = self.stream
            links = []
            me = comm.get_rank(group)
            if neighbours and comm.rccl_transport(rccl) == "direct":
                pairs = link_pairs(me, [self._local(int(g)) for g in neighbours if g is not None])
                err, made = "", []
                try:
                    for a, b in pairs:                       # one global creation order: (lower, higher) group rank
                        peer = b if a == me else a
                        members = sorted({a, b})
                        globals_ = [dist.get_global_rank(group, r) if group is not None else r for r in members]
                        c = comm.direct_comm(globals_, self.device, tag="pp-link")
                        made.append(((("pp-link"), tuple(globals_)), c))
                        st = new_stream_beside(self.device, [torch.cuda.current_stream(self.device)], avoid=self.streams)
                        self.streams.append(st)
                        links.append((peer, c, members.index(peer), st.stream_id, st.device_index, st.device_type))
                except Exception as e:      # noqa: BLE001 - the verdict is taken together, below
                    err = f"{type(e).__name__}: {e}"
                # every stage of the pipeline must use the same transport: agree over the pipeline group
                ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=self.device)
                if comm.get_world_size(group) > 1:
                    dist.all_reduce(ok, dist.ReduceOp.MIN, group=group)
                if int(ok.item()) != 1:
                    import sys
                    print(f"[rank {comm.get_rank()}] P2PChannel: per-neighbour RCCL links not available on every stage "
                          f"({err or 'a peer failed'}) - falling back to the c10d transport", file=sys.stderr, flush=True)
                    for key, c in made:
                        comm._DIRECT.pop(key, None)
                        c.abort()
                    links, self.streams = [], [self.stream]
            self.transport = "direct" if links else "c10d"
            self.native = ops.require_ext().P2PEngine(
                group if group is not None else dist.group.WORLD, s.stream_id, s.device_index, s.device_type,
                bool(debug_sync), bool(profiler is not None and profiler.enabled), links, me,
                comm.get_world_size(group), int(test_hook), float(hook_spin_us), int(skip_fence))
            if profiler is not None:
                profiler.add_source("p2p", self.native.drain_ms, self.native.set_timing)

    def _local(self, global_rank: int) -> int:
        return dist.get_group_rank(self.group, global_rank) if self.group is not None else global_rank

    def post(self, sends: Sequence[Transfer] = (), recvs: Sequence[Transfer] = (),
             after_compute: Optional[bool] = None) -> Optional[int]:
        """Issue ONE group (sends and receives complete together).  Returns a ticket for ``wait``; ``None`` for an
        empty group.  ``after_compute`` defaults to "the group carries a payload" (has sends)."""
        sends, recvs = list(sends), list(recvs)
        if not sends and not recvs:
            return None
        self.groups += 1
        if self.native is not None:
            if after_compute is None:
                after_compute = bool(sends)
            return self.native.post([t for t, _ in sends], [self._local(r) for _, r in sends],
                                    [t for t, _ in recvs], [self._local(r) for _, r in recvs], bool(after_compute))
        works = []
        for t, peer in sends:
            payload = t.detach()
            if payload.is_cuda:
                payload = payload.cpu()
            works.append(dist.isend(payload.contiguous(), peer, group=self.group))
        for t, peer in recvs:
            if t.is_cuda:
                host = torch.empty(t.shape, dtype=t.dtype)
                works.append(_HostRecv(dist.irecv(host, peer, group=self.group), host, t))
            else:
                works.append(dist.irecv(t, peer, group=self.group))
        ticket = self._next
        self._next += 1
        self._pending[ticket] = works
        return ticket

    def wait(self, ticket: Optional[int]) -> None:
        """The compute stream (native) / the host (gloo) waits for one posted group."""
        if ticket is None:
            return
        if self.native is not None:
            self.native.wait(ticket)
            return
        for w in self._pending.pop(ticket, ()):
            w.wait()

    def fence(self) -> None:
        """Transfers posted from now on start after everything the compute stream has enqueued so far (slot reuse
        across steps: a pre-posted receive must not land before the previous step's consumer has read the slot).
        The gloo twin is host-synchronous and needs nothing."""
        if self.native is not None:
            self.native.fence()

    def drain(self) -> None:
        if self.native is not None:
            self.native.drain()
            return
        for t in sorted(self._pending):
            for w in self._pending[t]:
                w.wait()
        self._pending.clear()
