"""P2P channel of the pipeline schedules: posted transfer groups + tickets.

On a GPU with an RCCL group this is a thin face of the native ``P2PEngine`` (csrc/p2p.cpp): transfers are issued
from a dedicated P2P stream, a send group is fenced behind the compute stream at the point its payload was
produced, receives can be posted ahead of the compute that consumes them, and ``wait(ticket)`` makes the compute
stream (never the host) wait for exactly one group.  Over gloo - CPU plumbing tests, or several ranks sharing one
GPU - the same interface runs on ``isend`` / ``irecv`` with host staging for GPU tensors.

Reference: ``torch.distributed.pipelining`` issues ``batch_isend_irecv`` from the schedule loop on the compute
stream and waits right away (pp.py:138-150 -> torch schedules.py:505,758-800).
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .. import comm, ops
from ..utils.streams import new_stream

Transfer = Tuple[torch.Tensor, int]          # (tensor, GLOBAL rank of the peer)


class _HostRecv:
    """gloo cannot move GPU tensors: receive into host memory, copy up when waited for."""

    def __init__(self, work, host: torch.Tensor, dev: torch.Tensor) -> None:
        self.work, self.host, self.dev = work, host, dev

    def wait(self) -> None:
        self.work.wait()
        self.dev.copy_(self.host)


class P2PChannel:
    def __init__(self, group, device, debug_sync: bool = False, profiler=None) -> None:
        self.group = group
        self.device = torch.device(device)
        self.profiler = profiler
        self.native = None
        self.stream = None
        self.groups = 0
        self._next = 0
        self._pending = {}
        if (self.device.type == "cuda" and comm.is_dist() and comm.backend_of(group) == "nccl" and ops.have_ext()
                and os.environ.get("DDL_NATIVE_P2P", "1") == "1"):
            self.stream = new_stream(self.device)
            s = self.stream
            self.native = ops.require_ext().P2PEngine(
                group if group is not None else dist.group.WORLD, s.stream_id, s.device_index, s.device_type,
                bool(debug_sync), bool(profiler is not None and profiler.enabled))
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
