"""Shard-by-character across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in the CPU
tests). Characters are split into contiguous blocks [g*N/W, (g+1)*N/W); skeleton, profiles,
source mesh and the static collision world are replicated. Configs 2-4 need no data-path
communication at all. Only character-vs-character collision (config 5) exchanges data:
one all-gather of the 32-byte AgentSweepState snapshot per step (Systems.swift:1592-1611
builds that snapshot before the per-entity loop), after which every rank bins all
capsules into its own XZ grid.
"""
import numpy as np

from . import abi


def shard_range(n_total, rank, world):
    """Contiguous block of characters owned by `rank` (first, count)."""
    first = (n_total * rank) // world
    last = (n_total * (rank + 1)) // world
    return first, last - first


def max_shard(n_total, world):
    return max(shard_range(n_total, r, world)[1] for r in range(world))


class AgentExchange:
    """The config-5 exchange. Works on any engine exposing agents_export/agents_import with raw
    pointers: the HIP product (cuda tensors, RCCL) and, in the CPU tests, the oracle (cpu tensors, gloo).

    On the product nothing here synchronises with the host: the export kernel, the collective and the import are ordered on the
    engine's own stream — torch sees that stream as an ExternalStream, and ProcessGroupNCCL orders its collective behind / ahead
    of the current stream with events. With one rank the whole exchange is the library's own C entry point
    (sge_agents_allgather, which a Swift / C++ host calls with its ncclComm_t)."""

    def __init__(self, engine, n_total, rank, world, device, dist=None):
        import torch

        self.torch = torch
        self.dist = dist
        self.engine = engine
        self.rank, self.world = rank, world
        self.first, self.count = shard_range(n_total, rank, world)
        self.slot = max_shard(n_total, world)  # all_gather needs equal contributions: pad with radius < 0
        self.self_offset = rank * self.slot
        self.product = bool(getattr(engine.t, "is_product", False))
        self.stream = None
        self.staged = False
        import os
        self.force_sync = os.environ.get("SGE_EXCHANGE_SYNC", "0") not in ("", "0")  # host synchronisation around the collective
        self.checked = False
        self.self_check = None  # "passed" once the first stream-ordered exchange has been compared with a synchronised one
        if self.product and world == 1:
            return  # sge_agents_allgather keeps its own buffers
        self.staged = self.product and dist is not None and dist.get_backend() != "nccl"  # rehearsal: gloo moves host memory
        if self.product:
            self.stream = torch.cuda.ExternalStream(engine.stream_handle(), device=device)
        ctx = torch.cuda.stream(self.stream) if self.stream is not None else _Null()
        with ctx:  # allocation and fill on the stream that will use the buffers
            self.local = torch.zeros((self.slot, 8), dtype=torch.float32, device=device)
            self.local[:, 3] = -1.0
            self.all = torch.zeros((world * self.slot, 8), dtype=torch.float32, device=device)

    def describe(self):
        """What a bench record says about the exchange that ran."""
        if self.product and self.world == 1:
            path = "sge_agents_allgather (one rank: export + import, no collective)"
        elif self.product and not self.staged:
            path = "stream-ordered: export kernel -> all_gather_into_tensor (RCCL) -> import, on the engine's stream, no host synchronisation" + (
                " [SGE_EXCHANGE_SYNC: host synchronisation around the collective]" if self.force_sync else "")
        elif self.staged:
            path = "staged rehearsal: export -> host memory -> gloo all-gather -> device -> import (synchronising)"
        else:
            path = "host arrays (CPU oracle under test)"
        return {"path": path, "records_per_rank": self.slot, "bytes_per_rank": self.slot * 32, "self_check": self.self_check}

    def step(self, dt=1.0 / 60.0, stages=abi.STAGE_ALL, gravity=(0.0, -98.0, 0.0)):
        """One fixed step with character-vs-character sweeps. The snapshot is taken inside
        KinematicMoveStopSystem, i.e. after this step's intent and gravity on every rank."""
        eng = self.engine
        pre = stages & (abi.STAGE_INTENT | abi.STAGE_GRAVITY)
        if pre:
            eng.tick(dt=dt, stages=pre, gravity=gravity)
        if self.product and self.world == 1:
            eng.agents_allgather(None, 0, 1, self.slot)
        elif self.product and not self.staged:
            with self.torch.cuda.stream(self.stream):
                eng.agents_export(self.local.data_ptr())
                if self.force_sync:
                    eng.synchronize()
                self.dist.all_gather_into_tensor(self.all, self.local)
                if self.force_sync:
                    self.torch.cuda.synchronize()
            if not self.checked:
                # The stream-ordered path (no host synchronisation between export, collective and import) is repeated once with host
                # synchronisation on both sides and compared. A mismatch is an ordering bug, not a condition to run with: every rank
                # learns of it (all-reduce of the verdict, so that no rank is left waiting in a collective) and raises.
                # SGE_EXCHANGE_SYNC=1 is the explicit escape hatch (host synchronisation around every exchange).
                self.checked = True
                eng.synchronize()
                self.torch.cuda.synchronize()
                check = self.torch.empty_like(self.all)
                self.dist.all_gather_into_tensor(check, self.local)
                self.torch.cuda.synchronize()
                differ = int((check != self.all).any(dim=1).sum().item())
                verdict = self.torch.tensor([differ], dtype=self.torch.int64, device=self.all.device)
                self.dist.all_reduce(verdict, op=self.dist.ReduceOp.MAX)
                self.self_check = "passed" if int(verdict.item()) == 0 else "failed"
                if self.self_check == "failed":
                    raise RuntimeError("[sge] rank %d: the stream-ordered agent exchange disagrees with the synchronised one (%d of %d records "
                                       "differ on this rank; some rank saw %d). Set SGE_EXCHANGE_SYNC=1 to run with host synchronisation "
                                       "around the collective." % (self.rank, differ, self.all.shape[0], int(verdict.item())))
            eng.agents_import(self.all.data_ptr(), self.all.shape[0], self.self_offset)
        else:
            eng.agents_export(self.local.data_ptr())
            if self.staged:  # one-GPU rehearsal of N ranks over gloo: through host memory, synchronising
                eng.synchronize()
                host_all = self.torch.zeros(self.all.shape, dtype=self.torch.float32)
                self.dist.all_gather_into_tensor(host_all, self.local.cpu())
                self.all.copy_(host_all)
                self.torch.cuda.synchronize()
            elif self.world > 1:
                self.dist.all_gather_into_tensor(self.all, self.local)
            else:
                self.all.copy_(self.local)
            eng.agents_import(self.all.data_ptr(), self.all.shape[0], self.self_offset)
        eng.tick(dt=dt, stages=(stages & ~(abi.STAGE_INTENT | abi.STAGE_GRAVITY)) | abi.STAGE_AGENTS, gravity=gravity)


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
