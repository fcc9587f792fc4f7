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
    pointers: the HIP product (cuda tensors, RCCL) and, in the CPU tests, the oracle (cpu tensors, gloo)."""

    def __init__(self, engine, n_total, rank, world, device, dist=None):
        import torch

        self.torch = torch
        self.dist = dist
        self.engine = engine
        self.rank, self.world = rank, world
        self.first, self.count = shard_range(n_total, rank, world)
        self.slot = max_shard(n_total, world)  # all_gather needs equal contributions: pad with radius < 0
        self.local = torch.zeros((self.slot, 8), dtype=torch.float32, device=device)
        self.local[:, 3] = -1.0
        self.all = torch.zeros((world * self.slot, 8), dtype=torch.float32, device=device)
        self.self_offset = rank * self.slot

    def step(self, dt=1.0 / 60.0, stages=abi.STAGE_ALL, gravity=(0.0, -98.0, 0.0)):
        """One fixed step with character-vs-character sweeps. The snapshot is taken inside
        KinematicMoveStopSystem, i.e. after this step's intent and gravity on every rank."""
        eng = self.engine
        pre = stages & (abi.STAGE_INTENT | abi.STAGE_GRAVITY)
        if pre:
            eng.tick(dt=dt, stages=pre, gravity=gravity)
        eng.agents_export(self.local.data_ptr())
        eng.synchronize()  # the engine's stream -> torch's stream
        if self.world > 1:
            self.dist.all_gather_into_tensor(self.all, self.local)
            if self.all.is_cuda:
                self.torch.cuda.current_stream().synchronize()
        else:
            self.all.copy_(self.local)
            if self.all.is_cuda:
                self.torch.cuda.current_stream().synchronize()
        eng.agents_import(self.all.data_ptr(), self.all.shape[0], self.self_offset)
        eng.tick(dt=dt, stages=(stages & ~(abi.STAGE_INTENT | abi.STAGE_GRAVITY)) | abi.STAGE_AGENTS, gravity=gravity)
