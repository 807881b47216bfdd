"""A stand-in for mcmc_ref_hip.shard.Communicator backed by torch.distributed/gloo, for the CPU rehearsal of the
N > 1 control flow (world_size 2 on a box without GPUs, or several ranks sharing one GPU, which RCCL itself
refuses).  Test infrastructure only: the product's collective is ncclAllGather inside libmcmcref_hip."""
from __future__ import annotations

import numpy as np


class GlooComm:
    def __init__(self, dist):
        self.dist = dist
        self.world, self.rank = dist.get_world_size(), dist.get_rank()

    def all_gather(self, arr):
        import torch
        a = np.ascontiguousarray(arr, dtype=np.float64)
        out = torch.empty((self.world * max(a.size, 1),), dtype=torch.float64)
        src = torch.from_numpy(a.reshape(-1).copy()) if a.size else torch.zeros(1, dtype=torch.float64)
        self.dist.all_gather_into_tensor(out, src)
        return out.numpy().reshape((self.world,) + (a.shape if a.size else (1,)))

    def all_reduce(self, vals, op="max"):
        import torch
        t = torch.tensor(np.array(vals, dtype=np.float64, ndmin=1))
        self.dist.all_reduce(t, op={"sum": self.dist.ReduceOp.SUM, "max": self.dist.ReduceOp.MAX,
                                    "min": self.dist.ReduceOp.MIN}[op])
        return t.numpy()

    def barrier(self):
        self.dist.barrier()

    def close(self):
        pass
