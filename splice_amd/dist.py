"""Process-per-GPU replica helpers: independent image pairs, one
optimisation loop per GPU, NO data-path collective.  torch.distributed is used only for the start/stop
barrier and the max-over-ranks reduction of the elapsed time that bench.py reports: the default backend
is gloo on the host (north_star: "no RCCL required" -- an RCCL communicator is only created when a caller
asks for backend "nccl" explicitly)."""
import os


class Replicas:
    def __init__(self, backend=None, device=None):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dist = None
        self.device = device
        if self.world > 1:
            import torch.distributed as dist
            if not dist.is_initialized():
                kw = {}
                if backend == "nccl" and device is not None:
                    import torch
                    kw["device_id"] = torch.device(device)
                dist.init_process_group(backend or "gloo", **kw)
            self.dist = dist

    def pair_id(self):
        """Pair handled by this rank: pair i -> rank i mod world (one pair per rank in the benchmark)."""
        return self.rank

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        import torch
        dev = self.device if (self.device is not None and self.dist.get_backend() == "nccl") else "cpu"
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.item()

    def gather_floats(self, value):
        """All ranks' values (list ordered by rank) -- used by tests to check replica independence."""
        if self.dist is None:
            return [float(value)]
        import torch
        dev = self.device if (self.device is not None and self.dist.get_backend() == "nccl") else "cpu"
        t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
        out = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [o.item() for o in out]

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()


def aggregate_throughput(steps_per_rank, world, elapsed_max):
    """bench.py's `value`: whole-job steps / max-over-ranks time (weak scaling, fixed work per GPU)."""
    return steps_per_rank * world / elapsed_max
