"""Stream -> GPU assignment: streams are independent, so a node shards them with no
data-path collective.  A stream stays on one rank for life (its overlap delay and resampler history
never migrate).  torch.distributed is used only to agree on a wall-clock figure."""


def owner_of(stream, world):
    """Rank that owns global stream index `stream`."""
    return stream % world


def local_streams(total_streams, world, rank):
    """Global stream indices owned by `rank` (stream s -> rank s mod world)."""
    return list(range(rank, total_streams, world))


def reduce_elapsed(elapsed_s, device=None):
    """max over ranks of the timed region; identity when torch.distributed is not initialised."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return elapsed_s
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_units(units, device=None):
    """sum over ranks of the units each rank processed (host-side counter; not on the data path)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return units
    t = torch.tensor([units], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
