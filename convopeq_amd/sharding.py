"""Multi-GPU layout of the path: streams are independent (one MKLNonUniformConvolver pair and one filterState per
stream in the reference, src/ConvolverProcessor.h:669, src/eqprocessor/EQProcessor.h:637), so they shard across
ranks with NO data-path collective.  The only collective is the end-of-run reduction of the counters
(torch.distributed: RCCL on GPUs, gloo in the CPU tests)."""
import torch
import torch.distributed as dist


def streams_of_rank(total_streams, world, rank):
    """Global stream ids owned by `rank`: stream s lives on rank s mod world."""
    return list(range(rank, total_streams, world))


def weak_scaling_streams(streams_per_gpu, world, rank):
    """Weak scaling (bench.py): every rank owns `streams_per_gpu` streams; global ids are rank-major."""
    base = rank * streams_per_gpu
    return list(range(base, base + streams_per_gpu))


def reduce_counters(samples, elapsed_s, err_sq_sum=0.0, err_max=0.0, device="cpu"):
    """Whole-job counters: SUM of samples and squared error, MAX of elapsed time and abs error."""
    if not (dist.is_available() and dist.is_initialized()):
        return samples, elapsed_s, err_sq_sum, err_max
    s = torch.tensor([float(samples), float(err_sq_sum)], dtype=torch.float64, device=device)
    m = torch.tensor([float(elapsed_s), float(err_max)], dtype=torch.float64, device=device)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return s[0].item(), m[0].item(), s[1].item(), m[1].item()


def gather_values(value, device="cpu"):
    """One float per rank, in rank order (per-rank throughput: a slow GPU shows up as the minimum)."""
    if not (dist.is_available() and dist.is_initialized()):
        return [float(value)]
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [t.item() for t in out]
