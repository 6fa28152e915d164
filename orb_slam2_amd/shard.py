"""Frame sharding across GPUs of one node.

Every frame is independent (the reference keeps no state between operator() calls), so the
multi-GPU path is: rank r takes a contiguous run of stereo pairs, extracts them on its own GPU,
and only scalars (timings, counts) cross ranks.  No collective touches image or keypoint data.
A stereo pair (left, right = frames 2k, 2k+1) always stays on one rank, so that stereo matching
(reference src/Frame.cc:485-659) can follow on the same device.
"""
import os


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def pair_shard(n_pairs, rank, world):
    """[first, last) stereo-pair indices of `rank`; sizes differ by at most one pair."""
    base, rem = divmod(n_pairs, world)
    first = rank * base + min(rank, rem)
    return first, first + base + (1 if rank < rem else 0)


def frame_shard(n_frames, rank, world):
    """[first, last) frame indices of `rank` (pairs kept together); the same cut as the library's
    orbx_host_frame_shard, which the in-process worker pool uses."""
    a, b = pair_shard((n_frames + 1) // 2, rank, world)
    return min(2 * a, n_frames), min(2 * b, n_frames)


def max_over_ranks(value, dist=None, device=None):
    """MAX all-reduce of one float (the job's wall time is its slowest rank's)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist=None, device=None):
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
