"""Multi-GPU partitioning of the progressive path (one process per GPU).

The reference is single-GPU.  Frames of a progressive render are
independent given their per-frame constants, so the path shards without any
exchange until the very end (partitioning A):

    rank r renders frames {f : f mod R == r} into a local fp32 SUM buffer
    (RT_ACCUM_SUM), then ONE all-reduce(sum) of the w*h*4 floats over RCCL/xGMI
    combines the shards and the result is divided by the total frame count.

This equals the reference's running mean (ProgressiveRaytracing.hlsl:36-38) up
to fp32 re-association.  The scene and its BVH are replicated on every GPU.
The functions below hold only the host logic; they work on any torch device /
backend (the tests drive them over gloo on CPU).
"""
import torch
import torch.distributed as dist


def shard_frames(rank, world_size, n_frames, first_frame=0):
    """Frame indices rank `rank` renders: first_frame + {f : f mod world_size == rank}."""
    return list(range(first_frame + rank, first_frame + n_frames, world_size))


def frames_per_rank(world_size, n_frames):
    return [len(range(r, n_frames, world_size)) for r in range(world_size)]


def reduce_accumulation(sum_buffer, n_local_frames, group=None):
    """All-reduce the per-rank SUM buffers in place and return (mean_image, total_frames).

    sum_buffer: float32 tensor (H, W, 4) holding the sum of this rank's frames.
    A single collective moves w*h*4 floats (33.2 MB at 1080p); the frame count rides in
    a second, 1-element all-reduce so ranks may hold different numbers of frames."""
    count = torch.tensor([float(n_local_frames)], dtype=torch.float64, device=sum_buffer.device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(sum_buffer, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(count, op=dist.ReduceOp.SUM, group=group)
    total = int(count.item())
    return sum_buffer / float(max(total, 1)), total


def tile_rows(rank, world_size, height, band=64):
    """Partitioning B (image tiles): interleaved bands of `band` rows owned by `rank`,
    as (y0, y1) pairs.  Pixels are seeded by their GLOBAL index, so the tiled image is
    bit-identical to the single-GPU one and needs no arithmetic exchange."""
    out = []
    for b, y0 in enumerate(range(0, height, band)):
        if b % world_size == rank:
            out.append((y0, min(y0 + band, height)))
    return out


def combine_tiles(image, group=None):
    """Partitioning B, the only exchange: every rank rendered its own bands into an otherwise ZERO buffer
    (rt_pipeline_render_tile leaves foreign pixels untouched), so a SUM all-reduce is a gather -- x + 0 == x
    exactly -- and every rank ends up with the whole image, bit-identical to the single-GPU frame."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(image, op=dist.ReduceOp.SUM, group=group)
    return image
