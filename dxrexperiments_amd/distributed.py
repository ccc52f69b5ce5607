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

from . import capi


def shard_frames(rank, world_size, n_frames, first_frame=0):
    """Frame indices rank `rank` renders: first_frame + {f : f mod world_size == rank}."""
    return list(range(first_frame + rank, first_frame + n_frames, world_size))


def _native_partitions():
    """The partition functions are host logic of the C ABI (no device needed), but the shared library links the HIP runtime:
    on a box where it cannot even be loaded, the same two formulas are evaluated here.  Rendering has no such
    fallback."""
    try:
        capi.lib()
        return True
    except (ImportError, OSError):
        return False


def frames_per_rank(world_size, n_frames):
    """Frames each rank renders: the C ABI's own partition (rt_shard_frame_count), so C++ and Python callers agree."""
    if not _native_partitions():
        return [(n_frames - r + world_size - 1) // world_size if n_frames > r else 0 for r in range(world_size)]
    return [capi.shard_frame_count(r, world_size, n_frames) for r in range(world_size)]


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


def tile_rows(rank, world_size, height, band=16):
    """Partitioning B (image tiles): interleaved bands of `band` rows owned by `rank`, as (y0, y1) pairs -- the C ABI's
    rt_tile_bands.  Pixels are seeded by their GLOBAL index, so the tiled image is bit-identical to the single-GPU one
    and needs no arithmetic exchange."""
    if not _native_partitions():
        return [(b * band, min((b + 1) * band, height)) for b in range(rank, (height + band - 1) // band, world_size)]
    return capi.tile_bands(height, band, rank, world_size)


def combine_tiles(image, group=None):
    """Partitioning B combined with ONE SUM all-reduce: every rank rendered its own bands into an otherwise ZERO buffer
    (rt_pipeline_render_tile leaves foreign pixels untouched), so the sum is a gather -- x + 0 == x exactly.  Simple, but
    it moves two images per rank; gather_tiles moves one."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(image, op=dist.ReduceOp.SUM, group=group)
    return image


def gather_tiles(image, band=16, group=None):
    """Partitioning B combined with ONE all-gather of the disjoint bands (what rt_dist_gather_bands does from C): rank r
    packs its bands b = r, r + R, ... into ceil(bands / R) slots, the all-gather delivers every rank's slots, and each band
    is copied to its rows.  Every rank receives (R-1)/R of one image: half the bytes of the SUM all-reduce.  In place on
    `image` (H, W, 4 float32); foreign rows need not be zero."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return image
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    H, W = image.shape[0], image.shape[1]
    slots, floats = capi.tile_gather_layout(W, H, band, world)
    send = torch.zeros((slots, band, W, 4), dtype=image.dtype, device=image.device)
    for s, (y0, y1) in enumerate(tile_rows(rank, world, H, band)):
        send[s, :y1 - y0] = image[y0:y1]
    assert send.numel() == floats
    flat = torch.empty((world * slots, band, W, 4), dtype=image.dtype, device=image.device)     # rank-major concatenation
    dist.all_gather_into_tensor(flat, send, group=group)
    recv = flat.view(world, slots, band, W, 4)
    for r in range(world):
        if r == rank:
            continue
        for s, (y0, y1) in enumerate(tile_rows(r, world, H, band)):
            image[y0:y1] = recv[r, s, :y1 - y0]
    return image
