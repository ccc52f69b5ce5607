"""Run under torch.distributed.run: BASELINE config 5's multi-GPU scheme (image tiles, no arithmetic exchange).
Every rank renders its interleaved row bands of N progressive frames of the same scene (band by band, all bands of a frame in
one set of launches, and all bands of ALL frames in one set of launches: rt_pipeline_render_bands_batch), the bands are combined
with one all-gather (and, for comparison, with one all-reduce), and rank 0 checks the result bit for bit against the
whole frame rendered by itself.
Test hook: DXR_BENCH_DEVICE / DXR_BENCH_BACKEND as in bench.py (two ranks on the one GPU of a test box, gloo)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from dxrexperiments_amd import capi, distributed as D, rtypes as T, scenes  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    device = int(os.environ.get("DXR_BENCH_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    assert torch.cuda.is_available()
    torch.cuda.set_device(device)
    dev = torch.device("cuda", device)
    dist.init_process_group(os.environ.get("DXR_BENCH_BACKEND", "nccl"))
    W, H, frames = 320, 200, 9
    v, t = scenes.displaced_grid(96, seed=7)
    ctx = capi.Context(device, stream=torch.cuda.current_stream().cuda_stream)
    scene = capi.Scene(ctx)
    scene.add_model(capi.Model(ctx, v, t))
    mat = T.default_material()
    mat["type"] = 2
    cam = capi.camera_array((0.0, 6.0, 19.0), (0.0, -4.0, 0.0), (0, 1, 0), 0.8, W / H)
    host = capi.ProgressiveHost(5)
    pfcs = [host.update(cam, 0.0, f + 1, W, H) for f in range(frames)]

    def render(rows, bands=None, sets=False):
        img = torch.zeros((H, W, 4), dtype=torch.float32, device=dev)
        pipe = capi.Pipeline(ctx)
        pipe.set_scene(scene)
        pipe.add_material(mat)
        pipe.set_depth_limits(4, 2)
        pipe.bind_output(img.data_ptr(), W, H)
        pipe.build_acceleration_structures()
        if sets:                                         # all frames of this rank's bands through shared sets of launches (round 4)
            pipe.render_bands_batch(*bands, pfcs)
        for pfc in ([] if sets else pfcs):
            pipe.update(pfc)
            if bands:
                pipe.render_bands(*bands)                # all bands of a rank in one set of launches
            else:
                for (y0, y1) in rows:
                    pipe.render(tile=(0, y0, W, y1))
        ctx.synchronize()
        return img

    mine = render(D.tile_rows(rank, world, H, band=16))
    assert torch.equal(mine, render(None, bands=(16, rank, world)))      # rt_pipeline_render_bands == band by band
    assert torch.equal(mine, render(None, bands=(16, rank, world), sets=True))      # rt_pipeline_render_bands_batch == frame by frame
    whole = D.gather_tiles(mine.clone(), band=16)            # one all-gather of the disjoint bands
    summed = D.combine_tiles(mine)                           # the simpler form: one SUM all-reduce over a zero background
    torch.cuda.synchronize()
    assert torch.equal(whole, summed)
    if rank == 0:
        ref = render([(0, H)])
        torch.cuda.synchronize()
        assert torch.equal(whole, ref), "tiled image differs from the single-GPU frame"
        assert float(ref[..., 3].min()) == 1.0
        print("tiles ok: %d ranks, %dx%d, %d frames" % (world, W, H, frames))
    dist.barrier()
    dist.destroy_process_group()


main()
