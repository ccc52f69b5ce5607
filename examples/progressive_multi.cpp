// Multi-GPU progressive rendering from the C / C++ boundary: one process per GPU, RCCL over xGMI, no Python.
//
//   progressive_multi <model.obj> <width> <height> <frames> <out.pfm|out.png> <gpus> [samples|tiles]
//
// The reference application drives ONE device (src/DXRExperimentsApp.cpp:107-130).  This launcher forks <gpus> copies of
// that application BEFORE anything touches the GPU (a process that has initialised HIP must never be forked or re-exec'd),
// hands rank 0's RCCL unique id round through pipes, and every rank then runs the reference-shaped pipeline
// (dxrexperiments_amd/include) on its own GPU with the scene replicated:
//   samples  rank r renders frames {f : f mod R == r} into an fp32 SUM image; one rt_dist_all_reduce_sum; mean = sum / frames
//   tiles    every rank renders every frame, but only its interleaved 16-row bands; one rt_dist_gather_bands per image
// Rank 0 writes the image.  Both results equal the single-GPU image (tiles: bit for bit; samples: up to fp32 re-association).
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "ProgressiveRaytracingPipeline.h"

using namespace DXRFramework;

static bool write_all(int fd, const void *p, size_t n) { return write(fd, p, n) == (ssize_t)n; }
static bool read_all(int fd, void *p, size_t n)
{
    size_t got = 0;
    while (got < n) {
        const ssize_t r = read(fd, (char *)p + got, n - got);
        if (r <= 0) return false;
        got += (size_t)r;
    }
    return true;
}

static int rank_main(int rank, int world, int id_in, int id_out, char **argv)
{
    const UINT width = std::atoi(argv[2]), height = std::atoi(argv[3]), frames = std::atoi(argv[4]);
    const bool tiles = argv[7] && std::strcmp(argv[7], "tiles") == 0;
    const UINT band_rows = 16;
    try {
        int devices = rt_device_count();
        // (test hook: DXR_MULTI_DEVICE=<ordinal> puts every rank on that device -- rt_dist_create must then refuse, not hang)
        const char *forced = std::getenv("DXR_MULTI_DEVICE");
        if (!forced && devices < world) {
            std::fprintf(stderr, "rank %d: %d ranks but %d visible GPUs (RCCL wants one device per rank)\n", rank, world, devices);
            return 3;
        }
        auto context = RtContext::create(forced ? std::atoi(forced) : rank);
        // rank 0 creates the communicator id; the launcher relays it to the other ranks
        char id[128];
        if (rank == 0) {
            ThrowIfFailed(rt_dist_get_unique_id(id));
            if (!write_all(id_out, id, sizeof id)) return 4;
        } else if (!read_all(id_in, id, sizeof id)) return 4;
        rt_dist *dist = nullptr;
        ThrowIfFailed(rt_dist_create(context->getHandle(), rank, world, id, &dist));

        auto scene = RtScene::create();
        scene->addModel(RtModel::create(context, argv[1]), Matrix::identity());
        RaytracingPipeline::Material material{};                       // DXRExperimentsApp.cpp:95-104
        material.params.albedo = {0.95f, 0.05f, 0.0f, 1.0f};
        material.params.specular = {0.58f, 0.58f, 0.58f, 1.0f};
        material.params.roughness = 0.5f;
        material.params.reflectivity = 0.7f;
        material.params.type = 1;
        auto camera = std::make_shared<Math::Camera>();
        camera->SetAspectRatio(float(width) / float(height));
        camera->SetEyeAtUp({0.0f, 0.0f, 3.2f}, {0.0f, 0.0f, 0.0f}, {0, 1, 0});
        auto pipeline = ProgressiveRaytracingPipeline::create(context, 1234);        // every rank: the same host RNG seed
        pipeline->setScene(scene);
        pipeline->addMaterial(material);
        pipeline->setCamera(camera);
        pipeline->loadResources(3);
        pipeline->setEnvironmentConstant(0.5f, 0.5f, 0.5f);
        pipeline->createOutputResource(RT_FORMAT_R32G32B32A32_FLOAT, width, height);
        pipeline->buildAccelerationStructures();
        rt_pipeline *p = pipeline->getHandle();
        void *image_dev = pipeline->getOutputResource(0);

        if (!tiles) ThrowIfFailed(rt_pipeline_set_accumulation_mode(p, RT_ACCUM_SUM));
        if (!tiles) pipeline->setDeferredFrames(32);          // a headless accumulation: nobody looks at the image before the collective
        ThrowIfFailed(rt_context_synchronize(context->getHandle()));
        const auto t0 = std::chrono::steady_clock::now();
        uint32_t mine = 0;
        if (tiles) {
            // every rank advances the SAME host state (jitter RNG, frame / accumulation counters) for every frame and renders its
            // bands of each -- 16 frames at a time through shared sets of launches (round 4: at 8 ranks a band set of ONE frame is an
            // eighth of a frame per persistent launch; sets give the launches back their length)
            const UINT per_set = 16;
            for (UINT first = 1; first <= frames; first += per_set)
                pipeline->renderBandsBatch(0.0f, first, first + per_set - 1 <= frames ? per_set : frames - first + 1, width, height, band_rows, rank, world);
        } else {
            for (UINT frame = 1; frame <= frames; ++frame) {
                // every rank advances the SAME host state for every frame and renders its share of the frames (the pipeline records them
                // and renders them in sets of up to 32: deferred mode, opted into above; the collective below flushes the last set)
                pipeline->update(0.0f, frame, (frame + 2) % 3, frame % 3, width, height);
                if ((frame - 1) % (UINT)world == (UINT)rank) {
                    pipeline->render(frame % 3, width, height);
                    mine++;
                }
            }
        }
        if (tiles) ThrowIfFailed(rt_dist_gather_bands(dist, image_dev, width, height, band_rows));
        else ThrowIfFailed(rt_dist_all_reduce_sum(dist, image_dev, size_t(width) * height * 4));
        ThrowIfFailed(rt_context_synchronize(context->getHandle()));
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        {   // one line per rank: which device it really ran on and what the collective cost there
            char bus[64] = "?";
            float coll_ms = 0.0f;
            (void)rt_dist_device_pci_bus_id(context->getHandle(), bus, sizeof bus);
            ThrowIfFailed(rt_dist_last_collective_ms(dist, &coll_ms));
            std::fprintf(stderr, "rank %d of %d: device %d (PCI %s), %.3f s, collective %.3f ms\n", rank, world, context->getDevice(), bus, s, coll_ms);
        }
        if (!tiles) {
            uint32_t expect = 0;
            ThrowIfFailed(rt_shard_frame_count(rank, world, frames, &expect));
            if (expect != mine) { std::fprintf(stderr, "rank %d rendered %u frames, expected %u\n", rank, mine, expect); return 5; }
        }
        // every rank's traced rays, summed over the ranks by the same collective (a three-float device buffer)
        double rays_all = 0.0;
        {
            rt_stats st;
            ThrowIfFailed(rt_pipeline_get_totals(p, &st));
            float mine_rays[4] = {float(double(st.rays_primary) / 1e6), float(double(st.rays_secondary) / 1e6), float((double(st.rays_shadow) - double(st.rays_shadow_skipped)) / 1e6), 0.0f};
            void *d = nullptr;
            ThrowIfFailed(rt_device_alloc(context->getHandle(), sizeof mine_rays, &d));
            ThrowIfFailed(rt_device_upload(context->getHandle(), d, mine_rays, sizeof mine_rays));
            ThrowIfFailed(rt_dist_all_reduce_sum(dist, d, 4));
            ThrowIfFailed(rt_device_download(context->getHandle(), mine_rays, d, sizeof mine_rays));
            ThrowIfFailed(rt_device_free(context->getHandle(), d));
            rays_all = (double(mine_rays[0]) + mine_rays[1] + mine_rays[2]) * 1e6;
        }
        if (rank == 0) {
            // the line bench.py --total-frames prints for the same run (BASELINE configs[2]: a fixed total, the collective inside the time)
            std::printf("{\"metric\": \"Mrays/s (all traced rays), progressive, %u frames IN ALL over the ranks + one %s\", \"value\": %.3f, \"unit\": \"Mrays/s\", "
                        "\"n_gpus\": %d, \"steps\": %u, \"ms_per_step\": %.6f, \"higher_is_better\": true, \"scaling\": \"strong\", \"total_frames\": %u, "
                        "\"frames_per_s\": %.3f, \"config\": {\"workload\": \"%s, %ux%u\", \"parallelism\": \"%s x%d\"}}\n",
                        frames, tiles ? "all-gather" : "all-reduce", rays_all / s / 1e6, world, frames, s / frames * 1e3, frames, frames / s, argv[1], width, height,
                        tiles ? "tile bands" : "sample shards", world);
            std::vector<float> image(size_t(width) * height * 4);
            pipeline->readOutput(image.data(), image.size() * sizeof(float));
            if (!tiles) for (float &v : image) v /= float(frames);     // SUM of all ranks' frames -> mean
            std::printf("%s on %d GPU(s), %s: %u frames in %.3f s = %.2f fps, ~%.2f Million Primary Rays/s\n", pipeline->getName(), world,
                        tiles ? "tile bands + all-gather" : "sample shards + all-reduce", frames, s, frames / s, double(width) * height * frames / s / 1e6);
            const std::string out = argv[5];
            const bool png = out.size() > 4 && out.compare(out.size() - 4, 4, ".png") == 0;
            ThrowIfFailed(png ? rt_image_write_png(out.c_str(), image.data(), width, height, 1.0f, 2.2f, 1)
                              : rt_image_write_pfm(out.c_str(), image.data(), width, height));
        }
        ThrowIfFailed(rt_dist_destroy(dist));
    } catch (const std::exception &e) {
        std::fprintf(stderr, "rank %d: error: %s\n", rank, e.what());
        return 1;
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc < 7) {
        std::fprintf(stderr, "usage: %s model.obj width height frames out.pfm|out.png gpus [samples|tiles]\n", argv[0]);
        return 2;
    }
    const int world = std::atoi(argv[6]);
    if (world < 1 || world > 64) { std::fprintf(stderr, "gpus must be 1..64\n"); return 2; }
    // pipes first, processes second, GPU last: the launcher itself never initialises HIP
    std::vector<int> to_child(2 * (size_t)world);
    int from_zero[2];
    if (pipe(from_zero) != 0) return 6;
    for (int r = 0; r < world; r++) if (pipe(&to_child[2 * r]) != 0) return 6;
    std::vector<pid_t> pids;
    for (int r = 0; r < world; r++) {
        const pid_t pid = fork();
        if (pid < 0) return 6;
        if (pid == 0) {
            for (int k = 0; k < world; k++) { close(to_child[2 * k + 1]); if (k != r) close(to_child[2 * k]); }
            close(from_zero[0]);
            if (r != 0) close(from_zero[1]);           // only rank 0 writes the id: if it dies first, the launcher must see EOF
            const int rc = rank_main(r, world, to_child[2 * r], from_zero[1], argv);
            std::fflush(stdout);
            _exit(rc);
        }
        pids.push_back(pid);
    }
    close(from_zero[1]);
    for (int r = 0; r < world; r++) close(to_child[2 * r]);
    char id[128];
    bool ok = read_all(from_zero[0], id, sizeof id);
    for (int r = 1; r < world && ok; r++) ok = write_all(to_child[2 * r + 1], id, sizeof id);
    // whether or not the id went round, the write ends are closed now: a rank still waiting for its id reads EOF and exits
    // instead of blocking for ever (rank 0 died before it made the id: no GPU, no RCCL, model file missing ...)
    for (int r = 0; r < world; r++) close(to_child[2 * r + 1]);
    close(from_zero[0]);
    int worst = ok ? 0 : 7;
    for (pid_t pid : pids) {
        int status = 0;
        waitpid(pid, &status, 0);
        const int rc = WIFEXITED(status) ? WEXITSTATUS(status) : 128 + WTERMSIG(status);
        if (rc != 0) worst = rc;
    }
    return worst;
}
