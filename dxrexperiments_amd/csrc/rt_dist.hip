// rt_dist.hip -- multi-GPU from the C ABI.
//
// The reference is a single-GPU application (src/DXRExperimentsApp.cpp:107-130 drives one device); the path shards
// without any exchange until the end, so a multi-GPU caller is N copies of that application -- ONE PROCESS PER GPU,
// scene and acceleration structures replicated -- plus exactly one collective:
//   A. sample batches (BASELINE config 3): rank r renders frames {f : f mod R = r} into an fp32 SUM buffer
//      (RT_ACCUM_SUM); rt_dist_all_reduce_sum adds the buffers over RCCL / xGMI; mean = sum / total frames.
//   B. image tiles (config 5): rank r renders the interleaved row bands {b : b mod R = r} (rt_tile_bands, pixels are
//      seeded by their global index so a tile equals the same pixels of the whole frame); rt_dist_gather_bands packs the
//      rank's bands, ncclAllGather's them (each rank receives (R-1)/R of ONE image: half the bytes of the SUM
//      all-reduce round 1 used as a gather) and scatters every band to its place.
// RCCL is used directly (ncclCommInitRank / ncclAllReduce / ncclAllGather on the context's stream).  librccl.so is 570 MB,
// so it is opened on first use instead of being a load-time dependency of every single-GPU caller.
//
// Launch rule for callers: create the processes BEFORE any of them touches the GPU (examples/progressive_multi.cpp forks
// first); never re-exec a process that has initialised HIP.
#include <dirent.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <new>

#include "rt_internal.h"
#include "rt_rccl_abi.h"

namespace {

// the few RCCL entry points used: rt_rccl_abi.h, checked against <rccl/rccl.h> at compile time by rt_rccl_abi_check.cpp
using namespace rt_rccl;

struct Rccl {
    void *so = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_all_reduce all_reduce = nullptr;
    fn_all_gather all_gather = nullptr;
    fn_error_string error_string = nullptr;
};

int load_rccl(Rccl **out)
{
    static Rccl r;
    if (!r.so) {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) if ((r.so = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
        if (!r.so) { rt_set_error("rt_dist: cannot open librccl.so: %s", dlerror()); return RT_ERR_UNSUPPORTED; }
        r.get_unique_id = (fn_get_unique_id)dlsym(r.so, "ncclGetUniqueId");
        r.comm_init_rank = (fn_comm_init_rank)dlsym(r.so, "ncclCommInitRank");
        r.comm_destroy = (fn_comm_destroy)dlsym(r.so, "ncclCommDestroy");
        r.all_reduce = (fn_all_reduce)dlsym(r.so, "ncclAllReduce");
        r.all_gather = (fn_all_gather)dlsym(r.so, "ncclAllGather");
        r.error_string = (fn_error_string)dlsym(r.so, "ncclGetErrorString");
        if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce || !r.all_gather) {
            rt_set_error("rt_dist: librccl.so lacks an expected entry point");
            dlclose(r.so);
            r.so = nullptr;
            return RT_ERR_UNSUPPORTED;
        }
    }
    *out = &r;
    return RT_OK;
}

#define NCCL_TRY(lib, expr)                                                                                     \
    do {                                                                                                        \
        const int rc_ = (expr);                                                                                 \
        if (rc_ != 0) {                                                                                         \
            rt_set_error("%s: %s", #expr, (lib)->error_string ? (lib)->error_string(rc_) : "RCCL error");     \
            return RT_ERR_HIP;                                                                                  \
        }                                                                                                       \
    } while (0)

// band b of the image = rows [b * band_rows, ...): owner b % world, slot b / world in the owner's packed chunk
__global__ void __launch_bounds__(256) k_pack_bands(const float4 *__restrict__ image, float4 *__restrict__ chunk, uint32_t width, uint32_t height,
                                                    uint32_t band_rows, uint32_t world, uint32_t rank)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)width * height) return;
    const uint32_t y = (uint32_t)(i / width), x = (uint32_t)(i % width);
    const uint32_t b = y / band_rows;
    if (b % world != rank) return;
    chunk[((size_t)(b / world) * band_rows + y % band_rows) * width + x] = image[i];
}
__global__ void __launch_bounds__(256) k_unpack_bands(float4 *__restrict__ image, const float4 *__restrict__ gathered, uint32_t width, uint32_t height,
                                                      uint32_t band_rows, uint32_t world, size_t chunk_pixels)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)width * height) return;
    const uint32_t y = (uint32_t)(i / width), x = (uint32_t)(i % width);
    const uint32_t b = y / band_rows;
    image[i] = gathered[(size_t)(b % world) * chunk_pixels + ((size_t)(b / world) * band_rows + y % band_rows) * width + x];
}

// Two ranks on ONE device make ncclCommInitRank fail late or hang (RCCL wants one device per rank).  Before the communicator
// exists the ranks share nothing but the 128-byte id the launcher handed round, so they meet in /dev/shm (one node: the
// scope of this engine's multi-GPU mode): every rank publishes its device's PCI bus id in a file named after the id and its
// rank, then reads the others'.  A rank whose file does not appear within the time limit is not on this node (or not yet
// there): the check is skipped for it, nothing fails.  RT_DIST_CHECK_SECONDS=0 turns the rendezvous off.
std::string rendezvous_path(const void *id128, int rank)
{
    unsigned long long h = 1469598103934665603ull;                       // FNV-1a of the id
    for (int i = 0; i < 128; i++) { h ^= ((const unsigned char *)id128)[i]; h *= 1099511628211ull; }
    char name[96];
    snprintf(name, sizeof name, "/dev/shm/dxr_amd_%016llx_%d", h, rank);
    return name;
}

int check_one_device_per_rank(const rt_context *ctx, int rank, int world, const void *id128, std::string *mine)
{
    // how long a rank waits for the others' entries: ranks of one launcher arrive within milliseconds of each other, a rank on
    // another node never does -- so a few seconds by default (rt_debug_set_option "dist_check_seconds"; a rank that is not seen in time is simply
    // not checked), and no wait at all beyond the ranks the launcher says are local (LOCAL_WORLD_SIZE)
    const double limit = ctx->opt_dist_check_seconds;
    if (world < 2 || !(limit > 0.0)) return RT_OK;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, ctx->device) != hipSuccess) return RT_OK;       // nothing to compare
    {   // entries a failed create of THIS user left behind (see below) are swept once they are ten minutes old
        if (DIR *dir = opendir("/dev/shm")) {
            const time_t now = time(nullptr);
            const uid_t me = geteuid();
            while (struct dirent *e = readdir(dir)) {
                if (strncmp(e->d_name, "dxr_amd_", 8) != 0) continue;
                const std::string old = std::string("/dev/shm/") + e->d_name;
                struct stat sb;
                if (lstat(old.c_str(), &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_uid == me && now - sb.st_mtime > 600) unlink(old.c_str());
            }
            closedir(dir);
        }
    }
    const std::string path = rendezvous_path(id128, rank), tmp = path + ".tmp";
    // (a world-writable directory: the entry is created exclusively, never through a link somebody else has put there, and
    // readable by its owner only -- the ranks of one job run as one user)
    (void)unlink(tmp.c_str());
    const int fd = open(tmp.c_str(), O_CREAT | O_EXCL | O_NOFOLLOW | O_WRONLY | O_CLOEXEC, 0600);
    if (fd < 0) return RT_OK;                                                                      // no /dev/shm: skip
    const size_t len = strlen(bus);
    const bool wrote = write(fd, bus, len) == (ssize_t)len && write(fd, "\n", 1) == 1;
    close(fd);
    if (!wrote || rename(tmp.c_str(), path.c_str()) != 0) { unlink(tmp.c_str()); return RT_OK; }
    *mine = path;
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    // the ranks of this node, if the launcher says which (torchrun: LOCAL_WORLD_SIZE consecutive ranks per node)
    int first = 0, last = world;
    if (const char *e = getenv("LOCAL_WORLD_SIZE")) {
        const int lw = atoi(e);
        if (lw >= 1 && lw <= world) { first = rank / lw * lw; last = first + lw < world ? first + lw : world; }
    }
    for (int r = first; r < last; r++) {
        if (r == rank) continue;
        const std::string other = rendezvous_path(id128, r);
        for (;;) {
            FILE *g = fopen(other.c_str(), "r");
            if (g) {
                char theirs[64] = {0};
                const bool got = fgets(theirs, sizeof theirs, g) != nullptr;
                fclose(g);
                if (got) {
                    theirs[strcspn(theirs, "\n")] = 0;
                    if (strcmp(theirs, bus) == 0) {
                        rt_set_error("rt_dist_create: ranks %d and %d are both on the device at PCI %s; RCCL needs one device per rank", rank < r ? rank : r,
                                     rank < r ? r : rank, bus);
                        return RT_ERR_INVALID_ARG;
                    }
                    break;
                }
            }
            struct timespec t;
            clock_gettime(CLOCK_MONOTONIC, &t);
            if ((double)(t.tv_sec - t0.tv_sec) + 1e-9 * (double)(t.tv_nsec - t0.tv_nsec) > limit) break;       // not on this node (yet): skip
            usleep(2000);
        }
    }
    return RT_OK;
}

}  // namespace

struct rt_dist {
    rt_context *ctx = nullptr;
    Rccl *lib = nullptr;
    nccl_comm comm = nullptr;
    int rank = 0, world = 1;
    DevBuf gathered;
    std::string rendezvous_file;       // this rank's entry of the one-device-per-rank check (removed on destroy)
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;                // an event pair brackets the last collective
};

extern "C" {

// ---- host-side partition logic (no device, no RCCL): shared by the C++ example, bench.py and the CPU tests ----

int rt_shard_frame_count(uint32_t rank, uint32_t world, uint32_t n_frames, uint32_t *count)
{
    RT_REQUIRE(count && world > 0 && rank < world, "bad argument");
    *count = n_frames > rank ? (n_frames - rank + world - 1) / world : 0u;      // |{f < n_frames : f mod world == rank}|
    return RT_OK;
}

int rt_tile_bands(uint32_t height, uint32_t band_rows, uint32_t rank, uint32_t world, uint32_t *y0, uint32_t *y1, uint32_t capacity,
                  uint32_t *n_bands)
{
    RT_REQUIRE(n_bands && world > 0 && rank < world && band_rows > 0, "bad argument");
    uint32_t n = 0;
    for (uint32_t b = rank; (uint64_t)b * band_rows < height; b += world) {
        if (y0 && y1) {
            RT_REQUIRE(n < capacity, "band arrays too small");
            y0[n] = b * band_rows;
            y1[n] = (uint64_t)(b + 1) * band_rows < height ? (b + 1) * band_rows : height;
        }
        n++;
    }
    *n_bands = n;
    return RT_OK;
}

int rt_tile_gather_layout(uint32_t width, uint32_t height, uint32_t band_rows, uint32_t world, uint32_t *slots_per_rank, size_t *floats_per_rank)
{
    RT_REQUIRE(world > 0 && band_rows > 0 && width > 0 && height > 0, "bad argument");
    const uint32_t bands = (height + band_rows - 1) / band_rows;
    const uint32_t slots = (bands + world - 1) / world;              // every rank sends the same count: short ranks pad
    if (slots_per_rank) *slots_per_rank = slots;
    if (floats_per_rank) *floats_per_rank = (size_t)slots * band_rows * width * 4;
    return RT_OK;
}

// ---- RCCL ------------------------------------------------------------------------------------------------------

int rt_dist_get_unique_id(void *id128)
{
    RT_REQUIRE(id128, "null argument");
    Rccl *lib;
    RT_TRY(load_rccl(&lib));
    nccl_id id;
    NCCL_TRY(lib, lib->get_unique_id(&id));
    memcpy(id128, &id, sizeof id);
    return RT_OK;
}

int rt_dist_create(rt_context *ctx, int rank, int world, const void *id128, rt_dist **out)
{
    RT_REQUIRE(ctx && id128 && out, "null argument");
    RT_REQUIRE(world >= 1 && rank >= 0 && rank < world, "rank outside [0, world)");
    Rccl *lib;
    RT_TRY(load_rccl(&lib));
    HIP_TRY(hipSetDevice(ctx->device));
    rt_dist *d = new (std::nothrow) rt_dist();
    if (!d) { rt_set_error("out of host memory"); return RT_ERR_OOM; }
    d->ctx = ctx; d->lib = lib; d->rank = rank; d->world = world;
    {
        const int chk = check_one_device_per_rank(ctx, rank, world, id128, &d->rendezvous_file);
        if (chk != RT_OK) {
            // (this rank's entry stays: the other rank on the same device has to find it to fail the same way instead of
            // waiting for RCCL; the next rendezvous on this node sweeps it)
            delete d;
            return chk;
        }
    }
    nccl_id id;
    memcpy(&id, id128, sizeof id);
    const int rc = lib->comm_init_rank(&d->comm, world, id, rank);
    if (rc != 0) {
        rt_set_error("ncclCommInitRank(rank %d of %d): %s", rank, world, lib->error_string ? lib->error_string(rc) : "RCCL error");
        if (!d->rendezvous_file.empty()) unlink(d->rendezvous_file.c_str());
        delete d;
        return RT_ERR_HIP;
    }
    if (hipEventCreate(&d->ev0) != hipSuccess || hipEventCreate(&d->ev1) != hipSuccess) { d->ev0 = nullptr; d->ev1 = nullptr; }
    rt_context_retain(ctx);
    *out = d;
    return RT_OK;
}

int rt_dist_destroy(rt_dist *d)
{
    if (!d) return RT_OK;
    (void)hipSetDevice(d->ctx->device);
    (void)hipStreamSynchronize(d->ctx->stream);
    if (d->comm) (void)d->lib->comm_destroy(d->comm);
    if (!d->rendezvous_file.empty()) unlink(d->rendezvous_file.c_str());
    if (d->ev0) (void)hipEventDestroy(d->ev0);
    if (d->ev1) (void)hipEventDestroy(d->ev1);
    d->gathered.release();
    rt_context *ctx = d->ctx;
    delete d;
    rt_context_release(ctx);
    return RT_OK;
}

int rt_dist_get_rank(const rt_dist *d, int *rank, int *world)
{
    RT_REQUIRE(d && rank && world, "null argument");
    *rank = d->rank; *world = d->world;
    return RT_OK;
}

int rt_dist_all_reduce_sum(rt_dist *d, void *device_f32, size_t count)
{
    RT_REQUIRE(d && device_f32, "null argument");
    RT_TRY(rt_context_flush_deferred(d->ctx));        // frames a deferred pipeline still holds belong to the image that is exchanged
    HIP_TRY(hipSetDevice(d->ctx->device));
    d->timed = false;
    if (d->ev0) HIP_TRY(hipEventRecord(d->ev0, d->ctx->stream));
    NCCL_TRY(d->lib, d->lib->all_reduce(device_f32, device_f32, count, NCCL_FLOAT, NCCL_SUM, d->comm, d->ctx->stream));
    if (d->ev1) { HIP_TRY(hipEventRecord(d->ev1, d->ctx->stream)); d->timed = true; }
    return RT_OK;
}

int rt_dist_last_collective_ms(rt_dist *d, float *ms)
{
    RT_REQUIRE(d && ms, "null argument");
    *ms = 0.0f;
    if (!d->timed) { rt_set_error("rt_dist_last_collective_ms: no collective has been issued"); return RT_ERR_STATE; }
    HIP_TRY(hipSetDevice(d->ctx->device));
    HIP_TRY(hipEventSynchronize(d->ev1));
    HIP_TRY(hipEventElapsedTime(ms, d->ev0, d->ev1));
    return RT_OK;
}

int rt_dist_device_pci_bus_id(const rt_context *ctx, char *out, size_t capacity)
{
    RT_REQUIRE(ctx && out && capacity >= 16, "bad argument");
    HIP_TRY(hipDeviceGetPCIBusId(out, (int)capacity, ctx->device));
    return RT_OK;
}

int rt_dist_gather_bands(rt_dist *d, void *device_rgba32f, uint32_t width, uint32_t height, uint32_t band_rows)
{
    RT_REQUIRE(d && device_rgba32f, "null argument");
    RT_TRY(rt_context_flush_deferred(d->ctx));        // frames a deferred pipeline still holds belong to the image that is exchanged
    uint32_t slots = 0;
    size_t floats = 0;
    RT_TRY(rt_tile_gather_layout(width, height, band_rows, (uint32_t)d->world, &slots, &floats));
    HIP_TRY(hipSetDevice(d->ctx->device));
    hipStream_t st = d->ctx->stream;
    const size_t chunk_pixels = floats / 4;
    d->timed = false;
    if (d->ev0) HIP_TRY(hipEventRecord(d->ev0, st));
    RT_TRY(d->gathered.reserve(floats * 4 * (size_t)d->world));
    float4 *all = d->gathered.as<float4>();
    const unsigned grid = (unsigned)(((size_t)width * height + 255) / 256);
    // pack straight into this rank's slice of the receive buffer: the in-place form of ncclAllGather
    k_pack_bands<<<grid, 256, 0, st>>>((const float4 *)device_rgba32f, all + (size_t)d->rank * chunk_pixels, width, height, band_rows,
                                       (uint32_t)d->world, (uint32_t)d->rank);
    NCCL_TRY(d->lib, d->lib->all_gather(all + (size_t)d->rank * chunk_pixels, all, floats, NCCL_FLOAT, d->comm, st));
    k_unpack_bands<<<grid, 256, 0, st>>>((float4 *)device_rgba32f, all, width, height, band_rows, (uint32_t)d->world, chunk_pixels);
    HIP_TRY(hipGetLastError());
    if (d->ev1) { HIP_TRY(hipEventRecord(d->ev1, st)); d->timed = true; }
    return RT_OK;
}

}  // extern "C"
