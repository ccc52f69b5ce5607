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
#include <dlfcn.h>

#include <new>

#include "rt_internal.h"

namespace {

// the few RCCL entry points used, with the types of <rccl/rccl.h> (NCCL 2.x ABI)
struct nccl_id { char internal[128]; };
typedef void *nccl_comm;
typedef int (*fn_get_unique_id)(nccl_id *);
typedef int (*fn_comm_init_rank)(nccl_comm *, int, nccl_id, int);
typedef int (*fn_comm_destroy)(nccl_comm);
typedef int (*fn_all_reduce)(const void *, void *, size_t, int, int, nccl_comm, hipStream_t);
typedef int (*fn_all_gather)(const void *, void *, size_t, int, nccl_comm, hipStream_t);
typedef const char *(*fn_error_string)(int);
constexpr int NCCL_FLOAT = 7, NCCL_SUM = 0;

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

}  // namespace

struct rt_dist {
    rt_context *ctx = nullptr;
    Rccl *lib = nullptr;
    nccl_comm comm = nullptr;
    int rank = 0, world = 1;
    DevBuf gathered;
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
    nccl_id id;
    memcpy(&id, id128, sizeof id);
    const int rc = lib->comm_init_rank(&d->comm, world, id, rank);
    if (rc != 0) {
        rt_set_error("ncclCommInitRank(rank %d of %d): %s", rank, world, lib->error_string ? lib->error_string(rc) : "RCCL error");
        delete d;
        return RT_ERR_HIP;
    }
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
    HIP_TRY(hipSetDevice(d->ctx->device));
    NCCL_TRY(d->lib, d->lib->all_reduce(device_f32, device_f32, count, NCCL_FLOAT, NCCL_SUM, d->comm, d->ctx->stream));
    return RT_OK;
}

int rt_dist_gather_bands(rt_dist *d, void *device_rgba32f, uint32_t width, uint32_t height, uint32_t band_rows)
{
    RT_REQUIRE(d && device_rgba32f, "null argument");
    uint32_t slots = 0;
    size_t floats = 0;
    RT_TRY(rt_tile_gather_layout(width, height, band_rows, (uint32_t)d->world, &slots, &floats));
    HIP_TRY(hipSetDevice(d->ctx->device));
    hipStream_t st = d->ctx->stream;
    const size_t chunk_pixels = floats / 4;
    RT_TRY(d->gathered.reserve(floats * 4 * (size_t)d->world));
    float4 *all = d->gathered.as<float4>();
    const unsigned grid = (unsigned)(((size_t)width * height + 255) / 256);
    // pack straight into this rank's slice of the receive buffer: the in-place form of ncclAllGather
    k_pack_bands<<<grid, 256, 0, st>>>((const float4 *)device_rgba32f, all + (size_t)d->rank * chunk_pixels, width, height, band_rows,
                                       (uint32_t)d->world, (uint32_t)d->rank);
    NCCL_TRY(d->lib, d->lib->all_gather(all + (size_t)d->rank * chunk_pixels, all, floats, NCCL_FLOAT, d->comm, st));
    k_unpack_bands<<<grid, 256, 0, st>>>((float4 *)device_rgba32f, all, width, height, band_rows, (uint32_t)d->world, chunk_pixels);
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

}  // extern "C"
