// ndt_api.hip — the C ABI of include/ndt_mi355.h: handle, uploads, the Gauss-Newton driver loop.
// The host only queues kernels; the pose lives in a GnState in HBM for the whole align, and the host
// looks at it every SYNC_CHUNK iterations (one 8-byte-aligned D2H copy into pinned memory).
#include "ndt_internal.h"

#include <dlfcn.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>
#include <new>

#define SYNC_CHUNK 4

// RCCL is bound at run time (dlopen) so the library loads on hosts without it and never pulls a second
// copy into a process that already has one.
typedef int (*pfn_ncclGetUniqueId)(void*);
struct ndt_nccl_id { char internal[128]; };
typedef int (*pfn_ncclCommInitRank)(void**, int, ndt_nccl_id, int);
typedef int (*pfn_ncclAllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*pfn_ncclCommDestroy)(void*);
typedef int (*pfn_ncclCommCount)(const void*, int*);
static struct {
    void* lib;
    pfn_ncclGetUniqueId get_id;
    pfn_ncclCommInitRank init_rank;
    pfn_ncclAllReduce all_reduce;
    pfn_ncclCommDestroy destroy;
    pfn_ncclCommCount count;
} g_rccl;
static bool rccl_load() {
    static std::once_flag once;        // handles on several host threads may reach this together
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        void* lib = nullptr;
        for (const char* nm : names) {
            lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) return;
        g_rccl.get_id = (pfn_ncclGetUniqueId)dlsym(lib, "ncclGetUniqueId");
        g_rccl.init_rank = (pfn_ncclCommInitRank)dlsym(lib, "ncclCommInitRank");
        g_rccl.all_reduce = (pfn_ncclAllReduce)dlsym(lib, "ncclAllReduce");
        g_rccl.destroy = (pfn_ncclCommDestroy)dlsym(lib, "ncclCommDestroy");
        g_rccl.count = (pfn_ncclCommCount)dlsym(lib, "ncclCommCount");
        g_rccl.lib = lib;
    });
    return g_rccl.lib && g_rccl.get_id && g_rccl.init_rank && g_rccl.all_reduce && g_rccl.destroy;
}
enum { NCCL_FLOAT64 = 8, NCCL_SUM = 0 };

struct ndt_handle {
    ndt_config cfg;
    int n_cu = 256;              // compute units of cfg.device: every grid of this handle is sized from it
    hipStream_t stream = nullptr;
    hipStream_t lane_stream[3] = {nullptr, nullptr, nullptr};   // further launch lanes of a replay batch (independent pairs: no ordering between lanes)
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    int batch_lanes = 2;                       // NDT_BATCH_LANES=1..4 (A/B switch)
    VoxelMap map;
    MapWork work;
    MapBuildCfg build_cfg;      // parameters the resident map was built with
    bool have_target = false;
    DevBuf sx, sy, sz, sraw;
    size_t n_src = 0;
    bool have_source = false;
    DevBuf partials, partials2, cell_keys, cell_mask, pcl_partials;
    GnState* d_state = nullptr;
    GnState* d_state2 = nullptr;               // the chained loop ping-pongs between the two (GnBuffers below)
    unsigned char* static_scratch = nullptr;   // this device's copy of g_static_scratch (ndt_align_static), resolved at ndt_create
    bool use_chain = true;                     // NDT_NO_CHAIN=1: two launches per iteration (A/B switch)
    GnState* h_state = nullptr;  // pinned
    const unsigned char* src_aos = nullptr; size_t src_aos_stride = 0;   // ndt_set_source_async: records not yet unpacked into sx / sy / sz
    unsigned long long mirror_seq = 0;      // count of results requested through the mirror (the kernel echoes it after the state)
    GnState* h_state_dev = nullptr;   // the same memory as the device addresses it (the last kernel of a chained loop writes the result there)
    std::string err;
    int last_radix_passes = 0;
    int use_links = 1;           // NEARBY7 through the slots' neighbour links (NDT_NO_LINKS=1: seven hashed probes)
    // profiling
    bool prof = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    double prof_us = 0.0;
    int64_t prof_launches = 0;
    double prof_hits = 0.0, prof_pts = 0.0;
    // multi-GPU
    void* comm = nullptr;
    int n_ranks = 1;
    DevBuf comm_rows;            // [2][ACC_N] doubles: all-reduce send / receive rows
    // peer exchange (ndt_comm_p2p_*): the sharded loop's all-reduce as direct stores into the ranks' mailboxes
    struct PeerExchange {
        int n = 0, rank = 0;                              // n > 0: attached
        unsigned char* local = nullptr;                   // this rank's mailbox (PX_BYTES, fine-grained device memory)
        unsigned char* peer[NDT_MAX_PEERS] = {};          // every rank's mailbox as this process addresses it (peer[rank] == local)
        bool opened[NDT_MAX_PEERS] = {};                  // peer[i] came from hipIpcOpenMemHandle
        unsigned long long seq = 0;                       // all-reduces queued so far
        bool shared_device = false;                       // another rank runs on this rank's GPU (rehearsals): see gn_loop
    } px;
    // offline replay batch
    DevBuf braw, bx, by, bz, boff, bpair, bstates, bpartials, bdone;
    size_t b_group = 0;                        // pairs per launch (one lane) of the resident batch
    std::vector<int64_t> b_offsets; int64_t b_max_pts = 0;
    size_t b_pairs = 0; int b_rows = 0; bool b_has_pair_scan = false; double b_points = 0;   // geometry of the resident batch
};

#define H_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { h->err = std::string(#x) + ": " + hipGetErrorString(e_); return NDT_ERR_HIP; } } while (0)
static int ensure_source_soa(ndt_handle* h);

// true when p is device memory of device `dev` (a plain host pointer makes the query fail: that is not an error)
static bool points_on_device(const void* p, int dev) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return at.type == hipMemoryTypeDevice && at.device == dev;
}
static KeyParams key_params(const ndt_config& c) {
    KeyParams kp;
    kp.inv_leaf_f32 = 1.0f / c.resolution;
    kp.inv_res_d = (double)kp.inv_leaf_f32;
    kp.res_d = (double)c.resolution;
    kp.key_rule = c.key_rule;
    return kp;
}
static int check_cfg(const ndt_config* c, std::string& err) {
    if (!c) { err = "null config"; return NDT_ERR_INVALID_ARG; }
    if (!(c->resolution > 0.f) || !isfinite(c->resolution)) { err = "resolution must be positive"; return NDT_ERR_INVALID_ARG; }
    if (c->nearby != 1 && c->nearby != 7) { err = "nearby must be 1 or 7"; return NDT_ERR_INVALID_ARG; }
    if (c->max_iterations < 1) { err = "max_iterations must be >= 1"; return NDT_ERR_INVALID_ARG; }
    if (c->key_rule != NDT_KEY_FLOOR_F32 && c->key_rule != NDT_KEY_ROUND_F64) { err = "unknown key_rule"; return NDT_ERR_INVALID_ARG; }
    if (c->min_pts_in_voxel < 1) { err = "min_pts_in_voxel must be >= 1"; return NDT_ERR_INVALID_ARG; }
    if (!(c->eig_floor_ratio > 0.0) || c->eig_floor_ratio > 1.0) { err = "eig_floor_ratio must be in (0,1]"; return NDT_ERR_INVALID_ARG; }
    if (c->max_pts_in_voxel < 0) { err = "max_pts_in_voxel must be >= 0"; return NDT_ERR_INVALID_ARG; }
    if (c->info_rule != NDT_INFO_EIG_CLAMP && c->info_rule != NDT_INFO_ADD_1E3) { err = "unknown info_rule"; return NDT_ERR_INVALID_ARG; }
    if (c->capacity < 0 || c->capacity == 1) { err = "capacity must be 0 (unbounded) or >= 2 (capacity - 1 voxels are kept)"; return NDT_ERR_INVALID_ARG; }
    return NDT_OK;
}

// ---- ndt_align_static: the one entry point that allocates nothing (include/ndt_mi355.h).  Its Gauss-Newton loop keeps the
// per-block partial rows (two buffers, the chained loop ping-pongs) and the two GnStates in this file-scope device array
// instead of the handle's hipMalloc'd buffers.  The name below is a host-side stand-in, NOT a device address: every use
// of the memory goes through the per-device pointer that hipGetSymbolAddress returns (each GPU holds its own copy).
#define STATIC_MAX_ROWS 256                                        // one row per block, one block per CU on MI355X
#define STATIC_ROWS_BYTES ((size_t)STATIC_MAX_ROWS * ACC_N * sizeof(double))   // 65 536
#define STATIC_STATE_BYTES 512                                     // sizeof(GnState) = 424, padded to the array's alignment
#define STATIC_SCRATCH_BYTES (2 * STATIC_ROWS_BYTES + 2 * STATIC_STATE_BYTES)  // 132 096 bytes per GPU, held until the process ends
static_assert(sizeof(GnState) <= STATIC_STATE_BYTES, "GnState must fit its scratch cell");
static __device__ __align__(256) unsigned char g_static_scratch[STATIC_SCRATCH_BYTES];

#define NDT_MAX_DEVICES 64
static std::mutex g_scratch_mu;
static unsigned char* g_scratch_ptr[NDT_MAX_DEVICES];              // per device: resolved once, with that device current
static std::mutex g_static_call_mu[NDT_MAX_DEVICES];              // per device: one ndt_align_static at a time owns that GPU's array
// device `dev` must be current
static hipError_t static_scratch_for_device(int dev, unsigned char** out) {
    if (dev < 0 || dev >= NDT_MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_static_scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_scratch_ptr[dev] = (unsigned char*)p;
    }
    *out = g_scratch_ptr[dev];
    return hipSuccess;
}

// where one Gauss-Newton loop keeps its device-side working state: rows[2] are partial-row buffers (>= grid rows each),
// st[2] the two states the chained loop ping-pongs between
struct GnBuffers { double* rows[2]; GnState* st[2]; };
struct SourceView { const float *x, *y, *z; size_t n; AosIn aos; };

extern "C" {

void ndt_default_config(ndt_config* c) {
    if (!c) return;
    c->resolution = 1.0f; c->max_iterations = 35; c->eps = 1e-2; c->nearby = 7; c->min_pts_in_voxel = 6;
    c->res_outlier_th = 20.0; c->min_effective_pts = 10; c->key_rule = NDT_KEY_FLOOR_F32; c->eig_floor_ratio = 0.01;
    c->final_eval = 1; c->device = 0;
    c->max_pts_in_voxel = 0; c->info_rule = NDT_INFO_EIG_CLAMP; c->capacity = 0; c->reserved0 = 0;
}

const char* ndt_version(void) { return "ndt_mi355 0.1 (gfx950)"; }

ndt_handle* ndt_create(const ndt_config* cfg) {
    std::string err;
    if (check_cfg(cfg, err) != NDT_OK) return nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || cfg->device < 0 || cfg->device >= ndev) return nullptr;
    if (hipSetDevice(cfg->device) != hipSuccess) return nullptr;
    int n_cu = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, cfg->device) != hipSuccess || n_cu <= 0) n_cu = 256;
    ndt_handle* h = new (std::nothrow) ndt_handle();
    if (!h) return nullptr;
    h->cfg = *cfg;
    h->n_cu = n_cu;
    h->use_links = getenv("NDT_NO_LINKS") ? 0 : 1;
    h->use_chain = getenv("NDT_NO_CHAIN") == nullptr;
    if (const char* e = getenv("NDT_BATCH_LANES")) h->batch_lanes = std::min(4, std::max(1, atoi(e)));
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&h->lane_stream[0], hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&h->lane_stream[1], hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&h->lane_stream[2], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join[1], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&h->ev_join[2], hipEventDisableTiming) != hipSuccess ||
        hipMalloc((void**)&h->d_state, sizeof(GnState)) != hipSuccess ||
        hipMalloc((void**)&h->d_state2, sizeof(GnState)) != hipSuccess ||
        hipHostMalloc((void**)&h->h_state, sizeof(GnState) + 64, hipHostMallocDefault) != hipSuccess ||
        static_scratch_for_device(cfg->device, &h->static_scratch) != hipSuccess) {
        ndt_destroy(h);
        return nullptr;
    }
    // hipHostMalloc does not clear: the sequence word after the state must not hold a stale number from a destroyed handle's
    // page (mirror_seq restarts at 1 with every handle)
    memset(h->h_state, 0, sizeof(GnState) + 64);
    // optional (NDT_NO_MIRROR=1 switches it off): without it the loop copies the state back as before
    if (getenv("NDT_NO_MIRROR") || hipHostGetDevicePointer((void**)&h->h_state_dev, h->h_state, 0) != hipSuccess) h->h_state_dev = nullptr;
    (void)hipGetLastError();
    return h;
}

static void px_release(ndt_handle* h);
void ndt_destroy(ndt_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (hipStream_t ls : h->lane_stream) if (ls) (void)hipStreamSynchronize(ls);
    if (h->comm && g_rccl.destroy) g_rccl.destroy(h->comm);
    px_release(h);
    map_free(h->map);
    h->work.release();
    DevBuf* bufs[] = {&h->sx, &h->sy, &h->sz, &h->sraw, &h->partials, &h->partials2, &h->cell_keys, &h->cell_mask, &h->pcl_partials, &h->comm_rows, &h->braw, &h->bx, &h->by, &h->bz,
                      &h->boff, &h->bpair, &h->bstates, &h->bpartials, &h->bdone};
    for (DevBuf* b : bufs) b->release();
    for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
    if (h->d_state) (void)hipFree(h->d_state);
    if (h->d_state2) (void)hipFree(h->d_state2);
    if (h->h_state) (void)hipHostFree(h->h_state);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    for (hipStream_t ls : h->lane_stream) if (ls) (void)hipStreamDestroy(ls);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    for (hipEvent_t e : h->ev_join) if (e) (void)hipEventDestroy(e);
    delete h;
}

int ndt_configure(ndt_handle* h, const ndt_config* cfg) {
    if (!h) return NDT_ERR_INVALID_ARG;
    int rc = check_cfg(cfg, h->err);
    if (rc != NDT_OK) return rc;
    if (cfg->device != h->cfg.device) { h->err = "device cannot change after ndt_create"; return NDT_ERR_INVALID_ARG; }
    h->cfg = *cfg;
    return NDT_OK;
}

// a map starts (or starts again): empty table, the build parameters of the configuration in force now
static hipError_t start_map(ndt_handle* h) {
    hipError_t e = map_clear(h->map, h->stream);
    if (e != hipSuccess) return e;
    h->build_cfg.kp = key_params(h->cfg);
    h->build_cfg.min_pts = h->cfg.min_pts_in_voxel;
    h->build_cfg.eig_floor_ratio = h->cfg.eig_floor_ratio;
    h->build_cfg.max_pts = h->cfg.max_pts_in_voxel; h->build_cfg.info_rule = h->cfg.info_rule;
    h->build_cfg.capacity = h->cfg.capacity;
    return map_track_touch(h->map, h->cfg.capacity > 0, h->stream);
}

static int merge_common(ndt_handle* h, const void* pts, size_t n, size_t stride, const double T[16], bool reset) {
    if (!h) return NDT_ERR_INVALID_ARG;
    if ((n && !pts) || stride < 12 || (stride & 3)) { h->err = "bad point buffer (need stride >= 12, multiple of 4)"; return NDT_ERR_INVALID_ARG; }
    H_TRY(hipSetDevice(h->cfg.device));
    if (reset || !h->have_target) H_TRY(start_map(h));
    double R[9], t[3];
    if (T) {
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) R[3 * r + c] = T[c * 4 + r]; t[r] = T[12 + r]; }
    }
    int rc = map_merge_points(h->map, h->work, h->build_cfg, pts, nullptr, nullptr, n, stride, T ? R : nullptr, T ? t : nullptr, h->stream,
                              &h->last_radix_passes, h->err);
    if (rc == NDT_OK) rc = map_lru_step(h->map, h->work, h->build_cfg, h->stream, h->err);
    if (rc == NDT_OK) h->have_target = true;
    return rc;
}

int ndt_set_target(ndt_handle* h, const void* pts, size_t n, size_t stride) { return merge_common(h, pts, n, stride, nullptr, true); }
int ndt_add_target(ndt_handle* h, const void* pts, size_t n, size_t stride, const double T[16]) {
    return merge_common(h, pts, n, stride, T, false);
}

static int set_source_impl(ndt_handle* h, const void* pts, size_t n, size_t stride, bool stream_ordered) {
    if (!h) return NDT_ERR_INVALID_ARG;
    if ((n && !pts) || stride < 12 || (stride & 3)) { h->err = "bad point buffer (need stride >= 12, multiple of 4)"; return NDT_ERR_INVALID_ARG; }
    if (n >= (1ull << 31) - (1ull << 21)) { h->err = "more than 2^31 - 2^21 source points (the kernels index points with 32-bit integers)"; return NDT_ERR_INVALID_ARG; }
    H_TRY(hipSetDevice(h->cfg.device));
    const bool on_device = n && points_on_device(pts, h->cfg.device);
    if (n && stream_ordered && !on_device) { h->err = "ndt_set_source_async takes records in this device's memory"; return NDT_ERR_INVALID_ARG; }
    // from here on the previous source is gone (its arrays are about to be overwritten, records of an earlier
    // ndt_set_source_async that nobody unpacked are dropped): a failure below leaves the handle without a source rather than
    // with a half-replaced one.  src_aos, n_src and have_source are committed together at the end.
    h->src_aos = nullptr; h->have_source = false; h->n_src = 0;
    H_TRY(h->sx.ensure(n * 4 + 16)); H_TRY(h->sy.ensure(n * 4 + 16)); H_TRY(h->sz.ensure(n * 4 + 16));
    H_TRY(h->partials.ensure((size_t)residual_grid(n, h->n_cu) * ACC_N * sizeof(double)));
    H_TRY(h->partials2.ensure((size_t)residual_grid(n, h->n_cu) * ACC_N * sizeof(double)));
    const unsigned char* pending = nullptr;
    if (n) {
        // records already in this device's memory are unpacked in place (the unpack kernel is their only reader and has
        // finished when this call returns); host records are staged through sraw
        const void* d_in = pts;
        if (!on_device) {
            H_TRY(h->sraw.ensure(n * stride + 16));
            H_TRY(hipMemcpyAsync(h->sraw.p, pts, n * stride, hipMemcpyDefault, h->stream));
            d_in = h->sraw.p;
        }
        if (stream_ordered) pending = (const unsigned char*)d_in;      // unpacked by the first launch that reads the source
        else H_TRY(unpack_points_soa(d_in, n, stride, h->sx.as<float>(), h->sy.as<float>(), h->sz.as<float>(), h->stream));
    }
    if (!stream_ordered) H_TRY(hipStreamSynchronize(h->stream));   // the caller's buffer is free again on return
    h->src_aos = pending; h->src_aos_stride = stride;
    h->n_src = n;
    h->have_source = true;
    return NDT_OK;
}
int ndt_set_source(ndt_handle* h, const void* pts, size_t n, size_t stride) { return set_source_impl(h, pts, n, stride, false); }
int ndt_set_source_async(ndt_handle* h, const void* d_pts, size_t n, size_t stride) { return set_source_impl(h, d_pts, n, stride, true); }

// ---- scan pre-processing on the device

int ndt_set_source_downsampled(ndt_handle* h, const void* pts, size_t n, size_t stride, float leaf, const double T[16]) {
    if (!h) return NDT_ERR_INVALID_ARG;
    if ((n && !pts) || stride < 12 || (stride & 3)) { h->err = "bad point buffer (need stride >= 12, multiple of 4)"; return NDT_ERR_INVALID_ARG; }
    if (!(leaf > 0.f) || !isfinite(leaf)) { h->err = "leaf size must be positive"; return NDT_ERR_INVALID_ARG; }
    if (n >= (1ull << 31) - (1ull << 21)) { h->err = "more than 2^31 - 2^21 points (the kernels index points with 32-bit integers)"; return NDT_ERR_INVALID_ARG; }
    H_TRY(hipSetDevice(h->cfg.device));
    // the source arrays are rewritten below: records of an earlier ndt_set_source_async that nobody unpacked must not be
    // unpacked over them later (and the old source is gone whether or not this call succeeds)
    h->src_aos = nullptr; h->have_source = false; h->n_src = 0;
    KeyParams kp;
    kp.inv_leaf_f32 = 1.0f / leaf; kp.inv_res_d = (double)kp.inv_leaf_f32; kp.res_d = (double)leaf; kp.key_rule = NDT_KEY_FLOOR_F32;
    double R[9], t[3];
    if (T) { for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) R[3 * r + c] = T[c * 4 + r]; t[r] = T[12 + r]; } }
    const uint32_t* d_n_out = nullptr;
    int rc = downsample_points(h->work, kp, pts, n, stride, T ? R : nullptr, T ? t : nullptr, h->sx, h->sy, h->sz, &d_n_out, h->stream, h->err);
    if (rc != NDT_OK) return rc;
    uint32_t n32 = 0;
    if (d_n_out) H_TRY(hipMemcpyAsync(&n32, d_n_out, 4, hipMemcpyDeviceToHost, h->stream));
    H_TRY(hipStreamSynchronize(h->stream));        // the only wait after the bounding box: the caller's buffer is free, n is known
    const size_t n_out = n32;
    H_TRY(h->partials.ensure((size_t)residual_grid(n_out, h->n_cu) * ACC_N * sizeof(double)));
    H_TRY(h->partials2.ensure((size_t)residual_grid(n_out, h->n_cu) * ACC_N * sizeof(double)));
    h->n_src = n_out;
    h->have_source = true;
    return NDT_OK;
}

int ndt_get_source(ndt_handle* h, float* xyz, size_t cap, size_t* n_out) {
    if (!h) return NDT_ERR_INVALID_ARG;
    if (n_out) *n_out = h->n_src;
    if (!h->have_source) { h->err = "no source"; return NDT_ERR_NO_SOURCE; }
    if (!xyz) return NDT_OK;
    if (cap < h->n_src) { h->err = "output buffer smaller than the source"; return NDT_ERR_CAPACITY; }
    if (!h->n_src) return NDT_OK;
    H_TRY(hipSetDevice(h->cfg.device));
    { int rc = ensure_source_soa(h); if (rc != NDT_OK) return rc; }
    std::vector<float> tmp(h->n_src * 3);
    H_TRY(hipMemcpyAsync(tmp.data(), h->sx.p, h->n_src * 4, hipMemcpyDeviceToHost, h->stream));
    H_TRY(hipMemcpyAsync(tmp.data() + h->n_src, h->sy.p, h->n_src * 4, hipMemcpyDeviceToHost, h->stream));
    H_TRY(hipMemcpyAsync(tmp.data() + 2 * h->n_src, h->sz.p, h->n_src * 4, hipMemcpyDeviceToHost, h->stream));
    H_TRY(hipStreamSynchronize(h->stream));
    for (size_t i = 0; i < h->n_src; ++i) { xyz[3 * i] = tmp[i]; xyz[3 * i + 1] = tmp[h->n_src + i]; xyz[3 * i + 2] = tmp[2 * h->n_src + i]; }
    return NDT_OK;
}

int ndt_add_target_from_source(ndt_handle* h, const double T[16]) {
    if (!h) return NDT_ERR_INVALID_ARG;
    if (!h->have_source) { h->err = "no source: call ndt_set_source first"; return NDT_ERR_NO_SOURCE; }
    H_TRY(hipSetDevice(h->cfg.device));
    { int rc = ensure_source_soa(h); if (rc != NDT_OK) return rc; }
    if (!h->have_target) H_TRY(start_map(h));
    double R[9], t[3];
    if (T) { for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) R[3 * r + c] = T[c * 4 + r]; t[r] = T[12 + r]; } }
    int rc = map_merge_points(h->map, h->work, h->build_cfg, h->sx.p, h->sy.as<float>(), h->sz.as<float>(), h->n_src, 0, T ? R : nullptr,
                              T ? t : nullptr, h->stream, &h->last_radix_passes, h->err);
    if (rc == NDT_OK) rc = map_lru_step(h->map, h->work, h->build_cfg, h->stream, h->err);
    if (rc == NDT_OK) h->have_target = true;
    return rc;
}

int ndt_last_radix_passes(const ndt_handle* h) { return h ? h->last_radix_passes : 0; }
int64_t ndt_source_size(const ndt_handle* h) { return h ? (int64_t)h->n_src : 0; }
int64_t ndt_num_voxels(const ndt_handle* h) { return h ? h->map.n_vox : 0; }

static void state_from_T(GnState* st, const double T[16]) {
    memset(st, 0, sizeof *st);
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) st->R[3 * r + c] = T[c * 4 + r]; st->t[r] = T[12 + r]; }
}
static void T_from_state(const GnState* st, double T[16]) {
    memset(T, 0, 16 * sizeof(double));
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T[c * 4 + r] = st->R[3 * r + c]; T[12 + r] = st->t[r]; }
    T[15] = 1.0;
}
static void unpack_acc(const double* acc, double H[36], double b[6]) {
    int t = 0;
    for (int r = 0; r < 6; ++r)
        for (int c = r; c < 6; ++c, ++t) { if (H) { H[r * 6 + c] = acc[t]; H[c * 6 + r] = acc[t]; } }
    if (b) for (int r = 0; r < 6; ++r) b[r] = acc[21 + r];
}
static int ensure_source_soa(ndt_handle* h);
static int ready(ndt_handle* h, bool keep_pending_records = false) {
    if (!h) return NDT_ERR_INVALID_ARG;
    if (!h->have_target || h->map.cap == 0) { h->err = "no target: call ndt_set_target first"; return NDT_ERR_NO_TARGET; }
    if (!h->have_source) { h->err = "no source: call ndt_set_source first"; return NDT_ERR_NO_SOURCE; }
    // (the unpack below is a launch on the handle's stream: its device must be current, whatever the caller's thread had last)
    H_TRY(hipSetDevice(h->cfg.device));
    return keep_pending_records ? NDT_OK : ensure_source_soa(h);
}
static EvalParams eval_params(const ndt_handle* h) {
    EvalParams ep;
    ep.kp = h->build_cfg.kp;            // the grid the resident map was built on
    ep.outlier_th = h->cfg.res_outlier_th;
    ep.nearby = h->cfg.nearby;
    return ep;
}
static SolveParams solve_params(const ndt_handle* h) {
    SolveParams sp;
    sp.eps = h->cfg.eps; sp.max_iterations = h->cfg.max_iterations; sp.min_effective_pts = h->cfg.min_effective_pts;
    return sp;
}

static hipEvent_t next_event(ndt_handle* h) {
    if (h->ev_used == h->ev_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        h->ev_pool.push_back(e);
    }
    return h->ev_pool[h->ev_used++];
}
static SourceView resident_source(const ndt_handle* h) {
    SourceView v{h->sx.as<float>(), h->sy.as<float>(), h->sz.as<float>(), h->n_src, AosIn{nullptr, 0, nullptr, nullptr, nullptr}};
    if (h->src_aos) v.aos = AosIn{h->src_aos, (uint32_t)h->src_aos_stride, h->sx.as<float>(), h->sy.as<float>(), h->sz.as<float>()};
    return v;
}
// everything but the chained loop's first launch reads the three arrays: unpack records that are still pending
static int ensure_source_soa(ndt_handle* h) {
    if (!h->src_aos) return NDT_OK;
    H_TRY(unpack_points_soa(h->src_aos, h->n_src, h->src_aos_stride, h->sx.as<float>(), h->sy.as<float>(), h->sz.as<float>(), h->stream));
    h->src_aos = nullptr;
    return NDT_OK;
}
static GnBuffers handle_buffers(const ndt_handle* h) { return GnBuffers{{h->partials.as<double>(), h->partials2.as<double>()}, {h->d_state, h->d_state2}}; }
// peer exchange: the mailboxes at the parity all-reduce `seq` uses
static PeerRows px_rows(const ndt_handle* h, unsigned long long seq) {
    PeerRows r{};
    const size_t par = (size_t)(seq & 1ull);
    for (int p = 0; p < h->px.n; ++p) {
        r.rows[p] = (double*)(h->px.peer[p] + par * (NDT_MAX_PEERS * ACC_N * 8));
        r.flags[p] = (unsigned long long*)(h->px.peer[p] + PX_ROWS_BYTES + par * (NDT_MAX_PEERS * 8));
    }
    return r;
}
static const double* px_local_rows(const ndt_handle* h, unsigned long long seq) {
    return (const double*)(h->px.local + (size_t)(seq & 1ull) * (NDT_MAX_PEERS * ACC_N * 8));
}
static PeerWait px_wait(const ndt_handle* h, unsigned long long seq) {
    return PeerWait{(const unsigned long long*)(h->px.local + PX_ROWS_BYTES + (size_t)(seq & 1ull) * (NDT_MAX_PEERS * 8)), h->px.n, seq};
}
// queues one residual pass over `src` at the pose in `st` (+ fixed-order reduction of the rows, + all-reduce across
// ranks, + optional GN step)
static int queue_pass(ndt_handle* h, const SourceView& src, int grid, double* rows, GnState* st, int do_solve, int force, int64_t* keys_out,
                      uint8_t* mask_out) {
    const EvalParams ep = eval_params(h);
    const SolveParams sp = solve_params(h);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->prof) { e0 = next_event(h); e1 = next_event(h); if (e0) (void)hipEventRecord(e0, h->stream); }
    launch_residual(src.x, src.y, src.z, (int)src.n, h->map.keys, h->map.slots, (uint32_t)(h->map.cap - 1), st, ep, rows, grid, force, keys_out,
                    mask_out, h->use_links, h->stream);
    if (h->prof && e1) (void)hipEventRecord(e1, h->stream);
    if (h->px.n > 0) {
        // local fixed-order sum -> this rank's row in every rank's mailbox; then the step from all ranks' rows, summed in rank
        // order (the same kernel as the single-GPU step, with the wait for the rows in front)
        const unsigned long long seq = ++h->px.seq;
        launch_reduce_rows_p2p(rows, grid, st, px_rows(h, seq), h->px.n, h->px.rank, seq, force, h->stream);
        launch_reduce_solve(px_local_rows(h, seq), h->px.n, st, sp, do_solve, force, nullptr, 0, px_wait(h, seq), h->stream);
    } else if (h->comm) {
        // local fixed-order sum into the send row, all-reduce into the receive row, then the same step on
        // every rank.  A finished state still takes part in the collective (its rows are ignored), so the
        // ranks never diverge in the number of collectives they queue.
        double* send = h->comm_rows.as<double>();
        double* recv = send + ACC_N;
        launch_reduce_rows(rows, grid, st, send, force, h->stream);
        int rc = g_rccl.all_reduce(send, recv, ACC_N, NCCL_FLOAT64, NCCL_SUM, h->comm, h->stream);
        if (rc != 0) { h->err = "ncclAllReduce failed"; return NDT_ERR_COMM; }
        launch_solve_only(st, recv, sp, do_solve, force, h->stream);
    } else {
        launch_reduce_solve(rows, grid, st, sp, do_solve, force, nullptr, 0, PeerWait{nullptr, 0, 0}, h->stream);
    }
    H_TRY(hipGetLastError());
    return NDT_OK;
}
static void collect_profile(ndt_handle* h) {
    for (size_t i = 0; i + 1 < h->ev_used; i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->ev_pool[i], h->ev_pool[i + 1]) == hipSuccess) { h->prof_us += 1e3 * ms; h->prof_launches++; }
    }
    h->ev_used = 0;
}

int ndt_eval(ndt_handle* h, const double T[16], double H[36], double b[6], double* chi2, int64_t* n_eff) {
    int rc = ready(h);
    if (rc != NDT_OK) return rc;
    if (!T) { h->err = "null pose"; return NDT_ERR_INVALID_ARG; }
    H_TRY(hipSetDevice(h->cfg.device));
    state_from_T(h->h_state, T);
    H_TRY(hipMemcpyAsync(h->d_state, h->h_state, sizeof(GnState), hipMemcpyHostToDevice, h->stream));
    if ((rc = queue_pass(h, resident_source(h), residual_grid(h->n_src, h->n_cu), h->partials.as<double>(), h->d_state, 0, 1, nullptr, nullptr)) != NDT_OK) return rc;
    H_TRY(hipMemcpyAsync(h->h_state, h->d_state, sizeof(GnState), hipMemcpyDeviceToHost, h->stream));
    H_TRY(hipStreamSynchronize(h->stream));
    if (h->prof) { collect_profile(h); h->prof_hits += h->h_state->acc[ACC_HITS]; h->prof_pts += (double)h->n_src; }
    unpack_acc(h->h_state->acc, H, b);
    if (chi2) *chi2 = h->h_state->acc[ACC_CHI2];
    if (n_eff) *n_eff = (int64_t)h->h_state->acc[ACC_NEFF];
    return NDT_OK;
}

int ndt_source_cells(ndt_handle* h, const double T[16], int64_t* keys, uint8_t* hit_mask, size_t cap) {
    int rc = ready(h);
    if (rc != NDT_OK) return rc;
    if (!T || !keys || !hit_mask) { h->err = "null argument"; return NDT_ERR_INVALID_ARG; }
    if (cap < h->n_src) { h->err = "output buffers smaller than the source"; return NDT_ERR_CAPACITY; }
    if (!h->n_src) return NDT_OK;
    H_TRY(hipSetDevice(h->cfg.device));
    H_TRY(h->cell_keys.ensure(h->n_src * 8)); H_TRY(h->cell_mask.ensure(h->n_src));
    state_from_T(h->h_state, T);
    H_TRY(hipMemcpyAsync(h->d_state, h->h_state, sizeof(GnState), hipMemcpyHostToDevice, h->stream));
    void* saved = h->comm; h->comm = nullptr;   // a local diagnostic: no collective
    const int saved_px = h->px.n; h->px.n = 0;
    rc = queue_pass(h, resident_source(h), residual_grid(h->n_src, h->n_cu), h->partials.as<double>(), h->d_state, 0, 1, h->cell_keys.as<int64_t>(),
                    h->cell_mask.as<uint8_t>());
    h->comm = saved; h->px.n = saved_px;
    if (rc != NDT_OK) return rc;
    H_TRY(hipMemcpyAsync(keys, h->cell_keys.p, h->n_src * 8, hipMemcpyDeviceToHost, h->stream));
    H_TRY(hipMemcpyAsync(hit_mask, h->cell_mask.p, h->n_src, hipMemcpyDeviceToHost, h->stream));
    H_TRY(hipStreamSynchronize(h->stream));
    if (h->prof) collect_profile(h);
    return NDT_OK;
}

// The Gauss-Newton loop of ndt_align over `src`, with its device-side working state in `B` (the handle's buffers, or the
// static scratch of ndt_align_static) and `grid` rows per launch.
static int gn_loop(ndt_handle* h, const SourceView& src, int grid, const GnBuffers& B, const double guess[16], ndt_result* out) {
    memset(out, 0, sizeof *out);
    state_from_T(h->h_state, guess);
    const int max_it = h->cfg.max_iterations;
    // With eps == 0 the loop cannot stop early: queue everything and look once.
    const int chunk = h->cfg.eps > 0.0 ? SYNC_CHUNK : max_it;
    // (profiling times the residual launches alone; every block re-reads all rows of the previous launch and repeats the
    // solve, which only pays while no CU runs more than one block per launch: measured, 230 blocks 0.134 -> 0.123 ms per
    // align, 489 blocks 0.191 -> 0.221 ms)
    const bool chain_fits = h->use_chain && !h->prof && src.n > 0 && grid <= h->n_cu;
    const bool chained = chain_fits && !h->comm && h->px.n == 0;
    const bool chained_comm = chain_fits && h->comm && h->px.n == 0;
    // (a prologue that waits occupies every CU while it waits: only when no other rank needs this GPU to produce its row)
    const bool chained_p2p = chain_fits && h->px.n > 0 && !h->px.shared_device;
    unsigned long long mirror_want = 0;
    AosIn aos = src.aos;                // records still to be unpacked: the chained loop's first launch does it by the way
    if (aos.p && !chained) { int rc0 = ensure_source_soa(h); if (rc0 != NDT_OK) return rc0; aos.p = nullptr; }
    int cur = 0;                        // B.st[cur] is the current state; the chained launches ping-pong
    int rcur = 0;                       // B.rows[rcur] is where the next launch writes its rows
    // the chained loop's first launch takes the initial pose as a kernel argument; the other forms read it from HBM
    if (!chained) H_TRY(hipMemcpyAsync(B.st[cur], h->h_state, sizeof(GnState), hipMemcpyHostToDevice, h->stream));
    GnState init_state = *h->h_state;
    bool first_launch = true;
    const EvalParams ep = eval_params(h);
    const SolveParams sp = solve_params(h);
    int rc;
    int queued = 0;
    while (queued < max_it) {
        const int m = std::min(chunk, max_it - queued);
        if (chained) {
            // m iterations = m chained launches (launch i carries the step of launch i - 1) + the last step on its own
            for (int i = 0; i < m; ++i) {
                launch_residual_chain(src.x, src.y, src.z, (int)src.n, h->map.keys, h->map.slots, (uint32_t)(h->map.cap - 1), B.st[cur], B.st[cur ^ 1],
                                      B.rows[rcur ^ 1], i ? grid : 0, sp, ep, B.rows[rcur], grid, h->use_links, first_launch ? &init_state : nullptr,
                                      PeerWait{nullptr, 0, 0}, aos, h->stream);
                if (aos.p) { aos.p = nullptr; h->src_aos = nullptr; }      // the arrays are current from here on (stream order)
                first_launch = false;
                cur ^= 1;               // the launch published the state it used in the other buffer
                rcur ^= 1;              // rows[rcur ^ 1] = the rows just written
            }
            mirror_want = ++h->mirror_seq;
            launch_reduce_solve(B.rows[rcur ^ 1], grid, B.st[cur], sp, 1, 0, h->h_state_dev, mirror_want, PeerWait{nullptr, 0, 0}, h->stream);
            H_TRY(hipGetLastError());
        } else if (chained_comm) {
            // sharded source: launch i takes the step from the all-reduced row of launch i - 1 in its prologue (one "row" to sum),
            // then residual -> local fixed-order sum -> all-reduce; three launches per iteration instead of four
            double* send = h->comm_rows.as<double>();
            double* recv = send + ACC_N;
            for (int i = 0; i < m; ++i) {
                launch_residual_chain(src.x, src.y, src.z, (int)src.n, h->map.keys, h->map.slots, (uint32_t)(h->map.cap - 1), B.st[cur], B.st[cur ^ 1], recv,
                                      i ? 1 : 0, sp, ep, B.rows[0], grid, h->use_links, nullptr, PeerWait{nullptr, 0, 0}, AosIn{nullptr, 0, nullptr, nullptr, nullptr}, h->stream);
                cur ^= 1;
                launch_reduce_rows(B.rows[0], grid, B.st[cur], send, 0, h->stream);
                if (g_rccl.all_reduce(send, recv, ACC_N, NCCL_FLOAT64, NCCL_SUM, h->comm, h->stream) != 0) { h->err = "ncclAllReduce failed"; return NDT_ERR_COMM; }
            }
            launch_solve_only(B.st[cur], recv, sp, 1, 0, h->stream);
            H_TRY(hipGetLastError());
        } else if (chained_p2p) {
            // sharded source, peer exchange: launch i waits for all ranks' rows of launch i - 1 in its prologue and takes the step from
            // them, then residual -> local fixed-order sum stored into every rank's mailbox; two launches per iteration, no library call
            unsigned long long seq = 0;
            for (int i = 0; i < m; ++i) {
                launch_residual_chain(src.x, src.y, src.z, (int)src.n, h->map.keys, h->map.slots, (uint32_t)(h->map.cap - 1), B.st[cur], B.st[cur ^ 1],
                                      i ? px_local_rows(h, seq) : nullptr, i ? h->px.n : 0, sp, ep, B.rows[0], grid, h->use_links, nullptr,
                                      i ? px_wait(h, seq) : PeerWait{nullptr, 0, 0}, AosIn{nullptr, 0, nullptr, nullptr, nullptr}, h->stream);
                cur ^= 1;
                seq = ++h->px.seq;
                launch_reduce_rows_p2p(B.rows[0], grid, B.st[cur], px_rows(h, seq), h->px.n, h->px.rank, seq, 0, h->stream);
            }
            launch_reduce_solve(px_local_rows(h, seq), h->px.n, B.st[cur], sp, 1, 0, nullptr, 0, px_wait(h, seq), h->stream);
            H_TRY(hipGetLastError());
        } else {
            for (int i = 0; i < m; ++i)
                if ((rc = queue_pass(h, src, grid, B.rows[0], B.st[cur], 1, 0, nullptr, nullptr)) != NDT_OK) return rc;
        }
        queued += m;
        // (the chained loop's last kernel has written the state into the pinned buffer itself: one command fewer to wait for)
        if (chained && h->h_state_dev) {
            // poll the word the kernel writes after the state (bounded by the stream's own completion)
            volatile unsigned long long* flag = (volatile unsigned long long*)((unsigned char*)h->h_state + sizeof(GnState));
            static const bool poll = getenv("NDT_NO_POLL") == nullptr;
            bool seen = false;
            for (unsigned k = 1; poll && !seen; ++k) {
                seen = *flag == mirror_want;
                if (!seen && (k & 255u) == 0 && hipStreamQuery(h->stream) != hipErrorNotReady) break;
                __builtin_ia32_pause();
            }
            if (!seen) H_TRY(hipStreamSynchronize(h->stream));
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
        } else {
            H_TRY(hipMemcpyAsync(h->h_state, B.st[cur], sizeof(GnState), hipMemcpyDeviceToHost, h->stream));
            H_TRY(hipStreamSynchronize(h->stream));
        }
        if (h->h_state->done) break;
    }
    const GnState loop = *h->h_state;
    if (loop.status == NDT_OK && h->cfg.final_eval) {   // H, b, chi2 at the pose that is returned
        if ((rc = queue_pass(h, src, grid, B.rows[0], B.st[cur], 0, 1, nullptr, nullptr)) != NDT_OK) return rc;
        H_TRY(hipMemcpyAsync(h->h_state, B.st[cur], sizeof(GnState), hipMemcpyDeviceToHost, h->stream));
        H_TRY(hipStreamSynchronize(h->stream));
    }
    if (h->prof) collect_profile(h);
    const GnState* st = h->h_state;
    T_from_state(st, out->T);
    unpack_acc(st->acc, out->H, out->b);
    out->chi2 = st->acc[ACC_CHI2];
    out->n_effective = (int64_t)st->acc[ACC_NEFF];
    out->last_dx_norm = sqrt(loop.dx_norm2);
    out->iterations = loop.iter;
    out->converged = loop.converged;
    out->status = loop.status;
    if (h->prof) { h->prof_hits += st->acc[ACC_HITS]; h->prof_pts += (double)src.n; }
    if (loop.status != NDT_OK) {
        h->err = loop.status == NDT_ERR_TOO_FEW_POINTS ? "fewer accepted residuals than min_effective_pts"
                 : loop.status == NDT_ERR_COMM ? "a rank's row did not arrive within the bounded wait of the peer exchange" : "Hessian not positive definite";
        return loop.status;
    }
    return NDT_OK;
}

int ndt_align(ndt_handle* h, const double guess[16], ndt_result* out) {
    int rc = ready(h, true);
    if (rc != NDT_OK) return rc;
    if (!guess || !out) { h->err = "null argument"; return NDT_ERR_INVALID_ARG; }
    H_TRY(hipSetDevice(h->cfg.device));
    return gn_loop(h, resident_source(h), residual_grid(h->n_src, h->n_cu), handle_buffers(h), guess, out);
}

// The entry point that allocates nothing (declared in include/ndt_mi355.h): ndt_align on a source the caller holds in HBM
// as three float arrays, with the loop's partial rows and states in the static device array g_static_scratch (through the
// per-device pointer resolved at ndt_create).  No hipMalloc / hipFree / allocating routine on this path.
int ndt_align_static(ndt_handle* h, const float* d_x, const float* d_y, const float* d_z, size_t n, const double guess[16], ndt_result* out) {
    if (!h) return NDT_ERR_INVALID_ARG;
    if (!h->have_target || h->map.cap == 0) { h->err = "no target: call ndt_set_target first"; return NDT_ERR_NO_TARGET; }
    if (!guess || !out || (n && (!d_x || !d_y || !d_z))) { h->err = "null argument"; return NDT_ERR_INVALID_ARG; }
    if (n > (size_t)NDT_STATIC_MAX_POINTS) {
        h->err = "ndt_align_static: " + std::to_string(n) + " source points exceed NDT_STATIC_MAX_POINTS (" + std::to_string(NDT_STATIC_MAX_POINTS) +
                 "); use ndt_set_source + ndt_align";
        return NDT_ERR_CAPACITY;
    }
    if (h->comm || h->px.n > 0) { h->err = "ndt_align_static does not run with a communicator attached (its all-reduce rows are allocated buffers)"; return NDT_ERR_INVALID_ARG; }
    if (h->prof) { h->err = "ndt_align_static does not run with profiling enabled (the event pool allocates)"; return NDT_ERR_INVALID_ARG; }
    if (!h->static_scratch) { h->err = "static scratch not resolved on this device"; return NDT_ERR_HIP; }
    H_TRY(hipSetDevice(h->cfg.device));
    // the array is shared by every handle on this GPU: calls from several host threads take turns (the call ends
    // synchronised, so holding the lock for its duration costs nothing); other processes on the GPU have their own copy
    std::lock_guard<std::mutex> turn(g_static_call_mu[h->cfg.device]);
    // rows of a launch: as ndt_align sizes them, within the scratch's 256 rows (on MI355X the two are the same number,
    // so both entry points sum in the same order and return the same bits)
    const int grid = std::min(residual_grid(n, h->n_cu), STATIC_MAX_ROWS);
    unsigned char* base = h->static_scratch;               // the device address, never the array's name
    GnBuffers B;
    B.rows[0] = (double*)base;
    B.rows[1] = (double*)(base + STATIC_ROWS_BYTES);
    B.st[0] = (GnState*)(base + 2 * STATIC_ROWS_BYTES);
    B.st[1] = (GnState*)(base + 2 * STATIC_ROWS_BYTES + STATIC_STATE_BYTES);
    // every call on this GPU shares the array and finds the last call's contents in it: the two states start at zero
    // (the rows need no clearing: a launch writes each of its rows in full before any kernel reads them)
    H_TRY(hipMemsetAsync(base + 2 * STATIC_ROWS_BYTES, 0, 2 * STATIC_STATE_BYTES, h->stream));
    return gn_loop(h, SourceView{d_x, d_y, d_z, n, AosIn{nullptr, 0, nullptr, nullptr, nullptr}}, grid, B, guess, out);
}

int ndt_export_voxels(ndt_handle* h, int64_t* keys, int32_t* counts, double* mean, double* cov, double* info, uint8_t* valid,
                      size_t cap, size_t* n_out) {
    if (!h) return NDT_ERR_INVALID_ARG;
    if (n_out) *n_out = (size_t)h->map.n_vox;
    if (!h->have_target) { h->err = "no target"; return NDT_ERR_NO_TARGET; }
    if (cap < (size_t)h->map.n_vox) { h->err = "output buffers smaller than the map"; return NDT_ERR_CAPACITY; }
    H_TRY(hipSetDevice(h->cfg.device));
    std::vector<int64_t> k; std::vector<int32_t> c; std::vector<double> m, cv, inf; std::vector<uint8_t> v;
    int rc = map_export(h->map, h->build_cfg.kp, k, c, m, cv, inf, v, h->stream, h->err);
    if (rc != NDT_OK) return rc;
    const size_t nv = k.size();
    if (keys) memcpy(keys, k.data(), nv * 8);
    if (counts) memcpy(counts, c.data(), nv * 4);
    if (mean) memcpy(mean, m.data(), nv * 24);
    if (cov) memcpy(cov, cv.data(), nv * 72);
    if (info) memcpy(info, inf.data(), nv * 72);
    if (valid) memcpy(valid, v.data(), nv);
    return NDT_OK;
}

// ---- map serialisation (layout in ndt_mi355.h)

namespace {
struct MapBlobHeader {
    char magic[8];
    uint32_t version;
    float resolution;
    int32_t key_rule;
    int32_t reserved0;
    uint64_t n_voxels;
    unsigned char reserved[32];
};
static_assert(sizeof(MapBlobHeader) == 64, "map blob header is 64 bytes");
const char MAP_MAGIC[8] = {'N', 'D', 'T', 'M', 'I', '3', '5', '5'};
inline size_t map_blob_size(size_t n) { return sizeof(MapBlobHeader) + n * (8 + 24 + 48 + 4); }
}  // namespace

int ndt_map_save(ndt_handle* h, void* buf, size_t cap_bytes, size_t* n_bytes) {
    if (!h) return NDT_ERR_INVALID_ARG;
    if (!h->have_target) { h->err = "no target"; return NDT_ERR_NO_TARGET; }
    const size_t nv = (size_t)h->map.n_vox, need = map_blob_size(nv);
    if (n_bytes) *n_bytes = need;
    if (!buf) return NDT_OK;
    if (cap_bytes < need) { h->err = "buffer smaller than the serialised map"; return NDT_ERR_CAPACITY; }
    H_TRY(hipSetDevice(h->cfg.device));
    std::vector<int64_t> k; std::vector<int32_t> c; std::vector<double> m, cv;
    int rc = map_dump(h->map, k, c, m, cv, h->stream, h->err);
    if (rc != NDT_OK) return rc;
    if (k.size() != nv) { h->err = "voxel count changed during the dump"; return NDT_ERR_HIP; }
    MapBlobHeader hd;
    memset(&hd, 0, sizeof hd);
    memcpy(hd.magic, MAP_MAGIC, 8);
    hd.version = 1;
    hd.resolution = (float)h->build_cfg.kp.res_d;
    hd.key_rule = h->build_cfg.kp.key_rule;
    hd.n_voxels = nv;
    unsigned char* o = (unsigned char*)buf;
    memcpy(o, &hd, sizeof hd); o += sizeof hd;
    memcpy(o, k.data(), nv * 8); o += nv * 8;
    memcpy(o, m.data(), nv * 24); o += nv * 24;
    memcpy(o, cv.data(), nv * 48); o += nv * 48;
    memcpy(o, c.data(), nv * 4);
    return NDT_OK;
}

int ndt_map_load(ndt_handle* h, const void* buf, size_t n_bytes) {
    if (!h) return NDT_ERR_INVALID_ARG;
    if (!buf || n_bytes < sizeof(MapBlobHeader)) { h->err = "map blob too short"; return NDT_ERR_INVALID_ARG; }
    MapBlobHeader hd;
    memcpy(&hd, buf, sizeof hd);
    if (memcmp(hd.magic, MAP_MAGIC, 8) != 0 || hd.version != 1) { h->err = "not a version-1 NDTMI355 map blob"; return NDT_ERR_INVALID_ARG; }
    if (hd.resolution != h->cfg.resolution || hd.key_rule != h->cfg.key_rule) {
        h->err = "map blob was saved with another resolution or key_rule";
        return NDT_ERR_INVALID_ARG;
    }
    if (hd.n_voxels >= (1ull << 31) || n_bytes != map_blob_size((size_t)hd.n_voxels)) { h->err = "map blob size does not match its header"; return NDT_ERR_INVALID_ARG; }
    const size_t nv = (size_t)hd.n_voxels;
    const unsigned char* in = (const unsigned char*)buf + sizeof hd;
    std::vector<int64_t> k(nv); std::vector<double> m(nv * 3), cv(nv * 6); std::vector<int32_t> c(nv);
    memcpy(k.data(), in, nv * 8); in += nv * 8;
    memcpy(m.data(), in, nv * 24); in += nv * 24;
    memcpy(cv.data(), in, nv * 48); in += nv * 48;
    memcpy(c.data(), in, nv * 4);
    for (size_t i = 0; i < nv; ++i) {      // a blob is untrusted input: the insert kernel relies on distinct in-range keys
        int32_t x, y, z;
        ndt_unpack(k[i], x, y, z);
        const bool in_range = k[i] >= 0 && ndt_pack(x, y, z) == k[i];
        if (!in_range || (i && k[i] <= k[i - 1]) || c[i] < 1) { h->err = "map blob holds an invalid, unsorted or duplicate key, or an empty voxel"; return NDT_ERR_INVALID_ARG; }
    }
    for (size_t i = 0; i < nv * 3; ++i) if (!std::isfinite(m[i])) { h->err = "map blob holds a non-finite mean"; return NDT_ERR_INVALID_ARG; }
    for (size_t i = 0; i < nv * 6; ++i) if (!std::isfinite(cv[i])) { h->err = "map blob holds a non-finite covariance"; return NDT_ERR_INVALID_ARG; }
    H_TRY(hipSetDevice(h->cfg.device));
    H_TRY(start_map(h));
    h->have_target = false;
    int rc = map_restore(h->map, h->work, h->build_cfg, k.data(), c.data(), m.data(), cv.data(), nv, h->stream, h->err);
    if (rc == NDT_OK) h->have_target = true;
    return rc;
}

// ---- PCL compatibility mode (kernels and driver in ndt_pcl.hip)

void ndt_pcl_default_params(ndt_pcl_params* p) {
    if (!p) return;
    p->outlier_ratio = 0.55; p->step_size = 0.1; p->rotation_eps = 0.0; p->hessian_d1_as_pcl = 1; p->reserved = 0;
}
void ndt_pcl_vector_from_matrix(const double T[16], double p[6]) { if (T && p) pcl_vector_from_matrix(T, p); }
void ndt_pcl_matrix_from_vector(const double p[6], double T[16]) { if (T && p) pcl_matrix_from_vector(p, T); }

static int pcl_engine(ndt_handle* h, const ndt_pcl_params* prm, PclEngine* e) {
    int rc = ready(h);
    if (rc != NDT_OK) return rc;
    if (!prm || !(prm->outlier_ratio > 0.0) || !(prm->outlier_ratio < 1.0) || !(prm->step_size > 0.0)) {
        h->err = "PCL-mode parameters: outlier_ratio must be in (0,1) and step_size positive";
        return NDT_ERR_INVALID_ARG;
    }
    H_TRY(hipSetDevice(h->cfg.device));
    H_TRY(h->pcl_partials.ensure(((size_t)pcl_grid(h->n_src, h->n_cu) + 1) * ACC_N * sizeof(double)));
    e->stream = h->stream;
    e->keys = h->map.keys; e->slots = h->map.slots; e->mask = (uint32_t)(h->map.cap - 1);
    e->sx = h->sx.as<float>(); e->sy = h->sy.as<float>(); e->sz = h->sz.as<float>(); e->n = h->n_src;
    e->kp = h->build_cfg.kp;
    e->partials = h->pcl_partials.as<double>() + ACC_N;
    e->d_row = h->pcl_partials.as<double>();
    e->use_links = h->use_links;
    e->n_cu = h->n_cu;
    e->h_row = h->h_state->acc;          // pinned; the Gauss-Newton loop is not running while this mode is
    return NDT_OK;
}

int ndt_pcl_eval(ndt_handle* h, const ndt_pcl_params* prm, const double p[6], double* score, double g[6], double H[36], int64_t* n_neighbours) {
    if (!h || !p) return NDT_ERR_INVALID_ARG;
    PclEngine e;
    int rc = pcl_engine(h, prm, &e);
    if (rc != NDT_OK) return rc;
    for (int k = 0; k < 6; ++k) if (!std::isfinite(p[k])) { h->err = "non-finite transform vector"; return NDT_ERR_INVALID_ARG; }
    return pcl_eval(e, *prm, p, score, g, H, n_neighbours, h->err);
}

int ndt_pcl_time_pass(ndt_handle* h, const ndt_pcl_params* prm, const double p[6], int with_hessian, int reps, double* us_per_launch,
                      double* voxels_per_point, double* neighbours_per_point) {
    if (!h || !p || reps < 1) return NDT_ERR_INVALID_ARG;
    PclEngine e;
    int rc = pcl_engine(h, prm, &e);
    if (rc != NDT_OK) return rc;
    return pcl_time_pass(e, *prm, p, with_hessian != 0, reps, us_per_launch, voxels_per_point, neighbours_per_point, h->err);
}

int ndt_pcl_align(ndt_handle* h, const ndt_pcl_params* prm, const double guess[16], ndt_pcl_result* out) {
    if (!h || !guess || !out) return NDT_ERR_INVALID_ARG;
    PclEngine e;
    int rc = pcl_engine(h, prm, &e);
    if (rc != NDT_OK) { memset(out, 0, sizeof *out); out->status = rc; return rc; }
    for (int k = 0; k < 16; ++k) if (!std::isfinite(guess[k])) { h->err = "non-finite guess"; return NDT_ERR_INVALID_ARG; }
    return pcl_align(e, *prm, h->cfg.max_iterations, h->cfg.eps, guess, out, h->err);
}

// ---- the NDT Hessian as PointCloudFactor's information (host only, no GPU work)

int ndt_information_from_hessian(const double H[36], double scale, double rel_floor, double info[36]) {
    if (!H || !info || !(scale > 0.0) || !(rel_floor > 0.0) || rel_floor > 1.0) return NDT_ERR_INVALID_ARG;
    double A[6][6], V[6][6];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            const double v = 0.5 * (H[i * 6 + j] + H[j * 6 + i]);     // symmetrise
            if (!isfinite(v)) return NDT_ERR_INVALID_ARG;
            A[i][j] = v; V[i][j] = i == j ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 60; ++sweep) {                          // cyclic Jacobi
        double off = 0.0, dia = 0.0;
        for (int i = 0; i < 6; ++i) { dia += A[i][i] * A[i][i]; for (int j = i + 1; j < 6; ++j) off += A[i][j] * A[i][j]; }
        if (off <= 1e-30 * dia || off == 0.0) break;
        for (int p = 0; p < 5; ++p)
            for (int q = p + 1; q < 6; ++q) {
                if (A[p][q] == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < 6; ++k) { const double a = A[k][p], b = A[k][q]; A[k][p] = c * a - sn * b; A[k][q] = sn * a + c * b; }
                for (int k = 0; k < 6; ++k) { const double a = A[p][k], b = A[q][k]; A[p][k] = c * a - sn * b; A[q][k] = sn * a + c * b; }
                for (int k = 0; k < 6; ++k) { const double a = V[k][p], b = V[k][q]; V[k][p] = c * a - sn * b; V[k][q] = sn * a + c * b; }
            }
    }
    double lmax = 0.0;
    for (int i = 0; i < 6; ++i) lmax = fmax(lmax, A[i][i]);
    if (!(lmax > 0.0)) return NDT_ERR_SINGULAR;
    const double fl = rel_floor * lmax;
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double v = 0.0;
            for (int k = 0; k < 6; ++k) v += V[i][k] * fmax(A[k][k], fl) * V[j][k];
            info[i * 6 + j] = scale * v;
        }
    for (int i = 0; i < 6; ++i)
        for (int j = i + 1; j < 6; ++j) info[j * 6 + i] = info[i * 6 + j] = 0.5 * (info[i * 6 + j] + info[j * 6 + i]);
    return NDT_OK;
}

int64_t ndt_pack_key(int32_t ix, int32_t iy, int32_t iz) { return ndt_pack(ix, iy, iz); }
void ndt_unpack_key(int64_t key, int32_t* ix, int32_t* iy, int32_t* iz) {
    int32_t a, b, c;
    ndt_unpack(key, a, b, c);
    if (ix) *ix = a;
    if (iy) *iy = b;
    if (iz) *iz = c;
}

int ndt_set_batch_sources(ndt_handle* h, const void* pts, size_t stride, const int64_t* scan_offsets, size_t n_scans) {
    if (!h) return NDT_ERR_INVALID_ARG;
    if (!scan_offsets || !n_scans || stride < 12 || (stride & 3)) { h->err = "bad argument"; return NDT_ERR_INVALID_ARG; }
    int64_t max_pts = 0;
    for (size_t i = 0; i < n_scans; ++i) {
        const int64_t m = scan_offsets[i + 1] - scan_offsets[i];
        if (m < 0 || scan_offsets[0] != 0) { h->err = "scan_offsets must start at 0 and not decrease"; return NDT_ERR_INVALID_ARG; }
        max_pts = std::max(max_pts, m);
    }
    const size_t total = (size_t)scan_offsets[n_scans];
    if ((total && !pts) || max_pts >= (1ll << 31)) { h->err = "bad point buffer"; return NDT_ERR_INVALID_ARG; }
    H_TRY(hipSetDevice(h->cfg.device));
    hipStream_t s = h->stream;
    H_TRY(h->braw.ensure(total * stride + 16));
    H_TRY(h->bx.ensure(total * 4 + 16)); H_TRY(h->by.ensure(total * 4 + 16)); H_TRY(h->bz.ensure(total * 4 + 16));
    if (total) {
        H_TRY(hipMemcpyAsync(h->braw.p, pts, total * stride, hipMemcpyDefault, s));
        H_TRY(unpack_points_soa(h->braw.p, total, stride, h->bx.as<float>(), h->by.as<float>(), h->bz.as<float>(), s));
    }
    H_TRY(h->boff.ensure((n_scans + 1) * 8));
    H_TRY(hipMemcpyAsync(h->boff.p, scan_offsets, (n_scans + 1) * 8, hipMemcpyHostToDevice, s));
    H_TRY(hipStreamSynchronize(s));           // the caller's buffers are free again
    h->b_offsets.assign(scan_offsets, scan_offsets + n_scans + 1);
    h->b_max_pts = max_pts;
    h->b_pairs = 0;
    return NDT_OK;
}

int ndt_align_batch_resident(ndt_handle* h, const int32_t* pair_scan, size_t n_pairs, const double* guesses, ndt_result* results) {
    if (!h) return NDT_ERR_INVALID_ARG;
    if (!h->have_target || h->map.cap == 0) { h->err = "no target: call ndt_set_target first"; return NDT_ERR_NO_TARGET; }
    if (h->b_offsets.size() < 2) { h->err = "no batch sources: call ndt_set_batch_sources first"; return NDT_ERR_NO_SOURCE; }
    const size_t n_scans = h->b_offsets.size() - 1;
    const int64_t* scan_offsets = h->b_offsets.data();
    const int64_t max_pts = h->b_max_pts;
    if (!n_pairs || !guesses || !results) { h->err = "bad argument"; return NDT_ERR_INVALID_ARG; }
    if (!pair_scan && n_pairs != n_scans) { h->err = "without pair_scan there must be one scan per pair"; return NDT_ERR_INVALID_ARG; }
    if (h->comm || h->px.n > 0) { h->err = "batched replay has no collective: destroy the communicator first"; return NDT_ERR_INVALID_ARG; }
    if (pair_scan)
        for (size_t p = 0; p < n_pairs; ++p)
            if (pair_scan[p] < 0 || (size_t)pair_scan[p] >= n_scans) { h->err = "pair_scan out of range"; return NDT_ERR_INVALID_ARG; }
    H_TRY(hipSetDevice(h->cfg.device));
    hipStream_t s = h->stream;
    if (pair_scan) {
        H_TRY(h->bpair.ensure(n_pairs * 4));
        H_TRY(hipMemcpyAsync(h->bpair.p, pair_scan, n_pairs * 4, hipMemcpyHostToDevice, s));
    }
    std::vector<GnState> st(n_pairs);
    for (size_t p = 0; p < n_pairs; ++p) state_from_T(&st[p], guesses + 16 * p);
    H_TRY(h->bstates.ensure(n_pairs * sizeof(GnState)));
    H_TRY(hipMemcpyAsync(h->bstates.p, st.data(), n_pairs * sizeof(GnState), hipMemcpyHostToDevice, s));
    // Launch lanes.  The pairs are independent problems, so the batch is cut into (up to) two LANES, each on a HIP stream
    // of its own with no ordering between them: one lane's blocks fill the tail of the other's residual launch, and its
    // latency-bound reduce + Gauss-Newton kernel runs underneath the other's residual kernel instead of with the GPU to
    // itself.  (More lanes bring nothing; see DESIGN.md §5 for what this replaced.)
    const size_t CH = 32768;                  // pairs per launch: gridDim.y limit is 65535
    struct Lane { hipStream_t s; size_t p0, np; };
    const int n_lanes = (int)std::max<size_t>(1, std::min<size_t>((size_t)h->batch_lanes, n_pairs / 2));
    Lane lanes[4];
    for (int l = 0; l < n_lanes; ++l) {
        const size_t per = (n_pairs + n_lanes - 1) / n_lanes;
        lanes[l].s = l ? h->lane_stream[l - 1] : s;
        lanes[l].p0 = l * per; lanes[l].np = std::min(per, n_pairs - l * per);
    }
    const size_t group = std::min(CH, lanes[0].np);      // the largest launch: sizes the grid rows for all of them
    const int rows = residual_grid((size_t)max_pts, h->n_cu, group, h->cfg.nearby);
    H_TRY(h->bpartials.ensure(n_pairs * (size_t)rows * ACC_N * sizeof(double)));
    H_TRY(h->bdone.ensure(256));
    H_TRY(hipMemsetAsync(h->bdone.p, 0, 256, s));
    // no wait here: pair_scan (the caller's) and st stay alive until the loop's own synchronisation below
    if (n_lanes > 1) H_TRY(hipEventRecord(h->ev_fork, s));
    for (int l = 1; l < n_lanes; ++l) H_TRY(hipStreamWaitEvent(lanes[l].s, h->ev_fork, 0));

    h->b_pairs = n_pairs; h->b_rows = rows; h->b_has_pair_scan = pair_scan != nullptr; h->b_group = group;
    h->b_points = 0;
    for (size_t p = 0; p < n_pairs; ++p) { const size_t sc = pair_scan ? (size_t)pair_scan[p] : p; h->b_points += (double)(scan_offsets[sc + 1] - scan_offsets[sc]); }
    const EvalParams ep = eval_params(h);
    const SolveParams sp = solve_params(h);
    GnState* dst = h->bstates.as<GnState>();
    const int32_t* dps = pair_scan ? h->bpair.as<int32_t>() : nullptr;
    auto pass = [&](const Lane& L, int do_solve, int force) {
        for (size_t p0 = L.p0; p0 < L.p0 + L.np; p0 += CH) {
            const int np = (int)std::min(CH, L.p0 + L.np - p0);
            launch_residual_batch(h->bx.as<float>(), h->by.as<float>(), h->bz.as<float>(), h->boff.as<int64_t>(), dps, h->map.keys,
                                  h->map.slots, (uint32_t)(h->map.cap - 1), dst, ep, h->bpartials.as<double>(), rows, (int)p0, np, force, h->use_links, L.s);
            launch_reduce_solve_batch(h->bpartials.as<double>() + p0 * (size_t)rows * ACC_N, rows, dst + p0, np, sp, do_solve, force,
                                      h->bdone.as<int32_t>(), L.s);
        }
    };
    // m passes on every lane (launches interleaved so that both streams fill early), then the lanes are joined on s
    auto run = [&](int m, int do_solve, int force) -> hipError_t {
        for (int i = 0; i < m; ++i)
            for (int l = 0; l < n_lanes; ++l) pass(lanes[l], do_solve, force);
        hipError_t e = hipGetLastError();
        for (int l = 1; l < n_lanes && e == hipSuccess; ++l) {
            e = hipEventRecord(h->ev_join[l - 1], lanes[l].s);
            if (e == hipSuccess) e = hipStreamWaitEvent(s, h->ev_join[l - 1], 0);
        }
        return e;
    };
    const int max_it = h->cfg.max_iterations;
    const int chunk = h->cfg.eps > 0.0 ? SYNC_CHUNK : max_it;
    for (int queued = 0; queued < max_it;) {
        const int m = std::min(chunk, max_it - queued);
        H_TRY(run(m, 1, 0));
        queued += m;
        if (queued >= max_it) break;                  // nothing left to decide: the state read-back below is the synchronisation
        int32_t n_done = 0;
        H_TRY(hipMemcpyAsync(&n_done, h->bdone.p, 4, hipMemcpyDeviceToHost, s));
        H_TRY(hipStreamSynchronize(s));
        if ((size_t)n_done >= n_pairs) break;
    }
    std::vector<GnState> loop(n_pairs);
    H_TRY(hipMemcpyAsync(loop.data(), dst, n_pairs * sizeof(GnState), hipMemcpyDeviceToHost, s));
    H_TRY(hipStreamSynchronize(s));
    if (h->cfg.final_eval) {
        H_TRY(run(1, 0, 1));
        H_TRY(hipMemcpyAsync(st.data(), dst, n_pairs * sizeof(GnState), hipMemcpyDeviceToHost, s));
        H_TRY(hipStreamSynchronize(s));
    } else st = loop;
    for (size_t p = 0; p < n_pairs; ++p) {
        ndt_result* out = results + p;
        memset(out, 0, sizeof *out);
        const GnState* fin = (loop[p].status == NDT_OK) ? &st[p] : &loop[p];
        T_from_state(&loop[p], out->T);
        unpack_acc(fin->acc, out->H, out->b);
        out->chi2 = fin->acc[ACC_CHI2];
        out->n_effective = (int64_t)fin->acc[ACC_NEFF];
        out->last_dx_norm = sqrt(loop[p].dx_norm2);
        out->iterations = loop[p].iter;
        out->converged = loop[p].converged;
        out->status = loop[p].status;
    }
    return NDT_OK;
}

int ndt_align_batch(ndt_handle* h, const void* pts, size_t stride, const int64_t* scan_offsets, size_t n_scans,
                    const int32_t* pair_scan, size_t n_pairs, const double* guesses, ndt_result* results) {
    int rc = ndt_set_batch_sources(h, pts, stride, scan_offsets, n_scans);
    if (rc != NDT_OK) return rc;
    return ndt_align_batch_resident(h, pair_scan, n_pairs, guesses, results);
}

int ndt_comm_unique_id(char id_out[128]) {
    if (!id_out || !rccl_load()) return NDT_ERR_COMM;
    return g_rccl.get_id(id_out) == 0 ? NDT_OK : NDT_ERR_COMM;
}
int ndt_comm_init(ndt_handle* h, int rank, int n_ranks, const char id[128]) {
    if (!h || !id || n_ranks < 1 || rank < 0 || rank >= n_ranks) return NDT_ERR_INVALID_ARG;
    if (!rccl_load()) { h->err = "librccl.so not found"; return NDT_ERR_COMM; }
    H_TRY(hipSetDevice(h->cfg.device));
    if (h->comm) { g_rccl.destroy(h->comm); h->comm = nullptr; }
    if (h->px.local) px_release(h);
    ndt_nccl_id uid;
    memcpy(uid.internal, id, 128);
    if (g_rccl.init_rank(&h->comm, n_ranks, uid, rank) != 0) { h->comm = nullptr; h->err = "ncclCommInitRank failed"; return NDT_ERR_COMM; }
    h->n_ranks = n_ranks;
    H_TRY(h->comm_rows.ensure(2 * ACC_N * sizeof(double)));
    H_TRY(hipMemsetAsync(h->comm_rows.p, 0, 2 * ACC_N * sizeof(double), h->stream));
    return NDT_OK;
}
// ---- peer exchange: the same all-reduce without a library.  Every rank exports the IPC handle of its mailbox, the caller
// distributes the handles (as it distributes RCCL's unique id), every rank opens the others' mailboxes.
static void px_release(ndt_handle* h) {
    for (int p = 0; p < NDT_MAX_PEERS; ++p) {
        if (h->px.opened[p] && h->px.peer[p]) (void)hipIpcCloseMemHandle(h->px.peer[p]);
        h->px.peer[p] = nullptr; h->px.opened[p] = false;
    }
    if (h->px.local) (void)hipFree(h->px.local);
    h->px.local = nullptr; h->px.n = 0; h->px.rank = 0; h->px.seq = 0;
}
int ndt_comm_p2p_export(ndt_handle* h, unsigned char handle_out[NDT_P2P_HANDLE_BYTES]) {
    if (!h || !handle_out) return NDT_ERR_INVALID_ARG;
    static_assert(sizeof(hipIpcMemHandle_t) == 64 && NDT_P2P_HANDLE_BYTES == 128, "handle = 64-byte HIP IPC handle + the device's PCI bus id");
    H_TRY(hipSetDevice(h->cfg.device));
    (void)hipStreamSynchronize(h->stream);
    px_release(h);
    // fine-grained device memory: peers write it while this device's kernels poll it.  No fallback to plain (coarse-grained)
    // hipMalloc: stores of a peer into coarse-grained memory need not become visible to a kernel that is already polling,
    // and the failure would only show as a bounded-wait NDT_ERR_COMM on every align
    if (hipExtMallocWithFlags((void**)&h->px.local, PX_BYTES, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        h->px.local = nullptr;
        h->err = "ndt_comm_p2p_export: fine-grained device memory (hipExtMallocWithFlags, hipDeviceMallocFinegrained) is not available on this "
                 "device; the peer exchange needs it — use ndt_comm_init (RCCL) instead";
        return NDT_ERR_COMM;
    }
    H_TRY(hipMemset(h->px.local, 0, PX_BYTES));
    hipIpcMemHandle_t ih;
    H_TRY(hipIpcGetMemHandle(&ih, h->px.local));
    memset(handle_out, 0, NDT_P2P_HANDLE_BYTES);
    memcpy(handle_out, &ih, 64);
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, h->cfg.device) != hipSuccess) { (void)hipGetLastError(); bus[0] = 0; }
    memcpy(handle_out + 64, bus, 63);          // which GPU the mailbox lives on (ranks that share one are told apart below)
    return NDT_OK;
}
int ndt_comm_p2p_init(ndt_handle* h, int rank, int n_ranks, const unsigned char* handles) {
    if (!h || !handles || n_ranks < 1 || n_ranks > NDT_MAX_PEERS || rank < 0 || rank >= n_ranks) return NDT_ERR_INVALID_ARG;
    if (!h->px.local) { h->err = "ndt_comm_p2p_init: call ndt_comm_p2p_export first"; return NDT_ERR_COMM; }
    if (h->comm) { h->err = "an RCCL communicator is attached: destroy it first"; return NDT_ERR_INVALID_ARG; }
    H_TRY(hipSetDevice(h->cfg.device));
    h->px.shared_device = false;
    const char* my_bus = (const char*)(handles + (size_t)rank * NDT_P2P_HANDLE_BYTES + 64);
    for (int p = 0; p < n_ranks; ++p) {
        if (p == rank) { h->px.peer[p] = h->px.local; continue; }
        const char* bus = (const char*)(handles + (size_t)p * NDT_P2P_HANDLE_BYTES + 64);
        // (NDT_P2P_CHAIN=1: tests force the prologue wait with two ranks on one GPU, on grids that leave the other rank room)
        if (my_bus[0] && strncmp(my_bus, bus, 63) == 0 && !getenv("NDT_P2P_CHAIN")) h->px.shared_device = true;
        hipIpcMemHandle_t ih;
        memcpy(&ih, handles + (size_t)p * NDT_P2P_HANDLE_BYTES, 64);
        void* ptr = nullptr;
        if (hipIpcOpenMemHandle(&ptr, ih, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            (void)hipGetLastError();
            h->err = "hipIpcOpenMemHandle failed for rank " + std::to_string(p);
            for (int q = 0; q < p; ++q) if (h->px.opened[q]) { (void)hipIpcCloseMemHandle(h->px.peer[q]); h->px.peer[q] = nullptr; h->px.opened[q] = false; }
            return NDT_ERR_COMM;
        }
        h->px.peer[p] = (unsigned char*)ptr; h->px.opened[p] = true;
    }
    h->px.n = n_ranks; h->px.rank = rank; h->px.seq = 0;
    h->n_ranks = n_ranks;
    return NDT_OK;
}
int ndt_comm_size(const ndt_handle* h) {      // ranks of the attached communicator, as RCCL counts them (1 without one)
    if (h && h->px.n > 0) return h->px.n;
    if (!h || !h->comm) return 1;
    int n = 0;
    if (!g_rccl.count || g_rccl.count(h->comm, &n) != 0) return h->n_ranks;
    return n;
}
int ndt_comm_destroy(ndt_handle* h) {
    if (!h) return NDT_ERR_INVALID_ARG;
    (void)hipSetDevice(h->cfg.device);
    if (h->comm) { (void)hipStreamSynchronize(h->stream); g_rccl.destroy(h->comm); h->comm = nullptr; h->n_ranks = 1; }
    if (h->px.local) { (void)hipStreamSynchronize(h->stream); px_release(h); h->n_ranks = 1; }
    return NDT_OK;
}

int ndt_profile(ndt_handle* h, int enable) {
    if (!h) return NDT_ERR_INVALID_ARG;
    h->prof = enable != 0;
    h->prof_us = 0; h->prof_launches = 0; h->prof_hits = 0; h->prof_pts = 0; h->ev_used = 0;
    return NDT_OK;
}
int ndt_profile_read(ndt_handle* h, double* us, int64_t* launches, double* hits_per_point) {
    if (!h) return NDT_ERR_INVALID_ARG;
    if (us) *us = h->prof_us;
    if (launches) *launches = h->prof_launches;
    if (hits_per_point) *hits_per_point = h->prof_pts > 0 ? h->prof_hits / h->prof_pts : 0.0;
    return NDT_OK;
}

int ndt_time_residual(ndt_handle* h, const double T[16], int reps, double* us_per_launch) {
    int rc = ready(h);
    if (rc != NDT_OK) return rc;
    if (!T || reps < 1 || !us_per_launch) { h->err = "bad argument"; return NDT_ERR_INVALID_ARG; }
    H_TRY(hipSetDevice(h->cfg.device));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    H_TRY(hipEventCreate(&e0)); H_TRY(hipEventCreate(&e1));
    state_from_T(h->h_state, T);
    H_TRY(hipMemcpyAsync(h->d_state, h->h_state, sizeof(GnState), hipMemcpyHostToDevice, h->stream));
    const int grid = residual_grid(h->n_src, h->n_cu);
    const EvalParams ep = eval_params(h);
    auto pass = [&]() {
        launch_residual(h->sx.as<float>(), h->sy.as<float>(), h->sz.as<float>(), (int)h->n_src, h->map.keys, h->map.slots,
                        (uint32_t)(h->map.cap - 1), h->d_state, ep, h->partials.as<double>(), grid, 1, nullptr, nullptr, h->use_links, h->stream);
    };
    for (int i = 0; i < 3; ++i) pass();   // warm
    H_TRY(hipEventRecord(e0, h->stream));
    for (int i = 0; i < reps; ++i) pass();
    H_TRY(hipEventRecord(e1, h->stream));
    H_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    H_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *us_per_launch = 1e3 * (double)ms / reps;
    return NDT_OK;
}

// the chained launch of the single-scan loop (previous step in the prologue + residual pass), back to back: launch i
// reduces launch i - 1's rows, takes the step and evaluates at the new pose, exactly as inside ndt_align, with the
// iteration cap and eps lifted so that the state never finishes
int ndt_time_chain(ndt_handle* h, const double T[16], int reps, double* us_per_launch) {
    int rc = ready(h);
    if (rc != NDT_OK) return rc;
    if (!T || reps < 1 || !us_per_launch || !h->n_src) { h->err = "bad argument"; return NDT_ERR_INVALID_ARG; }
    H_TRY(hipSetDevice(h->cfg.device));
    const int grid = residual_grid(h->n_src, h->n_cu);
    if (grid > h->n_cu) { h->err = "the chained launch is not used for a source of this size"; return NDT_ERR_INVALID_ARG; }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    H_TRY(hipEventCreate(&e0)); H_TRY(hipEventCreate(&e1));
    GnState init;
    state_from_T(&init, T);
    const EvalParams ep = eval_params(h);
    SolveParams sp = solve_params(h);
    sp.eps = 0.0; sp.max_iterations = 0x7fffffff;
    const GnBuffers B = handle_buffers(h);
    int cur = 0, rcur = 0;
    bool first = true;
    auto pass = [&]() {
        launch_residual_chain(h->sx.as<float>(), h->sy.as<float>(), h->sz.as<float>(), (int)h->n_src, h->map.keys, h->map.slots, (uint32_t)(h->map.cap - 1),
                              B.st[cur], B.st[cur ^ 1], B.rows[rcur ^ 1], first ? 0 : grid, sp, ep, B.rows[rcur], grid, h->use_links, first ? &init : nullptr,
                              PeerWait{nullptr, 0, 0}, AosIn{nullptr, 0, nullptr, nullptr, nullptr}, h->stream);
        first = false; cur ^= 1; rcur ^= 1;
    };
    for (int i = 0; i < 3; ++i) pass();   // warm
    H_TRY(hipEventRecord(e0, h->stream));
    for (int i = 0; i < reps; ++i) pass();
    H_TRY(hipEventRecord(e1, h->stream));
    H_TRY(hipMemcpyAsync(h->h_state, B.st[cur], sizeof(GnState), hipMemcpyDeviceToHost, h->stream));
    H_TRY(hipEventSynchronize(e1));
    H_TRY(hipStreamSynchronize(h->stream));
    float ms = 0.f;
    H_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (h->h_state->done) { h->err = "the timed loop stopped (too few residuals or a singular Hessian at this pose)"; return h->h_state->status ? h->h_state->status : NDT_ERR_INVALID_ARG; }
    *us_per_launch = 1e3 * (double)ms / reps;
    return NDT_OK;
}

int ndt_time_residual_batch(ndt_handle* h, int reps, double* us_per_launch, double* points_per_launch, double* hits_per_point) {
    if (!h) return NDT_ERR_INVALID_ARG;
    if (!h->b_pairs || h->b_pairs > 32768 || reps < 1 || !us_per_launch) { h->err = "no resident batch (call ndt_align_batch first; at most 32768 pairs)"; return NDT_ERR_INVALID_ARG; }
    H_TRY(hipSetDevice(h->cfg.device));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    H_TRY(hipEventCreate(&e0)); H_TRY(hipEventCreate(&e1));
    const EvalParams ep = eval_params(h);
    const SolveParams sp = solve_params(h);
    // the launches the batched loop makes (one per lane), here back to back on one stream
    const size_t group = h->b_group ? h->b_group : h->b_pairs, n_groups = (h->b_pairs + group - 1) / group;
    auto pass = [&]() {
        for (size_t g = 0; g < n_groups; ++g) {
            const size_t p0 = g * group;
            launch_residual_batch(h->bx.as<float>(), h->by.as<float>(), h->bz.as<float>(), h->boff.as<int64_t>(),
                                  h->b_has_pair_scan ? h->bpair.as<int32_t>() : nullptr, h->map.keys, h->map.slots, (uint32_t)(h->map.cap - 1),
                                  h->bstates.as<GnState>(), ep, h->bpartials.as<double>(), h->b_rows, (int)p0, (int)std::min(group, h->b_pairs - p0), 1,
                                  h->use_links, h->stream);
        }
    };
    for (int i = 0; i < 3; ++i) pass();
    H_TRY(hipEventRecord(e0, h->stream));
    for (int i = 0; i < reps; ++i) pass();
    H_TRY(hipEventRecord(e1, h->stream));
    H_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    H_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *us_per_launch = 1e3 * (double)ms / ((double)reps * (double)n_groups);
    if (points_per_launch) *points_per_launch = h->b_points / (double)n_groups;      // mean over the groups
    if (hits_per_point) {   // measured at the poses the batch ended at
        launch_reduce_solve_batch(h->bpartials.as<double>(), h->b_rows, h->bstates.as<GnState>(), (int)h->b_pairs, sp, 0, 1, h->bdone.as<int32_t>(), h->stream);
        std::vector<GnState> st(h->b_pairs);
        H_TRY(hipMemcpyAsync(st.data(), h->bstates.p, h->b_pairs * sizeof(GnState), hipMemcpyDeviceToHost, h->stream));
        H_TRY(hipStreamSynchronize(h->stream));
        double hits = 0;
        for (auto& g : st) hits += g.acc[ACC_HITS];
        *hits_per_point = h->b_points > 0 ? hits / h->b_points : 0.0;
    }
    return NDT_OK;
}

int ndt_synchronize(ndt_handle* h) {
    if (!h) return NDT_ERR_INVALID_ARG;
    H_TRY(hipSetDevice(h->cfg.device));
    H_TRY(hipStreamSynchronize(h->stream));
    return NDT_OK;
}

const char* ndt_last_error(const ndt_handle* h) { return h ? h->err.c_str() : "null handle"; }

}  // extern "C"
