// ndt_map.hip — target voxel-map build and incremental keyframe merge.
//
// Replaces pcl::VoxelGridCovariance::filter (triggered by ndt_.setInputTarget, src/slam/src/process.cc:78,147)
// and the append + re-downsample + full rebuild of process.cc:134-147 with
//   keys -> LSD radix sort by compact voxel key -> segment heads -> per-segment statistics
//   (ComputeMeanAndCov, src/slam/reference/math_utils.h:55-72) -> find-or-insert in the resident hash ->
//   Gaussian merge (UpdateMeanAndCov, math_utils.h:87-98) -> regularised 3x3 inverse -> 40-byte payload.
// Every reduction has a fixed order (sorted by (key, input index), 8-lane strided sums, xor tree), so the
// map is bit-reproducible from run to run; integer atomics are used only for counters, min/max and the
// hash-slot claim.  All kernels are HBM-bound integer/byte work: coalesced 4/8-byte lanes, no MFMA.
#include "ndt_internal.h"

#include <string.h>

#include <algorithm>
#include <numeric>

#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = std::string(#x) + ": " + hipGetErrorString(e_); return NDT_ERR_HIP; } } while (0)

// ------------------------------------------------------------------ table init / alloc

// Empty table: zero payloads, unset links, empty keys, zero counts.  Neighbouring lanes write neighbouring 16-byte
// pieces (a thread per slot wrote four pieces 64 bytes apart: 1.5 TB/s on a 268 MB table).
static_assert(offsetof(VoxSlot, link) == 40, "k_init_slots writes the links as the last six words of the slot");
__global__ __launch_bounds__(256) void k_init_slots(VoxSlot* s, int64_t* keys, int32_t* counts, size_t cap) {
    const size_t stride = (size_t)gridDim.x * blockDim.x, t0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    int4* q = (int4*)s;
    const int none = (int)NDT_LINK_NONE;
    for (size_t i = t0; i < cap * 4; i += stride) {
        const int piece = (int)(i & 3);
        q[i] = piece == 3 ? make_int4(none, none, none, none) : piece == 2 ? make_int4(0, 0, none, none) : make_int4(0, 0, 0, 0);
    }
    for (size_t i = t0; i < cap + 4; i += stride) keys[i] = NDT_KEY_EMPTY;
    for (size_t i = t0; i < cap; i += stride) counts[i] = 0;
}
static unsigned init_slots_grid(size_t cap) { return (unsigned)std::min<size_t>(4096, (cap * 4 + 255) / 256); }

hipError_t map_alloc(VoxelMap& m, size_t cap, hipStream_t s) {
    hipError_t e;
    if ((e = hipMalloc((void**)&m.slots, cap * sizeof(VoxSlot))) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&m.keys, (cap + 4) * sizeof(int64_t))) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&m.counts, cap * sizeof(int32_t))) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&m.mean, cap * 3 * sizeof(double))) != hipSuccess) return e;
    if ((e = hipMalloc((void**)&m.cov, cap * 6 * sizeof(double))) != hipSuccess) return e;
    m.cap = cap;
    m.n_vox = 0; m.n_entries = 0;
    if (m.track_touch) {
        if ((e = hipMalloc((void**)&m.touch, cap * sizeof(unsigned long long))) != hipSuccess) return e;
        if ((e = hipMemsetAsync(m.touch, 0, cap * sizeof(unsigned long long), s)) != hipSuccess) return e;
    }
    k_init_slots<<<init_slots_grid(cap), 256, 0, s>>>(m.slots, m.keys, m.counts, cap);
    return hipGetLastError();
}
// turns the per-slot touch stamps on or off for a map (ndt_config.capacity); an existing table gets a cleared array
hipError_t map_track_touch(VoxelMap& m, bool on, hipStream_t s) {
    m.track_touch = on;
    if (on && !m.touch && m.cap) {
        hipError_t e;
        if ((e = hipMalloc((void**)&m.touch, m.cap * sizeof(unsigned long long))) != hipSuccess) return e;
        return hipMemsetAsync(m.touch, 0, m.cap * sizeof(unsigned long long), s);
    }
    if (!on && m.touch) {
        hipError_t e = hipStreamSynchronize(s);
        if (e != hipSuccess) return e;
        e = hipFree(m.touch); m.touch = nullptr;
        return e;
    }
    return hipSuccess;
}
// empties the table in place (keeps the allocation: a rebuild of a same-sized map allocates nothing)
hipError_t map_clear(VoxelMap& m, hipStream_t s) {
    m.cloud_no = 0;
    if (!m.cap) return hipSuccess;
    m.n_vox = 0; m.n_entries = 0;
    if (m.touch) { hipError_t e = hipMemsetAsync(m.touch, 0, m.cap * sizeof(unsigned long long), s); if (e != hipSuccess) return e; }
    k_init_slots<<<init_slots_grid(m.cap), 256, 0, s>>>(m.slots, m.keys, m.counts, m.cap);
    return hipGetLastError();
}
void map_free(VoxelMap& m) {
    if (m.slots) (void)hipFree(m.slots);
    if (m.keys) (void)hipFree(m.keys);
    if (m.counts) (void)hipFree(m.counts);
    if (m.mean) (void)hipFree(m.mean);
    if (m.cov) (void)hipFree(m.cov);
    if (m.touch) (void)hipFree(m.touch);
    m = VoxelMap();
}

// ------------------------------------------------------------------ AoS records -> SoA

__global__ void k_unpack_soa(const unsigned char* raw, size_t n, size_t stride, float* x, float* y, float* z) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* p = (const float*)(raw + i * stride);   // records are 4-byte aligned
    x[i] = p[0]; y[i] = p[1]; z[i] = p[2];
}
hipError_t unpack_points_soa(const void* d_raw, size_t n, size_t stride, float* x, float* y, float* z, hipStream_t s) {
    if (!n) return hipSuccess;
    k_unpack_soa<<<(unsigned)((n + 255) / 256), 256, 0, s>>>((const unsigned char*)d_raw, n, stride, x, y, z);
    return hipGetLastError();
}

// ------------------------------------------------------------------ pass 1: world point, packed key, bbox

// misc layout (int32): [0..2] min cell, [3..5] max cell, [6] n_valid, [7] nseg, [8..11] flag sums, [12] big segments
// Same-address global atomics from thousands of blocks serialise (~11 ns each), so every block writes one row
// of partials and a single block folds the rows.
#define TK_MAX_BLOCKS 1024
#define PLAN_SEQ_WORD 63          // word of the pinned 256-byte block that carries the sequence number the host polls
// Input is AoS records (raw, stride) or, when stride == 0, three SoA arrays (raw = x, soa_y, soa_z).
// one input point: AoS record or SoA arrays, optional keyframe transform (pcl::transformPointCloud, process.cc:134-136:
// float64 math, one rounding to float32), PCL's cell rule.  Returns false for a dropped point (non-finite / out of key range).
__device__ __forceinline__ bool load_target_point(const unsigned char* raw, const float* soa_y, const float* soa_z, size_t i, size_t stride, int has_T,
                                                  const double* Rt, const KeyParams& kp, float& x, float& y, float& z, int32_t c[3]) {
    if (stride) { const float* p = (const float*)(raw + i * stride); x = p[0]; y = p[1]; z = p[2]; }
    else { x = ((const float*)raw)[i]; y = soa_y[i]; z = soa_z[i]; }
    bool ok = isfinite(x) && isfinite(y) && isfinite(z);
    if (ok && has_T) {
        double q[3];
        ndt_xform(Rt, Rt + 9, x, y, z, q);
        x = (float)q[0]; y = (float)q[1]; z = (float)q[2];
    }
    return ok && ndt_target_cell(kp, x, y, z, c);
}

// WRITE = false: bounding box and valid count only (the fast path reads the input a second time in k_make_elems32)
// (the keyframe transform travels as a kernel argument: no host-to-device copy in front of the launch; zero_words: the
// radix passes' tile status words, cleared here by the way — one memset command less before the host's wait)
struct RtArg { double v[12]; };
template <bool WRITE>
__global__ __launch_bounds__(256) void k_target_keys(const unsigned char* raw, const float* soa_y, const float* soa_z, size_t n, size_t stride, int has_T, RtArg rt,
                                                     KeyParams kp, float* tx, float* ty, float* tz, int64_t* keys, uint32_t* vals,
                                                     int32_t* block_rows, unsigned long long* zero_words, size_t n_zero) {
    const double* Rt = rt.v;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_zero; i += (size_t)gridDim.x * blockDim.x) zero_words[i] = 0ull;
    __shared__ int32_t smn[3], smx[3], scnt;
    if (threadIdx.x < 3) { smn[threadIdx.x] = INT32_MAX; smx[threadIdx.x] = INT32_MIN; }
    if (threadIdx.x == 0) scnt = 0;
    __syncthreads();
    int32_t okc = 0, mn[3] = {INT32_MAX, INT32_MAX, INT32_MAX}, mx[3] = {INT32_MIN, INT32_MIN, INT32_MIN};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float x, y, z;
        int32_t c[3];
        const bool ok = load_target_point(raw, soa_y, soa_z, i, stride, has_T, Rt, kp, x, y, z, c);
        if (WRITE) {
            tx[i] = x; ty[i] = y; tz[i] = z;
            vals[i] = (uint32_t)i;
            keys[i] = ok ? ndt_pack(c[0], c[1], c[2]) : NDT_KEY_EMPTY;
        }
        if (ok) {
            okc += 1;
#pragma unroll
            for (int j = 0; j < 3; ++j) { mn[j] = min(mn[j], c[j]); mx[j] = max(mx[j], c[j]); }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        okc += __shfl_xor(okc, o);
#pragma unroll
        for (int j = 0; j < 3; ++j) { mn[j] = min(mn[j], __shfl_xor(mn[j], o)); mx[j] = max(mx[j], __shfl_xor(mx[j], o)); }
    }
    if ((threadIdx.x & 63) == 0 && okc) {
        atomicAdd(&scnt, okc);
#pragma unroll
        for (int j = 0; j < 3; ++j) { atomicMin(&smn[j], mn[j]); atomicMax(&smx[j], mx[j]); }
    }
    __syncthreads();
    int32_t* row = block_rows + (size_t)blockIdx.x * 8;
    if (threadIdx.x < 3) { row[threadIdx.x] = smn[threadIdx.x]; row[3 + threadIdx.x] = smx[threadIdx.x]; }
    if (threadIdx.x == 0) { row[6] = scnt; row[7] = 0; }
}
// Also clears what the rest of the call expects to start at zero — the plan words 40..47, the digit totals and tickets of
// the radix passes — and leaves the box and the valid count in pinned host memory (`mirror`) as well, so the host needs
// no memset and no copy command around its one early wait.
__global__ __launch_bounds__(TK_MAX_BLOCKS) void k_bbox_reduce(const int32_t* block_rows, int nrows, int32_t* misc, uint32_t* zero_hist, int n_hist,
                                                               int32_t* mirror, int32_t mirror_seq) {
    for (int i = threadIdx.x; i < n_hist; i += TK_MAX_BLOCKS) zero_hist[i] = 0u;
    if (threadIdx.x >= 40 && threadIdx.x < 48) misc[threadIdx.x] = 0;
    __shared__ int32_t sm[TK_MAX_BLOCKS / 64][8];
    int32_t v[7] = {INT32_MAX, INT32_MAX, INT32_MAX, INT32_MIN, INT32_MIN, INT32_MIN, 0};
    if ((int)threadIdx.x < nrows) {
        const int32_t* row = block_rows + (size_t)threadIdx.x * 8;
        if (row[6]) {
#pragma unroll
            for (int j = 0; j < 7; ++j) v[j] = row[j];
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { v[j] = min(v[j], __shfl_xor(v[j], o)); v[3 + j] = max(v[3 + j], __shfl_xor(v[3 + j], o)); }
        v[6] += __shfl_xor(v[6], o);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int j = 0; j < 7; ++j) sm[threadIdx.x >> 6][j] = v[j];
    }
    __syncthreads();
    if (threadIdx.x < 7) {
        int32_t r = sm[0][threadIdx.x];
        for (int w = 1; w < TK_MAX_BLOCKS / 64; ++w) {
            const int32_t x = sm[w][threadIdx.x];
            r = threadIdx.x < 3 ? min(r, x) : threadIdx.x < 6 ? max(r, x) : r + x;
        }
        misc[threadIdx.x] = r;
        if (mirror) mirror[threadIdx.x] = r;
    }
    if (threadIdx.x == 7) { misc[7] = 0; if (mirror) mirror[7] = 0; }
    if (mirror && threadIdx.x < 64) {          // (the first wave wrote all the mirrored words) the word the host polls
        __threadfence_system();
        if (threadIdx.x == 0) __hip_atomic_store(mirror + PLAN_SEQ_WORD, mirror_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (threadIdx.x == 12) misc[12] = 0;
    if (threadIdx.x == 13) misc[13] = 0;
}

// pass 2: packed key -> compact sort key (bbox-relative, bit-packed); invalid points sort last
// km.mn/bits are listed most-significant axis first; km.axis[] says which cell coordinate each one is
// (x,y,z for the map build; z,y,x for PCL VoxelGrid's output order).
__global__ void k_sort_keys(int64_t* keys, size_t n, SortKeyMap km, uint64_t invalid_sk) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int64_t k = keys[i];
    uint64_t sk = invalid_sk;      // one bit above every valid key: sorts behind the max-corner cell even when all its bits are ones
    if (k != NDT_KEY_EMPTY) {
        int32_t c[3];
        ndt_unpack(k, c[0], c[1], c[2]);
        sk = ((uint64_t)(c[km.axis[0]] - km.mn[0]) << (km.bits[1] + km.bits[2])) | ((uint64_t)(c[km.axis[1]] - km.mn[1]) << km.bits[2]) |
             (uint64_t)(c[km.axis[2]] - km.mn[2]);
    }
    keys[i] = (int64_t)sk;
}
__device__ inline int64_t sortkey_to_packed(uint64_t sk, const SortKeyMap& km) {
    int32_t c[3];
    c[km.axis[2]] = (int32_t)(sk & ((1ull << km.bits[2]) - 1)) + km.mn[2];
    c[km.axis[1]] = (int32_t)((sk >> km.bits[2]) & ((1ull << km.bits[1]) - 1)) + km.mn[1];
    c[km.axis[0]] = (int32_t)(sk >> (km.bits[1] + km.bits[2])) + km.mn[0];
    return ndt_pack(c[0], c[1], c[2]);
}

// ------------------------------------------------------------------ exclusive scan (uint32)

#define SCAN_THREADS 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_THREADS * SCAN_ITEMS)

__device__ inline uint32_t block_exclusive_scan_256(uint32_t v, uint32_t* total) {
    __shared__ uint32_t wsum[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t u = __shfl_up(inc, o);
        if (lane >= o) inc += u;
    }
    __syncthreads();                 // protect wsum across back-to-back calls
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (int j = 0; j < w; ++j) base += wsum[j];
    if (total) *total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    return base + inc - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_reduce(const uint32_t* in, size_t n, uint32_t* bsum) {
    size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) if (base + j < n) s += in[base + j];
    uint32_t tot;
    block_exclusive_scan_256(s, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}
// one block: exclusive scan of bsum[0..nb) in place, grand total to *total
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_spine(uint32_t* bsum, size_t nb, uint32_t* total) {
    uint32_t carry = 0;
    for (size_t base = 0; base < nb; base += SCAN_THREADS) {
        size_t i = base + threadIdx.x;
        uint32_t v = i < nb ? bsum[i] : 0, tot;
        uint32_t ex = block_exclusive_scan_256(v, &tot);
        if (i < nb) bsum[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total) *total = carry;
}
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_down(const uint32_t* in, uint32_t* out, size_t n, const uint32_t* bsum) {
    size_t base = (size_t)blockIdx.x * SCAN_TILE + (size_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) { v[j] = base + j < n ? in[base + j] : 0; s += v[j]; }
    uint32_t ex = block_exclusive_scan_256(s, nullptr) + bsum[blockIdx.x];
#pragma unroll
    for (int j = 0; j < SCAN_ITEMS; ++j) { if (base + j < n) out[base + j] = ex; ex += v[j]; }
}
// small arrays (the radix histograms of a LiDAR scan): the whole exclusive scan in ONE block, one launch instead of three
#define SCAN1_THREADS 1024
#define SCAN1_ITEMS 8
#define SCAN1_MAX (SCAN1_THREADS * SCAN1_ITEMS)    // one tile: beyond that a single CU's strided reads cost more than two extra launches
__global__ __launch_bounds__(SCAN1_THREADS) void k_scan_single(const uint32_t* in, uint32_t* out, size_t n, uint32_t* total) {
    __shared__ uint32_t wsum[SCAN1_THREADS / 64];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (size_t base = 0; base < n; base += (size_t)SCAN1_THREADS * SCAN1_ITEMS) {
        const size_t i0 = base + (size_t)threadIdx.x * SCAN1_ITEMS;
        uint32_t v[SCAN1_ITEMS], sum = 0;
#pragma unroll
        for (int j = 0; j < SCAN1_ITEMS; ++j) { v[j] = i0 + j < n ? in[i0 + j] : 0; sum += v[j]; }
        uint32_t inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { uint32_t u = __shfl_up(inc, o); if (lane >= o) inc += u; }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        uint32_t wbase = 0, tile = 0;
        for (int j = 0; j < SCAN1_THREADS / 64; ++j) { const uint32_t x = wsum[j]; if (j < w) wbase += x; tile += x; }
        uint32_t ex = carry_s + wbase + inc - sum;
#pragma unroll
        for (int j = 0; j < SCAN1_ITEMS; ++j) { if (i0 + j < n) out[i0 + j] = ex; ex += v[j]; }
        __syncthreads();
        if (threadIdx.x == 0) carry_s += tile;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total) *total = carry_s;
}

// out may alias in.  tmp needs ceil(n / SCAN_TILE) + 1 uint32.
static void exclusive_scan_u32(const uint32_t* in, uint32_t* out, size_t n, uint32_t* tmp, uint32_t* total, hipStream_t s) {
    size_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    if (!nb) { if (total) (void)hipMemsetAsync(total, 0, 4, s); return; }
    if (n <= SCAN1_MAX) { k_scan_single<<<1, SCAN1_THREADS, 0, s>>>(in, out, n, total); return; }
    k_scan_reduce<<<(unsigned)nb, SCAN_THREADS, 0, s>>>(in, n, tmp);
    k_scan_spine<<<1, SCAN_THREADS, 0, s>>>(tmp, nb, total);
    k_scan_down<<<(unsigned)nb, SCAN_THREADS, 0, s>>>(in, out, n, tmp);
}

// ------------------------------------------------------------------ LSD radix sort, 8-bit digits, (u64 key, u32 value)

#define RS_THREADS 256
// elements per thread: 16 for big clouds; 4 for a LiDAR scan, so that ~100k points still spread over > 100 blocks
#define RS_ITEMS_BIG 16
#define RS_ITEMS_SMALL 4

// hist layout: digit-major, hist[d * nblk + blk], so one exclusive scan yields every (digit, block) base
template <int RS_ITEMS>
__global__ __launch_bounds__(RS_THREADS) void k_radix_hist(const uint64_t* keys, size_t n, int shift, uint32_t* hist, unsigned nblk) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    size_t base = (size_t)blockIdx.x * (RS_THREADS * RS_ITEMS);
#pragma unroll 4
    for (int r = 0; r < RS_ITEMS; ++r) {
        size_t i = base + (size_t)r * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}

// stable scatter: element order inside a tile is (round, wave, lane) = input order
template <int RS_ITEMS>
__global__ __launch_bounds__(RS_THREADS) void k_radix_scatter(const uint64_t* keys, const uint32_t* vals, size_t n, int shift,
                                                              const uint32_t* offs, unsigned nblk, uint64_t* okeys, uint32_t* ovals) {
    __shared__ uint32_t running[256];
    __shared__ uint32_t wcnt[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    running[threadIdx.x] = offs[(size_t)threadIdx.x * nblk + blockIdx.x];
    size_t base = (size_t)blockIdx.x * (RS_THREADS * RS_ITEMS);
    for (int r = 0; r < RS_ITEMS; ++r) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wcnt[j][threadIdx.x] = 0;
        __syncthreads();
        size_t i = base + (size_t)r * RS_THREADS + threadIdx.x;
        bool act = i < n;
        uint64_t k = act ? keys[i] : 0;
        uint32_t v = act ? vals[i] : 0;
        unsigned d = (unsigned)(k >> shift) & 255u;
        unsigned long long m = __ballot(act);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            unsigned long long bal = __ballot(act && ((d >> b) & 1u));
            m &= ((d >> b) & 1u) ? bal : ~bal;
        }
        unsigned rank = __popcll(m & ((1ull << lane) - 1ull));
        if (act && rank == 0) wcnt[w][d] = __popcll(m);
        __syncthreads();
        if (act) {
            uint32_t pos = running[d] + rank;
            for (int j = 0; j < w; ++j) pos += wcnt[j][d];
            okeys[pos] = k;
            ovals[pos] = v;
        }
        __syncthreads();
        running[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
    }
}

// ------------------------------------------------------------------ segments

__global__ void k_heads(const uint64_t* sk, size_t n, uint32_t* flags) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flags[i] = (i == 0 || sk[i] != sk[i - 1]) ? 1u : 0u;
}
__global__ void k_seg_starts(const uint32_t* flags, const uint32_t* segid, size_t n, uint32_t* seg_start) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flags[i]) seg_start[segid[i]] = (uint32_t)i;
    if (i == n - 1) seg_start[segid[i] + flags[i]] = (uint32_t)n;   // = seg_start[nseg]: end of the last run
}

// 3x3 symmetric eigen-decomposition, cyclic Jacobi (double)
__device__ inline void jacobi3_dev(const double c[6], double lam[3], double V[3][3]) {
    double a[3][3] = {{c[0], c[1], c[2]}, {c[1], c[3], c[4]}, {c[2], c[4], c[5]}};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 16; ++sweep) {
        double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[1][2] * a[1][2];
        double dia = a[0][0] * a[0][0] + a[1][1] * a[1][1] + a[2][2] * a[2][2];
        if (off <= 1e-34 * dia || off == 0.0) break;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0.0) continue;
                double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    double akp = a[k][p], akq = a[k][q];
                    a[k][p] = cs * akp - sn * akq;
                    a[k][q] = sn * akp + cs * akq;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    double apk = a[p][k], aqk = a[q][k];
                    a[p][k] = cs * apk - sn * aqk;
                    a[q][k] = sn * apk + cs * aqk;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = cs * vkp - sn * vkq;
                    V[k][q] = sn * vkp + cs * vkq;
                }
            }
    }
    lam[0] = a[0][0]; lam[1] = a[1][1]; lam[2] = a[2][2];
}

// regularised 3x3 inverse + float32 payload (what the residual kernel reads)
__device__ inline void finalize_slot(VoxSlot* sl, int32_t n, const double mu[3], const double cv[6], const MapBuildCfg& cfg) {
    const int32_t min_pts = cfg.min_pts;
    const double ratio = cfg.eig_floor_ratio;
    float info[6] = {0, 0, 0, 0, 0, 0};
    int32_t valid = 0;
    if (n >= min_pts && cfg.info_rule == NDT_INFO_ADD_1E3) {
        // IncNdt3d's rule: info = (Sigma + 1e-3 I)^-1, a single point gets 100 I.  Plain cofactor
        // inverse in this fixed order (the oracle's finalize_voxel writes the same expressions).
        if (n == 1) {
            info[0] = info[3] = info[5] = 100.f;
            valid = 1;
        } else {
            const double a00 = cv[0] + 1e-3, a01 = cv[1], a02 = cv[2], a11 = cv[3] + 1e-3, a12 = cv[4], a22 = cv[5] + 1e-3;
            const double c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
            const double c11 = a00 * a22 - a02 * a02, c12 = a01 * a02 - a00 * a12, c22 = a00 * a11 - a01 * a01;
            const double det = (a00 * c00 + a01 * c01) + a02 * c02;
            const double v[6] = {c00 / det, c01 / det, c02 / det, c11 / det, c12 / det, c22 / det};
            valid = det > 0.0 && isfinite(det);
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                if (!isfinite(v[t])) valid = 0;
                info[t] = (float)v[t];
            }
        }
    } else if (n >= min_pts) {
        double lam[3], V[3][3];
        jacobi3_dev(cv, lam, V);
        double lmax = fmax(lam[0], fmax(lam[1], lam[2]));
        if (lmax > 0.0 && isfinite(lmax)) {
            double fl = ratio * lmax, il[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) il[j] = 1.0 / (lam[j] < fl ? fl : lam[j]);
            valid = 1;
            int t = 0;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = r; c < 3; ++c, ++t) {
                    double v = (V[r][0] * il[0] * V[c][0] + V[r][1] * il[1] * V[c][1]) + V[r][2] * il[2] * V[c][2];
                    if (!isfinite(v)) valid = 0;
                    info[t] = (float)v;
                }
        }
    }
    sl->off[0] = (float)mu[0]; sl->off[1] = (float)mu[1]; sl->off[2] = (float)mu[2];
    if (!valid) { info[0] = info[1] = info[2] = info[3] = info[4] = info[5] = 0.f; }
    sl->i00 = info[0]; sl->i01 = info[1]; sl->i02 = info[2]; sl->i11 = info[3]; sl->i12 = info[4]; sl->i22 = info[5];
    sl->valid = valid;
}


// One SEG_LANES-lane group per segment (= one voxel of the batch): strided sums in voxel-local float64
// coordinates over the sorted points, xor-tree over the group, then lane 0 claims / finds the hash slot and merges.
// SEG_LANES is 8 for sparse batches and 64 (a whole wave) when voxels hold many points.  The eigen-decomposition and
// the float32 payload (a long serial float64 chain) are left to k_finalize_segments, which runs one voxel per lane
// on every lane instead of on one lane per group here (k_segment_merge<64> on 2 M points: 77 -> see DESIGN.md §5).
// Segments far longer than the rest (the ground voxels next to the trajectory hold tens of thousands of points while
// the average voxel holds a hundred) would each keep one SEG_LANES group busy long after the launch has drained:
// those above `big` points are listed here and summed by a 1024-thread block each (k_big_sums); k_segment_merge
// then takes their sums from big_sums instead of looping.  The list order is arbitrary, each sum is not.
#define BIG_THREADS 1024
__global__ void k_big_list(const uint32_t* seg_start, uint32_t nseg, uint32_t big, uint32_t* big_pos, uint32_t* big_list, uint32_t* big_count) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nseg) return;
    uint32_t pos = 0xFFFFFFFFu;
    if (seg_start[g + 1] - seg_start[g] > big) { pos = atomicAdd(big_count, 1u); big_list[pos] = g; }
    big_pos[g] = pos;
}
__global__ __launch_bounds__(BIG_THREADS) void k_big_sums(const uint32_t* big_list, const uint32_t* big_count, const uint64_t* sk, const uint32_t* seg_start,
                                                          const uint32_t* __restrict__ sv, const float* tx, const float* ty, const float* tz, SortKeyMap km, MapBuildCfg cfg, double* big_sums) {
    if (blockIdx.x >= *big_count) return;
    __shared__ double sm[BIG_THREADS / 64][9];
    const uint32_t g = big_list[blockIdx.x];
    const uint32_t s = seg_start[g], e = seg_start[g + 1];
    int32_t c[3];
    ndt_unpack(sortkey_to_packed(sk[s], km), c[0], c[1], c[2]);
    const double o0 = (double)c[0] * cfg.kp.res_d, o1 = (double)c[1] * cfg.kp.res_d, o2 = (double)c[2] * cfg.kp.res_d;
    double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t i0 = s + threadIdx.x; i0 < e; i0 += 4 * BIG_THREADS) {       // four loads per array in flight, added in index order
        float x[4], y[4], z[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint32_t i = i0 + u * BIG_THREADS;
            const bool in = i < e;
            const uint32_t j = in ? sv[i] : 0u;          // the point's place in the unsorted arrays
            x[u] = in ? tx[j] : 0.f; y[u] = in ? ty[j] : 0.f; z[u] = in ? tz[j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u * BIG_THREADS >= e) break;
            const double d0 = (double)x[u] - o0, d1 = (double)y[u] - o1, d2 = (double)z[u] - o2;
            a[0] += d0; a[1] += d1; a[2] += d2;
            a[3] += d0 * d0; a[4] += d0 * d1; a[5] += d0 * d2;
            a[6] += d1 * d1; a[7] += d1 * d2; a[8] += d2 * d2;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1)
#pragma unroll
        for (int j = 0; j < 9; ++j) a[j] += __shfl_xor(a[j], o);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int j = 0; j < 9; ++j) sm[threadIdx.x >> 6][j] = a[j];
    }
    __syncthreads();
    if (threadIdx.x < 9) {
        double r = sm[0][threadIdx.x];
        for (int w = 1; w < BIG_THREADS / 64; ++w) r += sm[w][threadIdx.x];
        big_sums[(size_t)blockIdx.x * 9 + threadIdx.x] = r;
    }
}

template <int SEG_LANES>
__global__ __launch_bounds__(256) void k_segment_merge(const uint64_t* sk, const uint32_t* seg_start, uint32_t nseg, const uint32_t* __restrict__ sv,
                                                       const float* tx, const float* ty, const float* tz, SortKeyMap km, MapBuildCfg cfg,
                                                       int64_t* keys, int32_t* counts, double* mean, double* cov, uint64_t mask, uint32_t* fresh_flag,
                                                       uint32_t* entry_flag, uint32_t* seg_slot, const uint32_t* big_pos, const double* big_sums) {
    uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) / SEG_LANES;
    const int l = threadIdx.x % SEG_LANES;
    bool act = g < nseg;
    uint32_t s = 0, e = 0, bp = 0xFFFFFFFFu;
    int64_t key = 0;
    double org[3] = {0, 0, 0};
    if (act) {
        s = seg_start[g]; e = seg_start[g + 1]; bp = big_pos[g];
        key = sortkey_to_packed(sk[s], km);
        int32_t c[3];
        ndt_unpack(key, c[0], c[1], c[2]);
#pragma unroll
        for (int j = 0; j < 3; ++j) org[j] = (double)c[j] * cfg.kp.res_d;
    }
    double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t e_loop = bp == 0xFFFFFFFFu ? e : s;          // a big segment was summed by k_big_sums
    for (uint32_t i = s + l; i < e_loop; i += SEG_LANES) {
        const uint32_t j = sv[i];
        double d0 = (double)tx[j] - org[0], d1 = (double)ty[j] - org[1], d2 = (double)tz[j] - org[2];
        a[0] += d0; a[1] += d1; a[2] += d2;
        a[3] += d0 * d0; a[4] += d0 * d1; a[5] += d0 * d2;
        a[6] += d1 * d1; a[7] += d1 * d2; a[8] += d2 * d2;
    }
#pragma unroll
    for (int o = SEG_LANES / 2; o >= 1; o >>= 1)
#pragma unroll
        for (int j = 0; j < 9; ++j) a[j] += __shfl_xor(a[j], o);
    if (!act || l != 0) return;
    if (bp != 0xFFFFFFFFu) {
#pragma unroll
        for (int j = 0; j < 9; ++j) a[j] = big_sums[(size_t)bp * 9 + j];
    }

    const int32_t nb = (int32_t)(e - s);
    double mu[3] = {a[0] / nb, a[1] / nb, a[2] / nb};
    double cv[6] = {0, 0, 0, 0, 0, 0};
    if (nb > 1) {   // ComputeMeanAndCov: sample covariance over (n-1)
        double dn = (double)(nb - 1);
        cv[0] = (a[3] - a[0] * mu[0]) / dn; cv[1] = (a[4] - a[0] * mu[1]) / dn; cv[2] = (a[5] - a[0] * mu[2]) / dn;
        cv[3] = (a[6] - a[1] * mu[1]) / dn; cv[4] = (a[7] - a[1] * mu[2]) / dn; cv[5] = (a[8] - a[2] * mu[2]) / dn;
    }
    // find-or-insert (capacity was sized by the host so that the load factor stays <= 1/2)
    uint64_t slot = ndt_hash_key(key) & mask;
    bool fresh = false;
    for (;;) {
        unsigned long long prev = atomicCAS((unsigned long long*)&keys[slot], (unsigned long long)NDT_KEY_EMPTY, (unsigned long long)key);
        if ((int64_t)prev == NDT_KEY_EMPTY) { fresh = true; break; }
        if ((int64_t)prev == key) break;
        slot = (slot + 1) & mask;
    }
    seg_slot[g] = (uint32_t)slot;
    double* mp = mean + slot * 3;
    double* cp = cov + slot * 6;
    int32_t ntot = nb;
    const int32_t had = fresh ? 0 : counts[slot];            // 0 also for a ghost cell that now gets its first points
    entry_flag[g] = fresh ? 1u : 0u;    // new table entry        } summed afterwards: a same-address atomic per voxel
    fresh_flag[g] = had == 0 ? 1u : 0u; // new voxel with points  } would serialise
    if (fresh && slot < 4) keys[mask + 1 + slot] = key;      // mirror entries
    if (had != 0 && cfg.max_pts > 0 && had > cfg.max_pts) {      // frozen (IncNdt3d): the voxel keeps what it has
#pragma unroll
        for (int j = 0; j < 3; ++j) mu[j] = mp[j];
#pragma unroll
        for (int j = 0; j < 6; ++j) cv[j] = cp[j];
        ntot = had;
    } else if (had != 0) {        // UpdateMeanAndCov (math_utils.h:87-98)
        double hm = (double)had, cn = (double)nb, tot = hm + cn;
        double nm[3], dh[3], dc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            nm[j] = (hm * mp[j] + cn * mu[j]) / tot;
            dh[j] = mp[j] - nm[j];
            dc[j] = mu[j] - nm[j];
        }
        int t = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = r; c < 3; ++c, ++t) cv[t] = (hm * (cp[t] + dh[r] * dh[c]) + cn * (cv[t] + dc[r] * dc[c])) / tot;
#pragma unroll
        for (int j = 0; j < 3; ++j) mu[j] = nm[j];
        ntot = had + nb;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) mp[j] = mu[j];
#pragma unroll
    for (int j = 0; j < 6; ++j) cp[j] = cv[j];
    counts[slot] = ntot;
}
// one voxel of the batch per lane: regularised inverse + payload from the merged statistics
__global__ __launch_bounds__(256) void k_finalize_segments(const uint32_t* seg_slot, uint32_t nseg, MapBuildCfg cfg, VoxSlot* slots,
                                                           const int32_t* counts, const double* mean, const double* cov) {
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= nseg) return;
    const size_t slot = seg_slot[g];
    double mu[3], cv[6];
#pragma unroll
    for (int j = 0; j < 3; ++j) mu[j] = mean[slot * 3 + j];
#pragma unroll
    for (int j = 0; j < 6; ++j) cv[j] = cov[slot * 6 + j];
    finalize_slot(&slots[slot], counts[slot], mu, cv, cfg);
}

// pcl::VoxelGrid centroid (src/slam/src/process.cc:68-73): mean of the cell's points, float64 sum in sorted order,
// one rounding to float32; output row = segment index (cells in PCL's index order when sorted z,y,x)
template <int SEG_LANES>
__global__ __launch_bounds__(256) void k_segment_centroid(const uint32_t* seg_start, const uint32_t* d_nseg, const uint32_t* __restrict__ sv, const float* gx,
                                                          const float* gy, const float* gz, float* ox, float* oy, float* oz) {
    const uint32_t nseg = *d_nseg;      // the launch covers the upper bound n_valid; groups past nseg have nothing to do
    uint32_t g = (blockIdx.x * blockDim.x + threadIdx.x) / SEG_LANES;
    const int l = threadIdx.x % SEG_LANES;
    uint32_t s = 0, e = 0;
    if (g < nseg) { s = seg_start[g]; e = seg_start[g + 1]; }
    double a[3] = {0, 0, 0};
    for (uint32_t i = s + l; i < e; i += SEG_LANES) { const uint32_t j = sv[i]; a[0] += (double)gx[j]; a[1] += (double)gy[j]; a[2] += (double)gz[j]; }
#pragma unroll
    for (int o = SEG_LANES / 2; o >= 1; o >>= 1)
#pragma unroll
        for (int j = 0; j < 3; ++j) a[j] += __shfl_xor(a[j], o);
    if (g < nseg && l == 0) {
        const double nn = (double)(e - s);
        ox[g] = (float)(a[0] / nn); oy[g] = (float)(a[1] / nn); oz[g] = (float)(a[2] / nn);
    }
}

// ------------------------------------------------------------------ rehash (capacity growth)

__global__ void k_rehash(const int64_t* ok, const VoxSlot* os, const int32_t* on, const double* om, const double* oc, size_t ocap,
                         VoxSlot* ns, int64_t* nk, int32_t* nn, double* nm, double* nc, uint64_t nmask, const unsigned long long* ot,
                         unsigned long long* nt) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ocap) return;
    const int64_t key = ok[i];
    if (key == NDT_KEY_EMPTY) return;
    uint64_t slot = ndt_hash_key(key) & nmask;
    for (;;) {
        unsigned long long prev = atomicCAS((unsigned long long*)&nk[slot], (unsigned long long)NDT_KEY_EMPTY, (unsigned long long)key);
        if ((int64_t)prev == NDT_KEY_EMPTY) break;
        slot = (slot + 1) & nmask;
    }
    if (slot < 4) nk[nmask + 1 + slot] = key;
    VoxSlot v = os[i];
#pragma unroll
    for (int j = 0; j < 6; ++j) v.link[j] = NDT_LINK_NONE;      // slots moved: the links are rebuilt by k_link_all
    ns[slot] = v;
    nn[slot] = on[i];
    for (int j = 0; j < 3; ++j) nm[slot * 3 + j] = om[i * 3 + j];
    for (int j = 0; j < 6; ++j) nc[slot * 6 + j] = oc[i * 6 + j];
    if (ot) nt[slot] = ot[i];
}

// ------------------------------------------------------------------ ghost cells and neighbour links
// A point's residuals come from its own cell and the six face neighbours.  So that only the point's own cell has
// to be hashed, every cell that touches a voxel is in the table too (a "ghost": no points, valid = 0) and every
// entry stores the slots of its six neighbours.

__device__ inline uint32_t table_find(const int64_t* keys, uint64_t mask, int64_t key) {
    uint64_t slot = ndt_hash_key(key) & mask;
    for (;;) {
        const int64_t k = keys[slot];
        if (k == key) return (uint32_t)slot;
        if (k == NDT_KEY_EMPTY) return NDT_LINK_NONE;
        slot = (slot + 1) & mask;
    }
}
__device__ inline bool neighbour_key(int64_t key, int d, int64_t* out) {   // d: -x +x -y +y -z +z; false if out of key range
    int32_t c[3];
    ndt_unpack(key, c[0], c[1], c[2]);
    c[d >> 1] += (d & 1) ? 1 : -1;
    if ((uint32_t)(c[d >> 1] + NDT_KEY_BIAS) >= (uint32_t)(2 * NDT_KEY_BIAS)) return false;
    *out = ndt_pack(c[0], c[1], c[2]);
    return true;
}
__global__ void k_link_all(const int64_t* keys, VoxSlot* slots, size_t cap) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    const int64_t key = keys[i];
    if (key == NDT_KEY_EMPTY) return;
    uint32_t l[6];
#pragma unroll
    for (int d = 0; d < 6; ++d) {
        int64_t nk;
        l[d] = neighbour_key(key, d, &nk) ? table_find(keys, cap - 1, nk) : NDT_LINK_NONE;
    }
#pragma unroll
    for (int d = 0; d < 6; ++d) slots[i].link[d] = l[d];
}

static size_t pow2_at_least(size_t v) { size_t c = 1024; while (c < v) c <<= 1; return c; }

// want_entries counts occupied slots (voxels + ghost cells).  Growing moves every entry, so the neighbour links are
// rebuilt over the whole new table (k_link_all): the only table-wide pass, and only when the capacity changes.
static int map_reserve(VoxelMap& m, size_t want_entries, hipStream_t s, std::string& err) {
    size_t need = pow2_at_least(want_entries * 2);       // load factor <= 1/2
    if (m.cap >= need && !(m.n_entries == 0 && m.cap > 16 * need)) return NDT_OK;   // an empty, far too large table is re-made
    const bool track = m.track_touch;
    const uint32_t cloud_no = m.cloud_no;
    if (m.n_entries == 0 && m.cap) { HIP_TRY(hipStreamSynchronize(s)); map_free(m); }
    size_t ncap = pow2_at_least(want_entries * 4);        // grow with headroom
    VoxelMap nm;
    nm.track_touch = track; nm.cloud_no = cloud_no;
    HIP_TRY(map_alloc(nm, ncap, s));
    if (m.cap) {
        k_rehash<<<(unsigned)((m.cap + 255) / 256), 256, 0, s>>>(m.keys, m.slots, m.counts, m.mean, m.cov, m.cap, nm.slots, nm.keys, nm.counts, nm.mean,
                                                                nm.cov, ncap - 1, m.touch, nm.touch);
        k_link_all<<<(unsigned)((ncap + 255) / 256), 256, 0, s>>>(nm.keys, nm.slots, ncap);
        HIP_TRY(hipGetLastError());
        nm.n_vox = m.n_vox; nm.n_entries = m.n_entries;
        HIP_TRY(hipStreamSynchronize(s));
        map_free(m);
    }
    m = nm;
    return NDT_OK;
}

static int bits_for(int64_t range) { int b = 0; while ((1ll << b) < range) ++b; return b; }

// ------------------------------------------------------------------ plan / counter words
// Everything the stages after the bounding box decide is decided on the DEVICE and read back once, at the end of the call:
// w.misc (int32 words): [0..5] bbox, [6] n_valid, [7] nseg, [8..11] flag sums (64-bit-key path), [12] big segments,
// [13] one-sweep sort error, [16..39] keyframe transform, [40..47] the words below.  The map's own counters (voxels,
// table entries) are the host's (VoxelMap::n_vox / n_entries, exact after every call's read-back) and reach the kernels
// as arguments.
#define PL_NSEG 7
#define PL_BIGCOUNT 12
#define PL_SORT_ERR 13   // a one-sweep pass gave up waiting for a predecessor tile (never seen; the host then sorts with the three-kernel passes)
#define PL_ABORT1 42     // the batch's voxels may not fit the table: the insert stages did nothing, the host grows the table and re-runs them
#define PL_ABORT2 43     // the ghost cells may not fit: the ghost stage did nothing, the host grows the table and re-runs it
#define PL_HUGECOUNT 44  // segments too long for one wave (summed by a block each)
#define PL_NEW_VOX 45    // voxels that got their first points (= length of the fresh list)
#define PL_NEW_ENT 46    // table entries created for voxels
#define PL_NEW_GHOST 47  // table entries created for ghost cells

// block-wide sum of a per-thread count, added to *dst by one atomic per block (same-address atomics serialise)
__device__ __forceinline__ void block_count_add(uint32_t v, int32_t* dst) {
    __shared__ uint32_t s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(&s_cnt, v);
    __syncthreads();
    if (threadIdx.x == 0 && s_cnt) atomicAdd((uint32_t*)dst, s_cnt);
}

// ------------------------------------------------------------------ fast path: compact keys of <= 32 bits
// One 16-byte element {sort key, x, y, z} per point travels through the radix passes (one dwordx4 per lane), so the
// per-voxel sums afterwards read their points as a contiguous stream instead of gathering them through a sorted index.
// Each pass is ONE kernel (histogram, chained scan over the tiles with look-back, scatter), after the digit totals of
// all passes were counted while the elements were made.

// lanes of the wave that hold the same 8-bit digit as this lane (for active lanes)
__device__ __forceinline__ unsigned long long same_digit_lanes(unsigned d, bool act) {
    unsigned long long m = __ballot(act);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const unsigned long long bal = __ballot(act && ((d >> b) & 1u));
        m &= ((d >> b) & 1u) ? bal : ~bal;
    }
    return m;
}
// LDS histogram add with RUNS of same-digit lanes combined first (a LiDAR ring's consecutive points share their digit:
// plain LDS atomics would serialise 64 ways).  Only the first lane of a run adds, the run's length: one lane shift, two
// ballots and a find-first-set instead of the nine ballots of a full same-digit match (round 3: the element kernel spent
// more on that match, four digits per point, than on its memory traffic).  Separate runs of one digit in a wave meet in the
// LDS atomic unit, which is exact either way.  Inactive lanes are a suffix of the wave (act = index < n).
__device__ __forceinline__ void wave_hist_add(uint32_t* h, unsigned d, bool act) {
    const unsigned long long am = __ballot(act);
    if (!am) return;
    const int lane = threadIdx.x & 63;
    const unsigned d0 = (unsigned)__builtin_amdgcn_readfirstlane((int)d);      // any lane's digit will do for the test below
    if (!__any(act && d != d0)) {          // all active lanes share one digit (the usual case for the higher digits): one add
        if (lane == __ffsll((long long)am) - 1) atomicAdd(&h[d0], (uint32_t)__popcll(am));
        return;
    }
    const unsigned dk = act ? d : 0xFFFFFFFFu;
    const unsigned prev = __shfl_up(dk, 1);
    const bool head = act && (lane == 0 || prev != dk);
    const unsigned long long stops = __ballot(head) | ~am;                   // run heads and the inactive tail
    const unsigned long long rest = lane == 63 ? 0ull : stops >> (lane + 1);
    const uint32_t run = rest ? (uint32_t)__ffsll((long long)rest) : (uint32_t)(64 - lane);
    if (head) atomicAdd(&h[d], run);
}

#define OS_ITEMS 16                  // elements per thread of a tile: 4096 per tile; small batches use OS_ITEMS_SMALL (1024 per tile)
#define OS_ITEMS_SMALL 4
#define OS_SMALL_N (1u << 19)         // below this many elements a keyframe-sized batch: more, shorter tiles
#define OS_TILE (RS_THREADS * OS_ITEMS)
static inline int os_items_for(size_t n) { return n < OS_SMALL_N ? OS_ITEMS_SMALL : OS_ITEMS; }
#define OS_MAX_PASSES 4
#define GH_COPIES 8                 // copies of the digit totals the blocks add into (by block index): 977 same-address adds in a row cost 15 us
#define GH_STRIDE (OS_MAX_PASSES * 256)
#define ME_MAX_BLOCKS 1024                      // blocks of k_make_elems32: bounds the same-address adds into the digit totals

// elements + digit totals of every pass (ghist[p][256]); blocks stride over the tiles
__global__ __launch_bounds__(RS_THREADS) void k_make_elems32(const unsigned char* raw, const float* soa_y, const float* soa_z, size_t n, size_t stride, int has_T,
                                                             RtArg rt, KeyParams kp, SortKeyMap km, uint32_t invalid_sk, int passes, uint4* elems,
                                                             uint32_t* ghist) {
    const double* Rt = rt.v;
    __shared__ uint32_t h[OS_MAX_PASSES][256];
#pragma unroll
    for (int p = 0; p < OS_MAX_PASSES; ++p) h[p][threadIdx.x] = 0;
    __syncthreads();
    // four points per thread in flight; blocks stride over the input
    for (size_t base = (size_t)blockIdx.x * (RS_THREADS * 4); base < n; base += (size_t)gridDim.x * (RS_THREADS * 4)) {      // (uniform over the block)
        float x[4], y[4], z[4];
        int32_t c[4][3];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = base + (size_t)u * RS_THREADS + threadIdx.x;
            ok[u] = i < n && load_target_point(raw, soa_y, soa_z, i, stride, has_T, Rt, kp, x[u], y[u], z[u], c[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t i = base + (size_t)u * RS_THREADS + threadIdx.x;
            const bool act = i < n;
            uint32_t sk = invalid_sk;
            if (ok[u])
                sk = ((uint32_t)(c[u][km.axis[0]] - km.mn[0]) << (km.bits[1] + km.bits[2])) | ((uint32_t)(c[u][km.axis[1]] - km.mn[1]) << km.bits[2]) |
                     (uint32_t)(c[u][km.axis[2]] - km.mn[2]);
            if (act) elems[i] = make_uint4(sk, __float_as_uint(x[u]), __float_as_uint(y[u]), __float_as_uint(z[u]));
            for (int p = 0; p < passes; ++p) wave_hist_add(h[p], (sk >> (8 * p)) & 255u, act);
        }
    }
    __syncthreads();
    for (int p = 0; p < passes; ++p)
        if (h[p][threadIdx.x]) atomicAdd(&ghist[(blockIdx.x % GH_COPIES) * GH_STRIDE + p * 256 + threadIdx.x], h[p][threadIdx.x]);
}

// One radix pass in one kernel.  A block takes the next tile (ticket), counts its digits, publishes the counts, adds up
// the counts of the tiles before it by looking back (a tile's word per digit is {flag, count}: its own count first, the
// inclusive prefix once known; 8-byte agent-scope atomic stores and loads, every word self-contained), then scatters its
// elements — kept in registers — stably: (round, wave, lane) = input order.  Tiles are numbered in ticket order, so
// every tile a block waits for has been started; the wait is bounded all the same and raises *err when it runs out.
#define OS_AGG (1ull << 62)
#define OS_PRE (2ull << 62)
#define OS_VAL ((1ull << 62) - 1ull)
#define OS_SPIN_LIMIT (1 << 22)
// Diagnostic build only (-DNDT_STAMPS, tools/exp_sort_stamps.py): the device-wide clock at the phase boundaries of every
// tile of a pass.  The product library is built without it.
#ifdef NDT_STAMPS
#define OS_STAMP_TILES 8192
__device__ unsigned long long g_os_stamps[OS_STAMP_TILES][8];
#define OS_STAMP(k) do { if (threadIdx.x == 0 && tile < OS_STAMP_TILES) { g_os_stamps[tile][k] = wall_clock64(); if (k == 1) g_os_stamps[tile][0] = t_entry; } } while (0)
extern "C" int ndt_debug_os_stamps(unsigned long long* out, int tiles) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_os_stamps), (size_t)tiles * 64); }
#else
#define OS_STAMP(k) do { } while (0)
#endif
template <int ITEMS>
__global__ __launch_bounds__(RS_THREADS) void k_onesweep32(const uint4* __restrict__ in, uint4* __restrict__ out, uint32_t n, int shift,
                                                           const uint32_t* __restrict__ ghist, unsigned long long* status, uint32_t* ticket, int32_t* err) {
    __shared__ uint32_t s_tile;
    __shared__ uint32_t wcnt[4][256];      // per wave: running count of each digit, then the wave's first output position of it
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#ifdef NDT_STAMPS
    const unsigned long long t_entry = wall_clock64();
#endif
    // (ticket == nullptr: every tile of the launch is resident at once — the host checked — so any order is free of deadlock and
    // the tiles need not queue for a number: 15 ns each, 7 us for the last of 488)
    if (threadIdx.x == 0) s_tile = ticket ? atomicAdd(ticket, 1u) : blockIdx.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) wcnt[j][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t tile = s_tile;
    OS_STAMP(1);
    // element order inside a tile: (wave, round, lane) — every wave owns 1024 consecutive elements, so ranking needs no
    // block barrier per round
    const size_t base = (size_t)tile * (RS_THREADS * ITEMS) + (size_t)w * (64 * ITEMS) + lane;
    uint4 e[ITEMS];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const size_t i = base + (size_t)r * 64;
        e[r] = i < n ? in[i] : make_uint4(0, 0, 0, 0);
    }
    uint32_t rank[ITEMS];               // rank of the element among the wave's elements of the same digit, in element order
    volatile uint32_t* wc = wcnt[w];
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const bool act = base + (size_t)r * 64 < n;
        const unsigned d = (e[r].x >> shift) & 255u;
        const unsigned long long am = __ballot(act);
        const unsigned d0 = (unsigned)__builtin_amdgcn_readfirstlane((int)d);
        unsigned long long m;
        if (!__any(act && d != d0)) m = am;                       // one digit in the whole wave: the usual case for the higher digits
        else m = same_digit_lanes(d, act);
        const uint32_t before = act ? wc[d] : 0u;
        rank[r] = before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        __builtin_amdgcn_wave_barrier();
        if (act && (m & ((1ull << lane) - 1ull)) == 0ull) wc[d] = before + (uint32_t)__popcll(m);
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    OS_STAMP(2);
    // this thread owns digit threadIdx.x: global base of the digit + what the earlier tiles hold of it
    uint32_t total = 0;
#pragma unroll
    for (int cpy = 0; cpy < GH_COPIES; ++cpy) total += ghist[cpy * GH_STRIDE + threadIdx.x];
    const uint32_t excl = block_exclusive_scan_256(total, nullptr);
    const uint32_t c0 = wcnt[0][threadIdx.x], c1 = wcnt[1][threadIdx.x], c2 = wcnt[2][threadIdx.x], c3 = wcnt[3][threadIdx.x];
    const unsigned long long cnt = (unsigned long long)c0 + c1 + c2 + c3;
    unsigned long long* mine = status + (size_t)tile * 256 + threadIdx.x;
    unsigned long long before = 0;
    if (tile == 0) {
        __hip_atomic_store(mine, OS_PRE | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        __hip_atomic_store(mine, OS_AGG | cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // look back eight tiles at a time (eight loads in flight: a walk of one dependent load per tile would cost the
        // later tiles of a small launch a microsecond per predecessor)
        int spins = 0;
        bool done = false;
        for (uint32_t t1 = tile; t1 > 0 && !done;) {
            const uint32_t nb = t1 < 8u ? t1 : 8u;
            unsigned long long v[8];
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j)
                v[j] = j < nb ? __hip_atomic_load(status + (size_t)(t1 - 1 - j) * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
            for (uint32_t j = 0; j < 8; ++j) {
                if (j >= nb || done) break;
                while (!(v[j] >> 62)) {               // not published yet: wait for this one (bounded)
                    if (++spins > OS_SPIN_LIMIT) { *err = 1; v[j] = OS_PRE; break; }
                    __builtin_amdgcn_s_sleep(1);
                    v[j] = __hip_atomic_load(status + (size_t)(t1 - 1 - j) * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                before += v[j] & OS_VAL;
                if ((v[j] >> 62) == 2ull) done = true;
            }
            t1 -= nb;
        }
        __hip_atomic_store(mine, OS_PRE | (before + cnt), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    OS_STAMP(3);
    __syncthreads();                       // (block_exclusive_scan_256's shared words and every wave's counts are read)
    OS_STAMP(4);
    const uint32_t p0 = excl + (uint32_t)before;
    wcnt[0][threadIdx.x] = p0; wcnt[1][threadIdx.x] = p0 + c0; wcnt[2][threadIdx.x] = p0 + c0 + c1; wcnt[3][threadIdx.x] = p0 + c0 + c1 + c2;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        if (base + (size_t)r * 64 < n) {
            const uint32_t pos = wcnt[w][(e[r].x >> shift) & 255u] + rank[r];
            if (pos < n) out[pos] = e[r];        // (always, unless a wait ran out)
        }
    }
    OS_STAMP(5);
}

// the three-kernel form of a pass (histograms per tile, exclusive scan, scatter): the fallback of k_onesweep32
template <int RS_ITEMS>
__global__ __launch_bounds__(RS_THREADS) void k_radix_hist32(const uint4* elems, size_t n, int shift, uint32_t* hist, unsigned nblk) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * (RS_THREADS * RS_ITEMS);
    for (int r = 0; r < RS_ITEMS; ++r) {
        const size_t i = base + (size_t)r * RS_THREADS + threadIdx.x;
        const bool act = i < n;
        const unsigned d = act ? (elems[i].x >> shift) & 255u : 0u;
        wave_hist_add(h, d, act);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblk + blockIdx.x] = h[threadIdx.x];
}
template <int RS_ITEMS>
__global__ __launch_bounds__(RS_THREADS) void k_radix_scatter32(const uint4* elems, size_t n, int shift, const uint32_t* offs, unsigned nblk, uint4* out) {
    __shared__ uint32_t running[256];
    __shared__ uint32_t wcnt[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    running[threadIdx.x] = offs[(size_t)threadIdx.x * nblk + blockIdx.x];
    const size_t base = (size_t)blockIdx.x * (RS_THREADS * RS_ITEMS);
    for (int r = 0; r < RS_ITEMS; ++r) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wcnt[j][threadIdx.x] = 0;
        __syncthreads();
        const size_t i = base + (size_t)r * RS_THREADS + threadIdx.x;
        const bool act = i < n;
        uint4 e = make_uint4(0, 0, 0, 0);
        if (act) e = elems[i];
        const unsigned d = (e.x >> shift) & 255u;
        const unsigned long long m = same_digit_lanes(d, act);
        const unsigned rank = __popcll(m & ((1ull << lane) - 1ull));
        if (act && rank == 0) wcnt[w][d] = __popcll(m);
        __syncthreads();
        if (act) {
            uint32_t pos = running[d] + rank;
            for (int j = 0; j < w; ++j) pos += wcnt[j][d];
            out[pos] = e;
        }
        __syncthreads();
        running[threadIdx.x] += wcnt[0][threadIdx.x] + wcnt[1][threadIdx.x] + wcnt[2][threadIdx.x] + wcnt[3][threadIdx.x];
    }
}

// segment heads of the sorted valid prefix, enumerated in arbitrary order (one returning atomic per 4096 elements): the
// order of the voxels of a batch does not matter, the order of the points inside a voxel (input order) is what the sums fix
#define SH_THREADS 1024
#define SH_ITEMS 4
__global__ __launch_bounds__(SH_THREADS) void k_seg_heads32(const uint4* elems, uint32_t n_valid, uint32_t* seg_pos, int32_t* plan) {
    __shared__ uint32_t wtot[SH_THREADS / 64];
    __shared__ uint32_t base_s;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const uint32_t* keys = (const uint32_t*)elems;          // key of element i = word 4 i
    const uint32_t base = blockIdx.x * (SH_THREADS * SH_ITEMS);
    unsigned long long b[SH_ITEMS];
    uint32_t tot = 0;
#pragma unroll
    for (int j = 0; j < SH_ITEMS; ++j) {
        const uint32_t i = base + j * SH_THREADS + threadIdx.x;
        const bool head = i < n_valid && (i == 0 || keys[4 * (size_t)i] != keys[4 * (size_t)i - 4]);
        b[j] = __ballot(head);
        tot += (uint32_t)__popcll(b[j]);
    }
    if (lane == 0) wtot[w] = tot;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int j = 0; j < SH_THREADS / 64; ++j) t += wtot[j];
        base_s = t ? atomicAdd((uint32_t*)&plan[PL_NSEG], t) : 0u;
    }
    __syncthreads();
    uint32_t pos = base_s;
    for (int j = 0; j < w; ++j) pos += wtot[j];
#pragma unroll
    for (int j = 0; j < SH_ITEMS; ++j) {
        if ((b[j] >> lane) & 1ull) seg_pos[pos + (uint32_t)__popcll(b[j] & ((1ull << lane) - 1ull))] = base + j * SH_THREADS + threadIdx.x;
        pos += (uint32_t)__popcll(b[j]);
    }
}

// what the segment kernels derive from the segment count, each for itself (no planning kernel in between)
// 64 lanes per segment when voxels hold many points, else 8 (eight segments per wave side by side: at 25 points per voxel
// the 20 M-point map's 815 k segments took 0.99 ms one wave each, the per-segment chain of dependent loads being the cost)
#ifndef SEG_WIDE_MIN
#define SEG_WIDE_MIN 256
#endif
__device__ __forceinline__ bool seg_wide(uint32_t nseg, uint32_t n_valid) { return nseg && n_valid / nseg >= SEG_WIDE_MIN; }
#ifndef SEG_BIG_WIDE
#define SEG_BIG_WIDE 2048u
#endif
__device__ __forceinline__ uint32_t seg_big(uint32_t nseg, uint32_t n_valid) { return seg_wide(nseg, n_valid) ? SEG_BIG_WIDE : 256u; }   // <= 32 trips of a group (8 lanes), <= 8 (64 lanes)

// end of every segment (8 lanes per segment: exponential then 8-ary search on the sorted keys) + the list of big ones
#define SEG_NONE 0xFFFFFFFFu
#define HUGE_SEG 4096u        // 8-lane mode: up to this many points a big segment is summed by one wave (16 trips), beyond it by a block
__global__ __launch_bounds__(256) void k_seg_ends32(const uint4* elems, uint32_t n_valid, const uint32_t* seg_pos, int32_t* plan, uint32_t* seg_end,
                                                   uint32_t* big_pos, uint32_t* big_list, uint32_t list_cap, uint32_t* big_count) {
    const uint32_t nseg = (uint32_t)plan[PL_NSEG], big = seg_big(nseg, n_valid);
    // (64-lane mode: everything above 2048 points goes to a block: a lone wave's 16 trips cost more than the block's barrier)
    const uint32_t huge = seg_wide(nseg, n_valid) ? big : HUGE_SEG;
    const uint32_t* keys = (const uint32_t*)elems;
    const int lane = threadIdx.x & 63, l = threadIdx.x & 7, gsh = lane & ~7;
    const uint32_t groups = gridDim.x * 32u;
    for (uint32_t g0 = (blockIdx.x * 256u + (threadIdx.x & ~63u)) / 8u; g0 < nseg; g0 += groups) {     // g0: first group of this wave
        const uint32_t g = g0 + (uint32_t)(lane >> 3);
        const bool act = g < nseg;
        const uint32_t s = act ? seg_pos[g] : 0u;
        const uint32_t key = act ? keys[4 * (size_t)s] : 0u;
        uint64_t lo = s, stride = 1;           // elems[lo] holds the key; grows while all eight probes match
        bool grow = act;
        while (__any(grow)) {
            const uint64_t p = lo + (uint64_t)(l + 1) * stride;
            const bool eq = grow && p < n_valid && keys[4 * p] == key;
            const unsigned k = __popc((unsigned)((__ballot(eq) >> gsh) & 0xFFull));    // sorted: matches are a prefix of the probes
            if (grow) { lo += k * stride; if (k < 8) grow = false; else stride *= 8; }
        }
        bool refine = act && stride > 1;       // the end lies in (lo, lo + stride]
        while (__any(refine)) {
            if (refine) stride >>= 3;
            const uint64_t p = lo + (uint64_t)(l + 1) * stride;
            const bool eq = refine && l < 7 && p < n_valid && keys[4 * p] == key;
            const unsigned k = __popc((unsigned)((__ballot(eq) >> gsh) & 0xFFull));
            if (refine) { lo += k * stride; if (stride == 1) refine = false; }
        }
        if (act && l == 0) {
            const uint32_t e = (uint32_t)lo + 1u;
            seg_end[g] = e;
            uint32_t pos = SEG_NONE;
            // big segments: listed from the front; those beyond one wave's reach from the back of the same array
            if (e - s > huge) { pos = atomicAdd((uint32_t*)&plan[PL_HUGECOUNT], 1u); big_list[list_cap - 1u - pos] = g; }
            else if (e - s > big) { pos = atomicAdd(big_count, 1u); big_list[pos] = g; }
            big_pos[g] = pos;
        }
    }
}

__device__ inline void sortkey32_origin(uint32_t sk, const SortKeyMap& km, double res, int64_t* key, double org[3]) {
    const int64_t k = sortkey_to_packed((uint64_t)sk, km);
    int32_t c[3];
    ndt_unpack(k, c[0], c[1], c[2]);
    *key = k;
#pragma unroll
    for (int j = 0; j < 3; ++j) org[j] = (double)c[j] * res;
}

// nine sums of every big segment: one wave per segment up to HUGE_SEG points (waves stride over the front list), one
// block per longer segment (blocks stride over the back list).  A block per segment for everything above 256 points took
// 0.37 ms on the 20 M-point map, whose voxels hold 25 points on average and a few hundred near the sensor tracks.
// (block `blk` of `nblk` blocks of BIG_THREADS threads with this role)
__device__ __forceinline__ void big_sums32_body(uint32_t blk, uint32_t nblk, const uint32_t* big_list, uint32_t list_cap, const uint32_t* big_count,
                                                const int32_t* plan, const uint4* elems, const uint32_t* seg_pos, const uint32_t* seg_end,
                                                const SortKeyMap& km, const MapBuildCfg& cfg, double* sums) {
    __shared__ double sm[BIG_THREADS / 64][9];
    const uint32_t n_big = *big_count, n_huge = (uint32_t)plan[PL_HUGECOUNT];
    const int lane = threadIdx.x & 63;
    const uint32_t n_waves = nblk * (BIG_THREADS / 64);
    for (uint32_t b = blk * (BIG_THREADS / 64) + (threadIdx.x >> 6); b < n_big; b += n_waves) {      // (uniform over the wave)
        const uint32_t g = big_list[b];
        const uint32_t s = seg_pos[g], e = seg_end[g];
        int64_t key;
        double o[3];
        sortkey32_origin(elems[s].x, km, cfg.kp.res_d, &key, o);
        double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t i0 = s + lane; i0 < e; i0 += 4 * 64) {       // four elements in flight per lane, added in index order
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + u * 64;
                v[u] = i < e ? elems[i] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + u * 64 >= e) break;
                const double d0 = (double)__uint_as_float(v[u].y) - o[0], d1 = (double)__uint_as_float(v[u].z) - o[1], d2 = (double)__uint_as_float(v[u].w) - o[2];
                a[0] += d0; a[1] += d1; a[2] += d2;
                a[3] += d0 * d0; a[4] += d0 * d1; a[5] += d0 * d2;
                a[6] += d1 * d1; a[7] += d1 * d2; a[8] += d2 * d2;
            }
        }
#pragma unroll
        for (int o2 = 32; o2 >= 1; o2 >>= 1)
#pragma unroll
            for (int j = 0; j < 9; ++j) a[j] += __shfl_xor(a[j], o2);
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < 9; ++j) sums[(size_t)g * 9 + j] = a[j];
        }
    }
    for (uint32_t b = blk; b < n_huge; b += nblk) {       // (uniform over the block)
        const uint32_t g = big_list[list_cap - 1u - b];
        const uint32_t s = seg_pos[g], e = seg_end[g];
        int64_t key;
        double o[3];
        sortkey32_origin(elems[s].x, km, cfg.kp.res_d, &key, o);
        double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t i0 = s + threadIdx.x; i0 < e; i0 += 4 * BIG_THREADS) {       // four elements in flight per lane, added in index order
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + u * BIG_THREADS;
                v[u] = i < e ? elems[i] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + u * BIG_THREADS >= e) break;
                const double d0 = (double)__uint_as_float(v[u].y) - o[0], d1 = (double)__uint_as_float(v[u].z) - o[1], d2 = (double)__uint_as_float(v[u].w) - o[2];
                a[0] += d0; a[1] += d1; a[2] += d2;
                a[3] += d0 * d0; a[4] += d0 * d1; a[5] += d0 * d2;
                a[6] += d1 * d1; a[7] += d1 * d2; a[8] += d2 * d2;
            }
        }
#pragma unroll
        for (int o2 = 32; o2 >= 1; o2 >>= 1)
#pragma unroll
            for (int j = 0; j < 9; ++j) a[j] += __shfl_xor(a[j], o2);
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < 9; ++j) sm[threadIdx.x >> 6][j] = a[j];
        }
        __syncthreads();
        if (threadIdx.x < 9) {
            double r = sm[0][threadIdx.x];
            for (int w = 1; w < BIG_THREADS / 64; ++w) r += sm[w][threadIdx.x];
            sums[(size_t)g * 9 + threadIdx.x] = r;          // straight into the segment's row
        }
        __syncthreads();
    }
}

// One group of L lanes (L = 8 or 64 by the batch's mean points per voxel) per segment, groups striding over the segments:
// strided sums in voxel-local float64 coordinates over the segment's contiguous elements, xor tree over the group — the
// same per-lane order and tree as the 64-bit-key kernel above, so both paths give the same bits.  Row g of `sums`.
// (block `blk` of `nblk` blocks of BIG_THREADS threads with this role)
__device__ __forceinline__ void segment_sums32_body(uint32_t blk, uint32_t nblk, const uint4* __restrict__ elems, uint32_t n_valid, const uint32_t* seg_pos,
                                                    const uint32_t* seg_end, const int32_t* plan, const SortKeyMap& km, const MapBuildCfg& cfg,
                                                    const uint32_t* big_pos, double* sums) {
    const uint32_t nseg = (uint32_t)plan[PL_NSEG];
    const int L = seg_wide(nseg, n_valid) ? 64 : 8;
    const int lane = threadIdx.x & 63, l = lane & (L - 1), gpw = 64 / L;      // groups per wave
    const uint32_t wpb = blockDim.x >> 6;
    const uint32_t groups = nblk * wpb * (uint32_t)gpw;
    for (uint32_t g0 = (blk * wpb + (threadIdx.x >> 6)) * (uint32_t)gpw; g0 < nseg; g0 += groups) {
        const uint32_t g = g0 + (uint32_t)(lane / L);
        const bool act = g < nseg;
        uint32_t s = 0, e = 0;
        double org[3] = {0, 0, 0};
        if (act) {
            s = seg_pos[g];
            e = big_pos[g] == SEG_NONE ? seg_end[g] : s;          // a big segment is summed by k_big_sums32
            int64_t key;
            sortkey32_origin(elems[s].x, km, cfg.kp.res_d, &key, org);
        }
        double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t i0 = s + l; i0 < e; i0 += 4 * L) {          // four elements in flight per lane, added in index order
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + u * L;
                v[u] = i < e ? elems[i] : make_uint4(0, 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (i0 + u * L >= e) break;
                const double d0 = (double)__uint_as_float(v[u].y) - org[0], d1 = (double)__uint_as_float(v[u].z) - org[1], d2 = (double)__uint_as_float(v[u].w) - org[2];
                a[0] += d0; a[1] += d1; a[2] += d2;
                a[3] += d0 * d0; a[4] += d0 * d1; a[5] += d0 * d2;
                a[6] += d1 * d1; a[7] += d1 * d2; a[8] += d2 * d2;
            }
        }
        for (int o = L >> 1; o >= 1; o >>= 1)
#pragma unroll
            for (int j = 0; j < 9; ++j) a[j] += __shfl_xor(a[j], o);
        if (act && l == 0 && e != s) {
#pragma unroll
            for (int j = 0; j < 9; ++j) sums[(size_t)g * 9 + j] = a[j];
        }
    }
}

// large batches: a launch each (256-thread blocks suit the per-segment sums better once the launch gap no longer matters:
// 20 M points 164 us as two launches, 193 us as one)
__global__ __launch_bounds__(BIG_THREADS) void k_big_sums32(const uint32_t* big_list, uint32_t list_cap, const uint32_t* big_count, const int32_t* plan,
                                                            const uint4* elems, const uint32_t* seg_pos, const uint32_t* seg_end, SortKeyMap km, MapBuildCfg cfg,
                                                            double* sums) {
    big_sums32_body(blockIdx.x, gridDim.x, big_list, list_cap, big_count, plan, elems, seg_pos, seg_end, km, cfg, sums);
}
__global__ __launch_bounds__(256) void k_segment_sums32(const uint4* __restrict__ elems, uint32_t n_valid, const uint32_t* seg_pos, const uint32_t* seg_end,
                                                        const int32_t* plan, SortKeyMap km, MapBuildCfg cfg, const uint32_t* big_pos, double* sums) {
    segment_sums32_body(blockIdx.x, gridDim.x, elems, n_valid, seg_pos, seg_end, plan, km, cfg, big_pos, sums);
}
// Both in ONE launch (they do not depend on each other: the big segments' blocks start first and run beside the others;
// as two launches the second waited for the first's slowest block: 17 + 35 us at 2 M points, 12 + 10 us per keyframe).
__global__ __launch_bounds__(BIG_THREADS) void k_sums32(uint32_t nb_big, const uint32_t* big_list, uint32_t list_cap, const uint32_t* big_count,
                                                        const int32_t* plan, const uint4* __restrict__ elems, uint32_t n_valid, const uint32_t* seg_pos,
                                                        const uint32_t* seg_end, SortKeyMap km, MapBuildCfg cfg, const uint32_t* big_pos, double* sums) {
    // the two roles alternate over the block index while both last, so that neither waits for the other's blocks to drain
    const uint32_t nb_seg = gridDim.x - nb_big, m = nb_big < nb_seg ? nb_big : nb_seg, b = blockIdx.x;
    const bool big_role = b < 2 * m ? (b & 1u) == 0u : nb_big > nb_seg;
    const uint32_t idx = b < 2 * m ? b >> 1 : b - m;
    if (big_role) big_sums32_body(idx, nb_big, big_list, list_cap, big_count, plan, elems, seg_pos, seg_end, km, cfg, sums);
    else segment_sums32_body(idx, nb_seg, elems, n_valid, seg_pos, seg_end, plan, km, cfg, big_pos, sums);
}

// ComputeMeanAndCov of the batch's points of one voxel from their nine sums, find-or-insert in the table,
// UpdateMeanAndCov with what the voxel held.  Returns the slot; *fresh_entry / *fresh_voxel tell what was new.
__device__ __forceinline__ uint32_t merge_voxel(const double a[9], int32_t nb, int64_t key, int64_t* keys, int32_t* counts, double* mean, double* cov,
                                                uint64_t mask, int32_t max_pts, bool* fresh_entry, bool* fresh_voxel, double mu[3], double cv[6], int32_t* n_total) {
    mu[0] = a[0] / nb; mu[1] = a[1] / nb; mu[2] = a[2] / nb;
#pragma unroll
    for (int j = 0; j < 6; ++j) cv[j] = 0.0;
    if (nb > 1) {   // ComputeMeanAndCov (math_utils.h:55-72): sample covariance over (n-1)
        const double dn = (double)(nb - 1);
        cv[0] = (a[3] - a[0] * mu[0]) / dn; cv[1] = (a[4] - a[0] * mu[1]) / dn; cv[2] = (a[5] - a[0] * mu[2]) / dn;
        cv[3] = (a[6] - a[1] * mu[1]) / dn; cv[4] = (a[7] - a[1] * mu[2]) / dn; cv[5] = (a[8] - a[2] * mu[2]) / dn;
    }
    uint64_t slot = ndt_hash_key(key) & mask;
    bool fresh = false;
    for (;;) {
        const unsigned long long prev = atomicCAS((unsigned long long*)&keys[slot], (unsigned long long)NDT_KEY_EMPTY, (unsigned long long)key);
        if ((int64_t)prev == NDT_KEY_EMPTY) { fresh = true; break; }
        if ((int64_t)prev == key) break;
        slot = (slot + 1) & mask;
    }
    double* mp = mean + slot * 3;
    double* cp = cov + slot * 6;
    int32_t ntot = nb;
    const int32_t had = fresh ? 0 : counts[slot];            // 0 also for a ghost cell that now gets its first points
    *fresh_entry = fresh;
    *fresh_voxel = had == 0;
    if (fresh && slot < 4) keys[mask + 1 + slot] = key;      // mirror entries
    if (had != 0 && max_pts > 0 && had > max_pts) {      // frozen (IncNdt3d): the voxel keeps what it has
#pragma unroll
        for (int j = 0; j < 3; ++j) mu[j] = mp[j];
#pragma unroll
        for (int j = 0; j < 6; ++j) cv[j] = cp[j];
        ntot = had;
    } else if (had != 0) {        // UpdateMeanAndCov (math_utils.h:87-98)
        const double hm = (double)had, cn = (double)nb, tot = hm + cn;
        double nm[3], dh[3], dc[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            nm[j] = (hm * mp[j] + cn * mu[j]) / tot;
            dh[j] = mp[j] - nm[j];
            dc[j] = mu[j] - nm[j];
        }
        int t = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = r; c < 3; ++c, ++t) cv[t] = (hm * (cp[t] + dh[r] * dh[c]) + cn * (cv[t] + dc[r] * dc[c])) / tot;
#pragma unroll
        for (int j = 0; j < 3; ++j) mu[j] = nm[j];
        ntot = had + nb;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) mp[j] = mu[j];
#pragma unroll
    for (int j = 0; j < 6; ++j) cp[j] = cv[j];
    counts[slot] = ntot;
    *n_total = ntot;
    return (uint32_t)slot;
}

// One voxel of the batch per thread: merge its sums into the table, then the regularised inverse + float32 payload.
// Voxels that got their first points go on the fresh list (the ghost stage works on those only).  Decides for the whole
// batch whether the table has room (n_entries: the host's exact count) and leaves the verdict in plan[PL_ABORT1].
__global__ __launch_bounds__(256) void k_merge_finalize32(const uint4* elems, const uint32_t* seg_pos, const uint32_t* seg_end, int32_t* plan, SortKeyMap km,
                                                          MapBuildCfg cfg, const double* sums, VoxSlot* slots, int64_t* keys, int32_t* counts, double* mean,
                                                          double* cov, uint64_t cap, uint64_t n_entries, uint32_t* fresh_list, uint32_t* slot_of) {
    if (plan[PL_SORT_ERR]) return;             // a radix pass gave up: the data is not sorted, nothing may be inserted (the host sorts again)
    const uint32_t nseg = (uint32_t)plan[PL_NSEG];
    const bool abort1 = (n_entries + nseg) * 2 > cap;
    if (blockIdx.x == 0 && threadIdx.x == 0) plan[PL_ABORT1] = abort1 ? 1 : 0;
    if (abort1) return;
    const int lane = threadIdx.x & 63;
    uint32_t n_new_ent = 0;
    for (uint32_t g0 = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); g0 < nseg; g0 += gridDim.x * blockDim.x) {     // (uniform over the wave)
        const uint32_t g = g0 + (uint32_t)lane;
        bool fv = false;
        if (g < nseg) {
            const uint32_t s = seg_pos[g], e = seg_end[g];
            int64_t key;
            double org[3], a[9], mu[3], cv[6];
            sortkey32_origin(elems[s].x, km, cfg.kp.res_d, &key, org);
#pragma unroll
            for (int j = 0; j < 9; ++j) a[j] = sums[(size_t)g * 9 + j];
            bool fe;
            int32_t ntot;
            const uint32_t slot = merge_voxel(a, (int32_t)(e - s), key, keys, counts, mean, cov, cap - 1, cfg.max_pts, &fe, &fv, mu, cv, &ntot);
            finalize_slot(&slots[slot], ntot, mu, cv, cfg);
            slot_of[g] = slot;                                   // for the ghost stage (the sums kernels are done with this array)
            n_new_ent += fe ? 1u : 0u;
        }
        const unsigned long long b = __ballot(fv);               // one returning atomic per wave
        uint32_t base = 0;
        if (lane == 0 && b) base = atomicAdd((uint32_t*)&plan[PL_NEW_VOX], (uint32_t)__popcll(b));
        base = __shfl(base, 0);
        if (fv) fresh_list[base + (uint32_t)__popcll(b & ((1ull << lane) - 1ull))] = g;
    }
    block_count_add(n_new_ent, &plan[PL_NEW_ENT]);
}

// ------------------------------------------------------------------ ghost cells and links of a batch (all paths)
// where the keys of a batch's voxels come from, and which of them got their first points in this batch (only those can
// need new ghost cells or links: every voxel's six neighbours are in the table from the moment it gets its first points)
struct SegKeys {
    const uint4* elems; const uint32_t* seg_pos;           // fast path: sorted 16-byte elements + segment positions
    const uint64_t* sk64; const uint32_t* seg_start;       // 64-bit-key path: sorted compact keys + segment starts
    const int64_t* plain;                                   // a plain list of packed keys (map_restore)
    SortKeyMap km;
    const int32_t* d_count;                                 // device word: number of items
    const uint32_t* list;                                   // item i is voxel list[i] of the batch (nullptr: voxel i)
    const uint32_t* fresh_flag;                             // nullptr, or per voxel: 0 = skip (it had points before)
    const uint32_t* slot_of;                                // nullptr, or per voxel: its slot in the table (saves the look-up)
};
__device__ inline bool seg_item(const SegKeys& K, uint32_t i, int64_t* key, uint32_t* slot) {
    const uint32_t g = K.list ? K.list[i] : i;
    *slot = K.slot_of ? K.slot_of[g] : NDT_LINK_NONE;
    if (K.fresh_flag && !K.fresh_flag[g]) return false;
    if (K.plain) *key = K.plain[g];
    else if (K.elems) *key = sortkey_to_packed((uint64_t)K.elems[K.seg_pos[g]].x, K.km);
    else *key = sortkey_to_packed(K.sk64[K.seg_start[g]], K.km);
    return true;
}

// one thread per (new voxel of the batch, face direction): find-or-insert the neighbour cell and link the two entries
// both ways.  An entry that is created here (a ghost) can only touch voxels that are new in this batch, whose own threads
// link it: no table-wide pass.  Links between two ghost cells are never read (a ghost contributes nothing to a residual)
// and stay unset.  Goes ahead whenever the table cannot overflow even if every neighbour were new (at most six ghosts per
// new voxel, kept under 7/8 of the capacity); the host brings the load factor back under 1/2 afterwards.  Verdict in
// plan[PL_ABORT2].
__global__ __launch_bounds__(256) void k_ghost_link(SegKeys K, const int32_t* d_new_vox, const int32_t* d_new_ent, int64_t* keys, VoxSlot* slots, uint64_t cap,
                                                    uint64_t n_entries, int32_t* plan) {
    if (plan[PL_ABORT1] || plan[PL_SORT_ERR]) return;
    const bool abort2 = (n_entries + (uint64_t)*d_new_ent + 6ull * (uint64_t)*d_new_vox) * 8 > cap * 7;
    if (blockIdx.x == 0 && threadIdx.x == 0) plan[PL_ABORT2] = abort2 ? 1 : 0;
    if (abort2) return;
    const uint64_t mask = cap - 1;
    const uint32_t n_items = (uint32_t)*K.d_count;
    uint32_t inserted = 0;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < (uint64_t)n_items * 8; t += (uint64_t)gridDim.x * blockDim.x) {
        const int d = (int)(t & 7);
        int64_t key, nk;
        uint32_t own;
        if (d >= 6 || !seg_item(K, (uint32_t)(t >> 3), &key, &own)) continue;
        if (own == NDT_LINK_NONE) own = table_find(keys, mask, key);
        if (own == NDT_LINK_NONE || !neighbour_key(key, d, &nk)) continue;
        uint64_t slot = ndt_hash_key(nk) & mask;
        for (;;) {
            // keys are never removed: a plain read that shows a key is final; only an empty slot needs the atomic
            int64_t prev = __hip_atomic_load(&keys[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == NDT_KEY_EMPTY)
                prev = (int64_t)atomicCAS((unsigned long long*)&keys[slot], (unsigned long long)NDT_KEY_EMPTY, (unsigned long long)nk);
            if (prev == NDT_KEY_EMPTY) { ++inserted; if (slot < 4) keys[mask + 1 + slot] = nk; break; }   // the slot is zeroed: a ghost
            if (prev == nk) break;
            slot = (slot + 1) & mask;
        }
        slots[own].link[d] = (uint32_t)slot;
        slots[slot].link[d ^ 1] = own;
    }
    block_count_add(inserted, &plan[PL_NEW_GHOST]);
}
__global__ void k_set_word(int32_t* dst, int32_t v) { *dst = v; }

#define STAGE_BLOCKS 2048      // grid of the segment-striding kernels

// The host's waits of a build / merge: kernels leave what the host needs in the pinned block and write a sequence word
// after it (system-scope release); the host polls that word instead of waiting for the stream's completion signal, which
// arrives a few microseconds later (the poll is bounded by the stream's own state).
__global__ __launch_bounds__(64) void k_publish_plan(const int32_t* misc, int32_t* mirror, int32_t seq) {
    if (threadIdx.x < PLAN_SEQ_WORD) mirror[threadIdx.x] = misc[threadIdx.x];
    __threadfence_system();
    if (threadIdx.x == 0) __hip_atomic_store(mirror + PLAN_SEQ_WORD, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static hipError_t wait_plan(MapWork& w, hipStream_t s) {
    static const bool poll = getenv("NDT_NO_POLL") == nullptr;
    if (w.h_plan_dev && poll) {
        volatile int32_t* flag = w.h_plan + PLAN_SEQ_WORD;
        for (unsigned k = 1;; ++k) {
            if (*flag == w.plan_seq) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return hipSuccess; }
            if ((k & 255u) == 0 && hipStreamQuery(s) != hipErrorNotReady) break;
            __builtin_ia32_pause();
        }
    }
    return hipStreamSynchronize(s);
}
// the pinned 256-byte block the host reads results from, and its device-side address
static hipError_t ensure_plan_host(MapWork& w) {
    if (w.h_plan) return hipSuccess;
    hipError_t e = hipHostMalloc((void**)&w.h_plan, 256, hipHostMallocDefault);
    if (e != hipSuccess) return e;
    // hipHostMalloc does not clear: a page recycled from a destroyed handle could hold an old sequence word equal to the one
    // the host is about to wait for (plan_seq restarts at 1 with every handle)
    memset(w.h_plan, 0, 256);
    if (hipHostGetDevicePointer((void**)&w.h_plan_dev, w.h_plan, 0) != hipSuccess) { (void)hipGetLastError(); w.h_plan_dev = nullptr; }
    return hipSuccess;
}
struct PlanWords { int32_t nseg, sort_err, abort1, abort2, new_vox, new_ent, new_ghost; };
// the one read-back of a call: the whole 256-byte plan block in one copy into pinned memory
static int read_plan(MapWork& w, int new_vox_word, int new_ent_word, PlanWords* p, hipStream_t s, std::string& err) {
    HIP_TRY(ensure_plan_host(w));
    if (w.h_plan_dev) k_publish_plan<<<1, 64, 0, s>>>(w.misc.as<int32_t>(), w.h_plan_dev, ++w.plan_seq);
    else HIP_TRY(hipMemcpyAsync(w.h_plan, w.misc.p, 256, hipMemcpyDeviceToHost, s));
    HIP_TRY(wait_plan(w, s));
    const int32_t* a = w.h_plan;
    *p = PlanWords{a[PL_NSEG], a[PL_SORT_ERR], a[PL_ABORT1], a[PL_ABORT2], a[new_vox_word], a[new_ent_word], a[PL_NEW_GHOST]};
    return NDT_OK;
}

// Ghost cells + links for the batch whose voxels are already merged (their counts in plan words new_vox_word / new_ent_word),
// then the one read-back of the call and the host's counters.  Grows the table first when the device found that the ghost
// cells might overflow it, and afterwards when the load factor went over 1/2.
static int ghost_stage(VoxelMap& m, MapWork& w, const SegKeys& K, uint32_t items_ub, int new_vox_word, int new_ent_word, PlanWords* pw, hipStream_t s,
                       std::string& err) {
    int32_t* plan = w.misc.as<int32_t>();
    const unsigned nb = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(STAGE_BLOCKS, ((uint64_t)items_ub * 8 + 255) / 256));
    k_ghost_link<<<nb, 256, 0, s>>>(K, plan + new_vox_word, plan + new_ent_word, m.keys, m.slots, (uint64_t)m.cap, (uint64_t)m.n_entries, plan);
    HIP_TRY(hipGetLastError());
    int rc = read_plan(w, new_vox_word, new_ent_word, pw, s, err);
    if (rc != NDT_OK || pw->abort1 || pw->sort_err) return rc;
    m.n_vox += pw->new_vox;
    m.n_entries += pw->new_ent;
    if (pw->abort2) {      // the voxels are in, their ghosts were not attempted: make room (all links are rebuilt), then link
        rc = map_reserve(m, (size_t)m.n_entries + 6 * (size_t)pw->new_vox, s, err);
        if (rc != NDT_OK) return rc;
        SegKeys K2 = K;
        K2.slot_of = nullptr;              // the table was rebuilt: the recorded slots are stale, look the voxels up
        k_ghost_link<<<nb, 256, 0, s>>>(K2, plan + new_vox_word, plan + new_ent_word, m.keys, m.slots, (uint64_t)m.cap, (uint64_t)(m.n_entries - pw->new_ent), plan);
        HIP_TRY(hipGetLastError());
        rc = read_plan(w, new_vox_word, new_ent_word, pw, s, err);
        if (rc != NDT_OK) return rc;
        if (pw->abort2) { err = "voxel table could not be sized for the batch's ghost cells"; return NDT_ERR_HIP; }
    }
    m.n_entries += pw->new_ghost;
    return map_reserve(m, (size_t)m.n_entries, s, err);          // load factor back under 1/2 if the ghosts pushed it over (rare)
}

// Stage 1 of every cloud ingest: staging (host records are copied; device records and SoA arrays are read in place),
// bounding box of the cells, sort-key layout.  The one read-back every ingest needs: the radix pass count depends on it.
struct IngestBox { const unsigned char* d_in; RtArg rt; size_t n_valid; int32_t misc[8]; SortKeyMap km; int total_bits; uint64_t bbox_cells; };
static int ingest_bbox(MapWork& w, const KeyParams& kp, const void* pts, const float* soa_y, const float* soa_z, size_t n, size_t stride, const double* R,
                       const double* t, bool zyx_order, bool write_keys, hipStream_t s, IngestBox* out, std::string& err) {
    const unsigned char* d_in = (const unsigned char*)pts;
    if (stride) {
        // records already in device memory are read in place (the key kernels are their only readers and have run when
        // the call returns, so the caller's buffer is free again on return either way)
        hipPointerAttribute_t at;
        int dev = -1;
        (void)hipGetDevice(&dev);
        const bool on_device = hipPointerGetAttributes(&at, pts) == hipSuccess && at.type == hipMemoryTypeDevice && at.device == dev;
        if (!on_device) {
            (void)hipGetLastError();                 // a plain host pointer makes the query fail: not an error
            HIP_TRY(w.raw.ensure(n * stride));
            HIP_TRY(hipMemcpyAsync(w.raw.p, pts, n * stride, hipMemcpyDefault, s));
            d_in = (const unsigned char*)w.raw.p;
        }
    }
    HIP_TRY(w.misc.ensure(256));
    RtArg rt{};
    if (R) {
        for (int i = 0; i < 9; ++i) rt.v[i] = R[i];
        for (int i = 0; i < 3; ++i) rt.v[9 + i] = t[i];
    }
    HIP_TRY(ensure_plan_host(w));
    // the fast path's digit totals, tickets and tile status words are cleared by the two kernels below
    const size_t tile_elems = (size_t)RS_THREADS * os_items_for(n);
    const size_t ntiles = (n + tile_elems - 1) / tile_elems;
    const size_t n_status = write_keys ? 0 : (size_t)OS_MAX_PASSES * ntiles * 256;
    const int n_hist = write_keys ? 0 : (GH_COPIES * OS_MAX_PASSES + 1) * 256;
    if (!write_keys) {
        HIP_TRY(w.hist.ensure((size_t)n_hist * 4));
        HIP_TRY(w.vals_a.ensure(n_status * 8));
    }
    const unsigned nb256 = (unsigned)((n + 255) / 256);
    const unsigned nb_tk = std::min<unsigned>(nb256, TK_MAX_BLOCKS);
    HIP_TRY(w.block_rows.ensure((size_t)TK_MAX_BLOCKS * 8 * 4));
    if (write_keys) {
        HIP_TRY(w.tx.ensure(n * 4)); HIP_TRY(w.ty.ensure(n * 4)); HIP_TRY(w.tz.ensure(n * 4));
        HIP_TRY(w.keys_a.ensure(n * 8)); HIP_TRY(w.vals_a.ensure(n * 4));
        k_target_keys<true><<<nb_tk, 256, 0, s>>>(d_in, soa_y, soa_z, n, stride, R ? 1 : 0, rt, kp, w.tx.as<float>(), w.ty.as<float>(), w.tz.as<float>(),
                                                  w.keys_a.as<int64_t>(), w.vals_a.as<uint32_t>(), w.block_rows.as<int32_t>(), nullptr, 0);
    } else {
        k_target_keys<false><<<nb_tk, 256, 0, s>>>(d_in, soa_y, soa_z, n, stride, R ? 1 : 0, rt, kp, nullptr, nullptr, nullptr, nullptr, nullptr,
                                                   w.block_rows.as<int32_t>(), w.vals_a.as<unsigned long long>(), n_status);
    }
    k_bbox_reduce<<<1, TK_MAX_BLOCKS, 0, s>>>(w.block_rows.as<int32_t>(), (int)nb_tk, w.misc.as<int32_t>(), n_hist ? w.hist.as<uint32_t>() : nullptr, n_hist,
                                              w.h_plan_dev, ++w.plan_seq);
    HIP_TRY(hipGetLastError());
    if (!w.h_plan_dev) HIP_TRY(hipMemcpyAsync(w.h_plan, w.misc.p, sizeof out->misc, hipMemcpyDeviceToHost, s));
    HIP_TRY(wait_plan(w, s));
    memcpy(out->misc, w.h_plan, sizeof out->misc);
    out->d_in = d_in; out->rt = rt;
    out->n_valid = (size_t)out->misc[6];
    out->total_bits = 0; out->bbox_cells = 1;
    for (int j = 0; j < 3; ++j) {
        const int ax = zyx_order ? 2 - j : j;
        out->km.axis[j] = ax;
        out->km.mn[j] = out->misc[ax];
        out->km.bits[j] = out->n_valid ? bits_for((int64_t)out->misc[3 + ax] - out->misc[ax] + 1) : 0;
        out->total_bits += out->km.bits[j];
        if (out->n_valid) out->bbox_cells *= (uint64_t)((int64_t)out->misc[3 + j] - out->misc[j] + 1);
    }
    return NDT_OK;
}

// LSD radix sort of (compact 64-bit key, point index) + ordered segments: the path for bounding boxes of more than 2^32
// cells and for the VoxelGrid downsample (whose output order is the cells' order).  The per-cell sums read the points through
// the sorted index.  pts: AoS records in host or device memory, or (stride == 0) three device SoA arrays passed through
// pts / soa_y / soa_z.
int sort_points_by_cell(MapWork& w, const KeyParams& kp, const void* pts, const float* soa_y, const float* soa_z, size_t n, size_t stride,
                        const double* R, const double* t, bool zyx_order, bool host_nseg, hipStream_t s, SortedCells* out, std::string& err,
                        const IngestBox* have_box = nullptr) {
    *out = SortedCells();
    if (n == 0) return NDT_OK;
    if (n >= (1ull << 31)) { err = "more than 2^31 points in one batch"; return NDT_ERR_INVALID_ARG; }
    IngestBox box;
    if (have_box) {         // the fast path measured the box and found the keys too wide: make the 64-bit keys now
        box = *have_box;
        const unsigned nb_tk = std::min<unsigned>((unsigned)((n + 255) / 256), TK_MAX_BLOCKS);
        HIP_TRY(w.tx.ensure(n * 4)); HIP_TRY(w.ty.ensure(n * 4)); HIP_TRY(w.tz.ensure(n * 4));
        HIP_TRY(w.keys_a.ensure(n * 8)); HIP_TRY(w.vals_a.ensure(n * 4));
        k_target_keys<true><<<nb_tk, 256, 0, s>>>(box.d_in, soa_y, soa_z, n, stride, R ? 1 : 0, box.rt, kp, w.tx.as<float>(), w.ty.as<float>(), w.tz.as<float>(),
                                                  w.keys_a.as<int64_t>(), w.vals_a.as<uint32_t>(), w.block_rows.as<int32_t>(), nullptr, 0);
        HIP_TRY(hipGetLastError());
    } else {
        int rc = ingest_bbox(w, kp, pts, soa_y, soa_z, n, stride, R, t, zyx_order, true, s, &box, err);
        if (rc != NDT_OK) return rc;
    }
    const size_t n_valid = box.n_valid;
    if (n_valid == 0) return NDT_OK;
    const SortKeyMap km = box.km;
    const int total_bits = box.total_bits;
    const unsigned nb256 = (unsigned)((n + 255) / 256);
    HIP_TRY(w.keys_b.ensure(n * 8)); HIP_TRY(w.vals_b.ensure(n * 4));
    // Points that were dropped (non-finite / out of key range) get a sort key of their own, 1 << total_bits, so that they
    // can never tie with the all-ones key of the bounding box's max corner (a 128 x 128 x 4 box has exactly 16 key bits);
    // the extra bit is sorted only when such points exist.
    const bool has_invalid = n_valid < n;
    k_sort_keys<<<nb256, 256, 0, s>>>(w.keys_a.as<int64_t>(), n, km, 1ull << total_bits);
    HIP_TRY(hipGetLastError());

    // LSD radix sort over the bits that exist
    const int P = std::max(1, (total_bits + (has_invalid ? 1 : 0) + 7) / 8);
    const bool big = n >= (1u << 20);
    const size_t rs_tile = (size_t)RS_THREADS * (big ? RS_ITEMS_BIG : RS_ITEMS_SMALL);
    const unsigned nblk = (unsigned)((n + rs_tile - 1) / rs_tile);
    const size_t nh = (size_t)256 * nblk;
    HIP_TRY(w.hist.ensure(nh * 4));
    HIP_TRY(w.scan_tmp.ensure(((std::max(nh, n) + SCAN_TILE - 1) / SCAN_TILE + 2) * 4));
    uint64_t* ka = w.keys_a.as<uint64_t>(); uint64_t* kb = w.keys_b.as<uint64_t>();
    uint32_t* va = w.vals_a.as<uint32_t>(); uint32_t* vb = w.vals_b.as<uint32_t>();
    for (int p = 0; p < P; ++p) {
        if (big) k_radix_hist<RS_ITEMS_BIG><<<nblk, RS_THREADS, 0, s>>>(ka, n, 8 * p, w.hist.as<uint32_t>(), nblk);
        else k_radix_hist<RS_ITEMS_SMALL><<<nblk, RS_THREADS, 0, s>>>(ka, n, 8 * p, w.hist.as<uint32_t>(), nblk);
        exclusive_scan_u32(w.hist.as<uint32_t>(), w.hist.as<uint32_t>(), nh, w.scan_tmp.as<uint32_t>(), nullptr, s);
        if (big) k_radix_scatter<RS_ITEMS_BIG><<<nblk, RS_THREADS, 0, s>>>(ka, va, n, 8 * p, w.hist.as<uint32_t>(), nblk, kb, vb);
        else k_radix_scatter<RS_ITEMS_SMALL><<<nblk, RS_THREADS, 0, s>>>(ka, va, n, 8 * p, w.hist.as<uint32_t>(), nblk, kb, vb);
        std::swap(ka, kb); std::swap(va, vb);
    }
    HIP_TRY(hipGetLastError());

    // segments over the valid prefix
    HIP_TRY(w.flags.ensure(n_valid * 4)); HIP_TRY(w.segid.ensure(n_valid * 4));
    const unsigned nbv = (unsigned)((n_valid + 255) / 256);
    k_heads<<<nbv, 256, 0, s>>>(ka, n_valid, w.flags.as<uint32_t>());
    uint32_t* d_nseg = (uint32_t*)(w.misc.as<int32_t>() + 7);
    exclusive_scan_u32(w.flags.as<uint32_t>(), w.segid.as<uint32_t>(), n_valid, w.scan_tmp.as<uint32_t>(), d_nseg, s);
    uint32_t nseg = 0;
    if (host_nseg) {
        HIP_TRY(hipMemcpyAsync(&nseg, d_nseg, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        HIP_TRY(w.seg_start.ensure(((size_t)nseg + 1) * 4));
    } else {
        HIP_TRY(w.seg_start.ensure((n_valid + 1) * 4));      // nseg <= n_valid: no read-back needed to size it
    }
    k_seg_starts<<<nbv, 256, 0, s>>>(w.flags.as<uint32_t>(), w.segid.as<uint32_t>(), n_valid, w.seg_start.as<uint32_t>());
    HIP_TRY(hipGetLastError());
    out->n_valid = n_valid; out->nseg = nseg; out->d_nseg = d_nseg; out->km = km; out->passes = P; out->sorted_keys = ka; out->sorted_vals = va;
    out->bbox_cells = box.bbox_cells;
    return NDT_OK;
}

// 64-bit-key path of a map merge (bounding boxes of more than 2^32 cells): ordered segments, sums through the sorted index
static int map_merge_wide(VoxelMap& m, MapWork& w, const MapBuildCfg& cfg, const void* pts, const float* soa_y, const float* soa_z, size_t n, size_t stride,
                          const double* R, const double* t, const IngestBox& box, hipStream_t s, int* passes, std::string& err) {
    SortedCells sc;
    int rc = sort_points_by_cell(w, cfg.kp, pts, soa_y, soa_z, n, stride, R, t, false, true, s, &sc, err, &box);
    if (rc != NDT_OK) return rc;
    if (passes) *passes = sc.passes;
    if (sc.nseg == 0) return map_reserve(m, 1, s, err);
    const uint32_t nseg = sc.nseg;
    rc = map_reserve(m, (size_t)m.n_entries + nseg, s, err);
    if (rc != NDT_OK) return rc;
    HIP_TRY(w.flags2.ensure((size_t)nseg * 4 + 16));
    const uint32_t* segs = w.seg_start.as<uint32_t>();
    uint32_t* seg_slot = w.segid.as<uint32_t>();        // free again once k_seg_starts has run; nseg <= n_valid entries
    const bool wide = sc.n_valid / nseg >= 24;
    const uint32_t big = wide ? 2048u : 512u;           // at most 32 / 64 trips of a group's loop
    const size_t big_ub = sc.n_valid / big;              // no more segments than this can be longer than `big`
    HIP_TRY(w.big_pos.ensure((size_t)nseg * 4)); HIP_TRY(w.big_list.ensure(big_ub * 4 + 16)); HIP_TRY(w.big_sums.ensure(big_ub * 9 * 8 + 16));
    uint32_t* big_count = (uint32_t*)(w.misc.as<int32_t>() + 12);
    k_big_list<<<(nseg + 255) / 256, 256, 0, s>>>(segs, nseg, big, w.big_pos.as<uint32_t>(), w.big_list.as<uint32_t>(), big_count);
    if (big_ub)
        k_big_sums<<<(unsigned)big_ub, BIG_THREADS, 0, s>>>(w.big_list.as<uint32_t>(), big_count, sc.sorted_keys, segs, sc.sorted_vals, w.tx.as<float>(), w.ty.as<float>(),
                                                           w.tz.as<float>(), sc.km, cfg, w.big_sums.as<double>());
    if (wide) {
        const unsigned nbs = (unsigned)(((size_t)nseg * 64 + 255) / 256);
        k_segment_merge<64><<<nbs, 256, 0, s>>>(sc.sorted_keys, segs, nseg, sc.sorted_vals, w.tx.as<float>(), w.ty.as<float>(), w.tz.as<float>(), sc.km, cfg, m.keys,
                                                m.counts, m.mean, m.cov, (uint64_t)(m.cap - 1), w.flags.as<uint32_t>(), w.flags2.as<uint32_t>(), seg_slot, w.big_pos.as<uint32_t>(), w.big_sums.as<double>());
    } else {
        const unsigned nbs = (unsigned)(((size_t)nseg * 8 + 255) / 256);
        k_segment_merge<8><<<nbs, 256, 0, s>>>(sc.sorted_keys, segs, nseg, sc.sorted_vals, w.tx.as<float>(), w.ty.as<float>(), w.tz.as<float>(), sc.km, cfg, m.keys,
                                               m.counts, m.mean, m.cov, (uint64_t)(m.cap - 1), w.flags.as<uint32_t>(), w.flags2.as<uint32_t>(), seg_slot, w.big_pos.as<uint32_t>(), w.big_sums.as<double>());
    }
    k_finalize_segments<<<(nseg + 255) / 256, 256, 0, s>>>(seg_slot, nseg, cfg, m.slots, m.counts, m.mean, m.cov);
    // sums of the per-voxel flags (new voxels, new table entries) stay on the device: misc[8], misc[9]
    int32_t* plan = w.misc.as<int32_t>();
    const size_t nbf = ((size_t)nseg + SCAN_TILE - 1) / SCAN_TILE;
    HIP_TRY(w.scan_tmp.ensure((nbf + 2) * 4));
    k_scan_reduce<<<(unsigned)nbf, SCAN_THREADS, 0, s>>>(w.flags.as<uint32_t>(), nseg, w.scan_tmp.as<uint32_t>());
    k_scan_spine<<<1, SCAN_THREADS, 0, s>>>(w.scan_tmp.as<uint32_t>(), nbf, (uint32_t*)(plan + 8));
    k_scan_reduce<<<(unsigned)nbf, SCAN_THREADS, 0, s>>>(w.flags2.as<uint32_t>(), nseg, w.scan_tmp.as<uint32_t>());
    k_scan_spine<<<1, SCAN_THREADS, 0, s>>>(w.scan_tmp.as<uint32_t>(), nbf, (uint32_t*)(plan + 9));
    HIP_TRY(hipGetLastError());
    SegKeys K{};
    K.sk64 = sc.sorted_keys; K.seg_start = segs; K.km = sc.km; K.d_count = plan + PL_NSEG; K.fresh_flag = w.flags.as<uint32_t>();
    PlanWords pw;
    return ghost_stage(m, w, K, nseg, 8, 9, &pw, s, err);      // the table was reserved for the voxels above: abort1 stays 0
}

// Map build / keyframe merge.  One read-back for the bounding box, one at the end for everything the device decided.
// Map build / keyframe merge.  One read-back for the bounding box, one at the end for everything the device decided.
int map_merge_points(VoxelMap& m, MapWork& w, const MapBuildCfg& cfg, const void* pts, const float* soa_y, const float* soa_z, size_t n,
                     size_t stride, const double* R, const double* t, hipStream_t s, int* passes, std::string& err) {
    if (passes) *passes = 0;
    w.last_n = 0;
    if (n == 0) return map_reserve(m, 1, s, err);
    if (n >= (1ull << 31)) { err = "more than 2^31 points in one batch"; return NDT_ERR_INVALID_ARG; }
    IngestBox box;
    int rc = ingest_bbox(w, cfg.kp, pts, soa_y, soa_z, n, stride, R, t, false, false, s, &box, err);
    if (rc != NDT_OK) return rc;
    w.last_in = box.d_in; w.last_y = soa_y; w.last_z = soa_z; w.last_n = n; w.last_stride = stride; w.last_has_T = R ? 1 : 0;
    for (int i = 0; i < 12; ++i) w.last_rt[i] = box.rt.v[i];
    if (box.n_valid == 0) return map_reserve(m, 1, s, err);
    const bool has_invalid = box.n_valid < n;
    if (box.total_bits + (has_invalid ? 1 : 0) > 32) return map_merge_wide(m, w, cfg, pts, soa_y, soa_z, n, stride, R, t, box, s, passes, err);

    // ---- 16-byte elements {compact key, x, y, z} + the digit totals of every pass, then one kernel per radix pass
    const size_t n_valid = box.n_valid;
    const int P = std::max(1, (box.total_bits + (has_invalid ? 1 : 0) + 7) / 8);      // <= 4
    if (passes) *passes = P;
    const int os_items = os_items_for(n);
    const unsigned ntiles = (unsigned)((n + (size_t)RS_THREADS * os_items - 1) / ((size_t)RS_THREADS * os_items));
    const size_t status_words = (size_t)ntiles * 256;
    HIP_TRY(w.keys_a.ensure(n * 16)); HIP_TRY(w.keys_b.ensure(n * 16));
    // hist: [P][256] digit totals, [P] tickets (padded to 256 words); vals_a: the tiles' status words of every pass
    // (hist and vals_a were sized and zeroed in ingest_bbox, before the host waited for the bounding box)
    uint4* ea = w.keys_a.as<uint4>(); uint4* eb = w.keys_b.as<uint4>();
    uint32_t* ghist = w.hist.as<uint32_t>();
    uint32_t* tickets = ghist + GH_COPIES * GH_STRIDE;
    int32_t* plan = w.misc.as<int32_t>();
    const uint32_t invalid_sk = has_invalid ? (uint32_t)(1ull << box.total_bits) : 0xFFFFFFFFu;
    const unsigned nb_me = (unsigned)std::min<size_t>(ME_MAX_BLOCKS, (n + RS_THREADS * 4 - 1) / (RS_THREADS * 4));       // one trip of four points per thread while that fits
    k_make_elems32<<<nb_me, RS_THREADS, 0, s>>>(box.d_in, soa_y, soa_z, n, stride, R ? 1 : 0, box.rt, cfg.kp, box.km, invalid_sk, P, ea, ghist);
    static const bool three_kernel_passes = getenv("NDT_NO_ONESWEEP") != nullptr;        // A/B switch
    auto sort_three_kernels = [&](uint4*& a, uint4*& b) -> int {
        const unsigned nblk = (unsigned)((n + OS_TILE - 1) / OS_TILE);
        const size_t nh = (size_t)256 * nblk;
        HIP_TRY(w.vals_b.ensure(nh * 4));
        HIP_TRY(w.scan_tmp.ensure(((nh + SCAN_TILE - 1) / SCAN_TILE + 2) * 4));
        for (int p = 0; p < P; ++p) {
            k_radix_hist32<OS_ITEMS><<<nblk, RS_THREADS, 0, s>>>(a, n, 8 * p, w.vals_b.as<uint32_t>(), nblk);
            exclusive_scan_u32(w.vals_b.as<uint32_t>(), w.vals_b.as<uint32_t>(), nh, w.scan_tmp.as<uint32_t>(), nullptr, s);
            k_radix_scatter32<OS_ITEMS><<<nblk, RS_THREADS, 0, s>>>(a, n, 8 * p, w.vals_b.as<uint32_t>(), nblk, b);
            std::swap(a, b);
        }
        HIP_TRY(hipGetLastError());
        return NDT_OK;
    };
    if (three_kernel_passes) {
        if ((rc = sort_three_kernels(ea, eb)) != NDT_OK) return rc;
    } else {
        // tiles that are all resident at once take their block index as their number (no ticket queue)
        if (!w.os_resident) {
            int per_cu = 0, dev = 0, n_cu = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_onesweep32<OS_ITEMS>, RS_THREADS, 0) == hipSuccess && hipGetDevice(&dev) == hipSuccess &&
                hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && per_cu > 0 && n_cu > 0)
                w.os_resident = (unsigned)(per_cu * n_cu);
            else { (void)hipGetLastError(); w.os_resident = 1; }
        }
        const bool numbered_by_block = ntiles <= w.os_resident;
        for (int p = 0; p < P; ++p) {
            unsigned long long* st_p = w.vals_a.as<unsigned long long>() + (size_t)p * status_words;
            uint32_t* tk = numbered_by_block ? nullptr : tickets + p;
            if (os_items == OS_ITEMS) k_onesweep32<OS_ITEMS><<<ntiles, RS_THREADS, 0, s>>>(ea, eb, (uint32_t)n, 8 * p, ghist + p * 256, st_p, tk, plan + PL_SORT_ERR);
            else k_onesweep32<OS_ITEMS_SMALL><<<ntiles, RS_THREADS, 0, s>>>(ea, eb, (uint32_t)n, 8 * p, ghist + p * 256, st_p, tk, plan + PL_SORT_ERR);
            std::swap(ea, eb);
        }
        static const bool test_sort_err = getenv("NDT_TEST_SORT_ERR") != nullptr;      // tests: pretend a pass gave up
        if (test_sort_err) k_set_word<<<1, 1, 0, s>>>(plan + PL_SORT_ERR, 1);
        HIP_TRY(hipGetLastError());
    }

    // ---- segments (unordered), ends, sums, merge + payloads, ghost cells of the new voxels
    const size_t nseg_ub = (size_t)std::min<uint64_t>(n_valid, box.bbox_cells);
    HIP_TRY(w.seg_start.ensure(nseg_ub * 4 + 16)); HIP_TRY(w.flags.ensure(nseg_ub * 4 + 16)); HIP_TRY(w.segid.ensure(nseg_ub * 4 + 16));
    HIP_TRY(w.big_pos.ensure(nseg_ub * 4 + 16)); HIP_TRY(w.big_sums.ensure(nseg_ub * 9 * 8 + 16));
    const size_t big_ub = n_valid / 256 + 1;                    // no more segments than this can be longer than the smaller threshold
    HIP_TRY(w.big_list.ensure(big_ub * 4 + 16));
    uint32_t* seg_pos = w.seg_start.as<uint32_t>(); uint32_t* seg_end = w.flags.as<uint32_t>(); uint32_t* fresh_list = w.segid.as<uint32_t>();
    double* sums = w.big_sums.as<double>();
    uint32_t* big_count = (uint32_t*)(plan + PL_BIGCOUNT);
    if (!m.cap) { rc = map_reserve(m, 1, s, err); if (rc != NDT_OK) return rc; }
    const unsigned nb_heads = (unsigned)((n_valid + SH_THREADS * SH_ITEMS - 1) / (SH_THREADS * SH_ITEMS));
    const unsigned nb_seg8 = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(STAGE_BLOCKS, (nseg_ub * 8 + 255) / 256));
    const unsigned nb_seg1 = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(STAGE_BLOCKS, (nseg_ub + 255) / 256));
    PlanWords pw;
    bool resorted = false;
    for (int attempt = 0;; ++attempt) {
        if (attempt == 0 || pw.sort_err) {
            k_seg_heads32<<<nb_heads, SH_THREADS, 0, s>>>(ea, (uint32_t)n_valid, seg_pos, plan);       // plan[PL_NSEG] starts at zero
            k_seg_ends32<<<nb_seg8, 256, 0, s>>>(ea, (uint32_t)n_valid, seg_pos, plan, seg_end, w.big_pos.as<uint32_t>(), w.big_list.as<uint32_t>(), (uint32_t)big_ub, big_count);
            if (n_valid < (4u << 20)) {
                const unsigned nb_big = (unsigned)std::min<size_t>(big_ub, 512);
                k_sums32<<<nb_big + STAGE_BLOCKS / 4, BIG_THREADS, 0, s>>>(nb_big, w.big_list.as<uint32_t>(), (uint32_t)big_ub, big_count, plan, ea, (uint32_t)n_valid,
                                                                           seg_pos, seg_end, box.km, cfg, w.big_pos.as<uint32_t>(), sums);
            } else {
                k_big_sums32<<<(unsigned)std::min<size_t>(big_ub, 1024), BIG_THREADS, 0, s>>>(w.big_list.as<uint32_t>(), (uint32_t)big_ub, big_count, plan, ea, seg_pos,
                                                                                              seg_end, box.km, cfg, sums);
                k_segment_sums32<<<STAGE_BLOCKS, 256, 0, s>>>(ea, (uint32_t)n_valid, seg_pos, seg_end, plan, box.km, cfg, w.big_pos.as<uint32_t>(), sums);
            }
        }
        k_merge_finalize32<<<nb_seg1, 256, 0, s>>>(ea, seg_pos, seg_end, plan, box.km, cfg, sums, m.slots, m.keys, m.counts, m.mean, m.cov, (uint64_t)m.cap,
                                                   (uint64_t)m.n_entries, fresh_list, w.big_pos.as<uint32_t>());
        HIP_TRY(hipGetLastError());
        SegKeys K{};
        K.elems = ea; K.seg_pos = seg_pos; K.km = box.km; K.d_count = plan + PL_NEW_VOX; K.list = fresh_list; K.slot_of = w.big_pos.as<uint32_t>();
        rc = ghost_stage(m, w, K, (uint32_t)nseg_ub, PL_NEW_VOX, PL_NEW_ENT, &pw, s, err);
        if (rc != NDT_OK) return rc;
        if (pw.sort_err) {
            // a one-sweep pass gave up waiting for a predecessor tile (its wait is bounded; never seen outside the test switch).
            // Nothing was inserted (the insert stages check the flag): make the elements again and sort them with the
            // three-kernel passes, which wait for nobody, then go on as usual.
            if (resorted) { err = "radix sort failed twice"; return NDT_ERR_HIP; }
            resorted = true;
            HIP_TRY(hipMemsetAsync((char*)w.misc.p + 160, 0, 32, s));
            HIP_TRY(hipMemsetAsync((char*)w.misc.p + 4 * PL_NSEG, 0, 28, s));       // nseg .. sort error
            ea = w.keys_a.as<uint4>(); eb = w.keys_b.as<uint4>();
            k_make_elems32<<<nb_me, RS_THREADS, 0, s>>>(box.d_in, soa_y, soa_z, n, stride, R ? 1 : 0, box.rt, cfg.kp, box.km, invalid_sk, P, ea, ghist);
            if ((rc = sort_three_kernels(ea, eb)) != NDT_OK) return rc;
            continue;
        }
        if (!pw.abort1) break;
        if (attempt >= 2) { err = "voxel table could not be sized for the batch"; return NDT_ERR_HIP; }
        // the device found that the batch's voxels might not fit (nothing was inserted): room for them and, at the usual
        // ratio, their ghost cells; then the insert stages run again on the sorted data and sums that are still in place
        rc = map_reserve(m, (size_t)m.n_entries + (size_t)pw.nseg + (size_t)pw.nseg * 3 / 2, s, err);
        if (rc != NDT_OK) return rc;
        HIP_TRY(hipMemsetAsync(plan + PL_ABORT1, 0, 24, s));               // abort flags, new_vox, new_ent, new_ghost
    }
    return NDT_OK;
}

// Scan pre-processing: pcl::transformPointCloud + pcl::VoxelGrid(leaf) of process.cc:62-73 —
// one centroid per occupied leaf cell, cells in PCL's index order (x fastest, z slowest); xyz only.
int downsample_points(MapWork& w, const KeyParams& leaf_kp, const void* pts, size_t n, size_t stride, const double* R, const double* t,
                      DevBuf& ox, DevBuf& oy, DevBuf& oz, const uint32_t** d_n_out, hipStream_t s, std::string& err) {
    *d_n_out = nullptr;
    SortedCells sc;
    int rc = sort_points_by_cell(w, leaf_kp, pts, nullptr, nullptr, n, stride, R, t, true, false, s, &sc, err);
    if (rc != NDT_OK) return rc;
    if (sc.n_valid == 0) return NDT_OK;
    HIP_TRY(ox.ensure(sc.n_valid * 4 + 16)); HIP_TRY(oy.ensure(sc.n_valid * 4 + 16)); HIP_TRY(oz.ensure(sc.n_valid * 4 + 16));
    // lanes per cell from what is known without a read-back: at least n_valid / bbox_cells points per cell on average
    const uint64_t cells_ub = std::min<uint64_t>(sc.bbox_cells, sc.n_valid);
    if (sc.n_valid / cells_ub >= 24) {
        k_segment_centroid<64><<<(unsigned)((cells_ub * 64 + 255) / 256), 256, 0, s>>>(w.seg_start.as<uint32_t>(), sc.d_nseg, sc.sorted_vals, w.tx.as<float>(),
                                                                                     w.ty.as<float>(), w.tz.as<float>(), ox.as<float>(), oy.as<float>(), oz.as<float>());
    } else {
        k_segment_centroid<8><<<(unsigned)((cells_ub * 8 + 255) / 256), 256, 0, s>>>(w.seg_start.as<uint32_t>(), sc.d_nseg, sc.sorted_vals, w.tx.as<float>(),
                                                                                   w.ty.as<float>(), w.tz.as<float>(), ox.as<float>(), oy.as<float>(), oz.as<float>());
    }
    HIP_TRY(hipGetLastError());
    *d_n_out = sc.d_nseg;
    return NDT_OK;
}

// ------------------------------------------------------------------ export (parity checks)

__global__ void k_export(const int64_t* keys, const VoxSlot* slots, const int32_t* counts, const double* mean, const double* cov, size_t cap,
                         uint32_t* counter, int64_t* okeys, int32_t* ocounts, VoxSlot* oslots, double* omean, double* ocov) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    if (keys[i] == NDT_KEY_EMPTY || counts[i] == 0) return;     // free slot or ghost cell
    uint32_t o = atomicAdd(counter, 1u);
    okeys[o] = keys[i]; ocounts[o] = counts[i];
    oslots[o] = slots[i];
    for (int j = 0; j < 3; ++j) omean[(size_t)o * 3 + j] = mean[i * 3 + j];
    for (int j = 0; j < 6; ++j) ocov[(size_t)o * 6 + j] = cov[i * 6 + j];
}

int map_export(const VoxelMap& m, const KeyParams& kp, std::vector<int64_t>& keys, std::vector<int32_t>& counts,
               std::vector<double>& mean, std::vector<double>& cov, std::vector<double>& info,
               std::vector<uint8_t>& valid, hipStream_t s, std::string& err) {
    const size_t nv = (size_t)m.n_vox;
    keys.assign(nv, 0); counts.assign(nv, 0); mean.assign(nv * 3, 0); cov.assign(nv * 9, 0); info.assign(nv * 9, 0); valid.assign(nv, 0);
    if (!nv) return NDT_OK;
    DevBuf dcnt, dsl, dm, dc, dk, dn;
    HIP_TRY(dcnt.ensure(4)); HIP_TRY(dsl.ensure(nv * sizeof(VoxSlot))); HIP_TRY(dm.ensure(nv * 24)); HIP_TRY(dc.ensure(nv * 48));
    HIP_TRY(dk.ensure(nv * 8)); HIP_TRY(dn.ensure(nv * 4));
    HIP_TRY(hipMemsetAsync(dcnt.p, 0, 4, s));
    k_export<<<(unsigned)((m.cap + 255) / 256), 256, 0, s>>>(m.keys, m.slots, m.counts, m.mean, m.cov, m.cap, dcnt.as<uint32_t>(), dk.as<int64_t>(),
                                                             dn.as<int32_t>(), dsl.as<VoxSlot>(), dm.as<double>(), dc.as<double>());
    HIP_TRY(hipGetLastError());
    std::vector<VoxSlot> hs(nv);
    std::vector<int64_t> hk(nv);
    std::vector<int32_t> hn(nv);
    std::vector<double> hm(nv * 3), hc(nv * 6);
    HIP_TRY(hipMemcpyAsync(hk.data(), dk.p, nv * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(hn.data(), dn.p, nv * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(hs.data(), dsl.p, nv * sizeof(VoxSlot), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(hm.data(), dm.p, nv * 24, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(hc.data(), dc.p, nv * 48, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    dcnt.release(); dsl.release(); dm.release(); dc.release(); dk.release(); dn.release();
    std::vector<size_t> ord(nv);
    std::iota(ord.begin(), ord.end(), 0);
    std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return hk[a] < hk[b]; });
    static const int U[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    for (size_t o = 0; o < nv; ++o) {
        const size_t i = ord[o];
        const VoxSlot& v = hs[i];
        int32_t c[3];
        ndt_unpack(hk[i], c[0], c[1], c[2]);
        keys[o] = hk[i]; counts[o] = hn[i]; valid[o] = (uint8_t)v.valid;
        const double inf6[6] = {v.i00, v.i01, v.i02, v.i11, v.i12, v.i22};
        for (int r = 0; r < 3; ++r) {
            mean[o * 3 + r] = (double)c[r] * kp.res_d + hm[i * 3 + r];
            for (int cc = 0; cc < 3; ++cc) {
                cov[o * 9 + r * 3 + cc] = hc[i * 6 + U[r][cc]];
                info[o * 9 + r * 3 + cc] = inf6[U[r][cc]];
            }
        }
    }
    return NDT_OK;
}

// ------------------------------------------------------------------ serialised maps
// A saved map is the exact per-voxel state (key, count, float64 mean offset and covariance); loading re-inserts the
// voxels and derives the float32 payload with the same finalize_slot, so a loaded map answers bit for bit like the
// map that was saved, and later merges continue from the same statistics.
__global__ void k_import(const int64_t* in_keys, const int32_t* in_counts, const double* in_mean, const double* in_cov, uint32_t n, MapBuildCfg cfg,
                         VoxSlot* slots, int64_t* keys, int32_t* counts, double* mean, double* cov, uint64_t mask, uint32_t* fresh_flag,
                         uint32_t* entry_flag, const unsigned long long* in_touch, unsigned long long stamp0, unsigned long long* touch) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n) return;
    const int64_t key = in_keys[g];
    uint64_t slot = ndt_hash_key(key) & mask;
    bool fresh = false;
    for (;;) {
        unsigned long long prev = atomicCAS((unsigned long long*)&keys[slot], (unsigned long long)NDT_KEY_EMPTY, (unsigned long long)key);
        if ((int64_t)prev == NDT_KEY_EMPTY) { fresh = true; break; }
        if ((int64_t)prev == key) break;          // cannot happen for a validated file into a cleared map
        slot = (slot + 1) & mask;
    }
    if (fresh && slot < 4) keys[mask + 1 + slot] = key;
    entry_flag[g] = fresh ? 1u : 0u;
    fresh_flag[g] = 1u;
    double mu[3], cv[6];
    for (int j = 0; j < 3; ++j) { mu[j] = in_mean[(size_t)g * 3 + j]; mean[slot * 3 + j] = mu[j]; }
    for (int j = 0; j < 6; ++j) { cv[j] = in_cov[(size_t)g * 6 + j]; cov[slot * 6 + j] = cv[j]; }
    counts[slot] = in_counts[g];
    finalize_slot(&slots[slot], in_counts[g], mu, cv, cfg);
    if (touch) touch[slot] = in_touch ? in_touch[g] : stamp0 | (unsigned long long)g;     // (a loaded file: one cloud, in file order)
}

__global__ void k_export_raw(const int64_t* keys, const int32_t* counts, const double* mean, const double* cov, size_t cap, uint32_t* counter,
                             int64_t* okeys, int32_t* ocounts, double* omean, double* ocov, const unsigned long long* touch, unsigned long long* otouch) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cap) return;
    if (keys[i] == NDT_KEY_EMPTY || counts[i] == 0) return;
    uint32_t o = atomicAdd(counter, 1u);
    okeys[o] = keys[i]; ocounts[o] = counts[i];
    for (int j = 0; j < 3; ++j) omean[(size_t)o * 3 + j] = mean[i * 3 + j];
    for (int j = 0; j < 6; ++j) ocov[(size_t)o * 6 + j] = cov[i * 6 + j];
    if (otouch) otouch[o] = touch[i];
}

// raw per-voxel state in ascending key order (host vectors)
int map_dump(const VoxelMap& m, std::vector<int64_t>& keys, std::vector<int32_t>& counts, std::vector<double>& mean, std::vector<double>& cov,
             hipStream_t s, std::string& err, std::vector<unsigned long long>* touch) {
    if (touch) touch->assign((size_t)m.n_vox, 0ull);
    const size_t nv = (size_t)m.n_vox;
    keys.assign(nv, 0); counts.assign(nv, 0); mean.assign(nv * 3, 0); cov.assign(nv * 6, 0);
    if (!nv) return NDT_OK;
    const bool with_touch = touch && m.touch;
    DevBuf dcnt, dm, dc, dk, dn, dt;
    HIP_TRY(dcnt.ensure(4)); HIP_TRY(dm.ensure(nv * 24)); HIP_TRY(dc.ensure(nv * 48)); HIP_TRY(dk.ensure(nv * 8)); HIP_TRY(dn.ensure(nv * 4));
    if (with_touch) HIP_TRY(dt.ensure(nv * 8));
    HIP_TRY(hipMemsetAsync(dcnt.p, 0, 4, s));
    k_export_raw<<<(unsigned)((m.cap + 255) / 256), 256, 0, s>>>(m.keys, m.counts, m.mean, m.cov, m.cap, dcnt.as<uint32_t>(), dk.as<int64_t>(),
                                                                 dn.as<int32_t>(), dm.as<double>(), dc.as<double>(), m.touch,
                                                                 with_touch ? dt.as<unsigned long long>() : nullptr);
    HIP_TRY(hipGetLastError());
    std::vector<int64_t> hk(nv); std::vector<int32_t> hn(nv); std::vector<double> hm(nv * 3), hc(nv * 6);
    std::vector<unsigned long long> ht(with_touch ? nv : 0);
    if (with_touch) HIP_TRY(hipMemcpyAsync(ht.data(), dt.p, nv * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(hk.data(), dk.p, nv * 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(hn.data(), dn.p, nv * 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(hm.data(), dm.p, nv * 24, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(hc.data(), dc.p, nv * 48, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    dcnt.release(); dm.release(); dc.release(); dk.release(); dn.release(); dt.release();
    std::vector<size_t> ord(nv);
    std::iota(ord.begin(), ord.end(), 0);
    std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) { return hk[a] < hk[b]; });
    for (size_t o = 0; o < nv; ++o) {
        const size_t i = ord[o];
        keys[o] = hk[i]; counts[o] = hn[i];
        for (int j = 0; j < 3; ++j) mean[o * 3 + j] = hm[i * 3 + j];
        for (int j = 0; j < 6; ++j) cov[o * 6 + j] = hc[i * 6 + j];
        if (with_touch) (*touch)[o] = ht[i];
    }
    return NDT_OK;
}

// the inverse: n voxels (distinct keys, host arrays) into a cleared map
int map_restore(VoxelMap& m, MapWork& w, const MapBuildCfg& cfg, const int64_t* keys, const int32_t* counts, const double* mean, const double* cov,
                size_t n, hipStream_t s, std::string& err, const unsigned long long* touch) {
    if (n >= (1ull << 31)) { err = "more than 2^31 voxels"; return NDT_ERR_INVALID_ARG; }
    int rc = map_reserve(m, n ? n * 5 / 2 : 1, s, err);      // voxels + their ghost cells at the usual ratio (the ghost stage grows the table if not)
    if (rc != NDT_OK || n == 0) return rc;
    HIP_TRY(w.misc.ensure(256));
    HIP_TRY(w.keys_a.ensure(n * 8)); HIP_TRY(w.vals_a.ensure(n * 4)); HIP_TRY(w.keys_b.ensure(n * 24)); HIP_TRY(w.raw.ensure(n * 48));
    HIP_TRY(w.flags.ensure(n * 4)); HIP_TRY(w.flags2.ensure(n * 4 + 16));
    HIP_TRY(hipMemcpyAsync(w.keys_a.p, keys, n * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(w.vals_a.p, counts, n * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(w.keys_b.p, mean, n * 24, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(w.raw.p, cov, n * 48, hipMemcpyHostToDevice, s));
    const unsigned long long* d_touch = nullptr;
    if (m.touch && touch) {
        HIP_TRY(w.sel.ensure(n * 8));
        HIP_TRY(hipMemcpyAsync(w.sel.p, touch, n * 8, hipMemcpyHostToDevice, s));
        d_touch = w.sel.as<unsigned long long>();
    } else if (m.touch) m.cloud_no += 1;          // no stamps came along: the restored voxels count as one cloud, in the order given
    k_import<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(w.keys_a.as<int64_t>(), w.vals_a.as<int32_t>(), w.keys_b.as<double>(), w.raw.as<double>(), (uint32_t)n,
                                                         cfg, m.slots, m.keys, m.counts, m.mean, m.cov, (uint64_t)(m.cap - 1), w.flags.as<uint32_t>(),
                                                         w.flags2.as<uint32_t>(), d_touch, (unsigned long long)m.cloud_no << 32, m.touch);
    HIP_TRY(hipGetLastError());
    // every key is new (the blob was validated: distinct keys into a cleared map): n voxels, n table entries
    int32_t* plan = w.misc.as<int32_t>();
    HIP_TRY(hipMemsetAsync((char*)w.misc.p + 160, 0, 32, s));
    HIP_TRY(hipMemsetAsync((char*)w.misc.p + 4 * PL_NSEG, 0, 28, s));       // nseg .. sort error
    k_set_word<<<1, 1, 0, s>>>(plan + PL_NSEG, (int32_t)n);
    k_set_word<<<1, 1, 0, s>>>(plan + PL_NEW_VOX, (int32_t)n);
    k_set_word<<<1, 1, 0, s>>>(plan + PL_NEW_ENT, (int32_t)n);
    HIP_TRY(hipGetLastError());
    SegKeys K{};
    K.plain = w.keys_a.as<int64_t>(); K.d_count = plan + PL_NSEG;
    PlanWords pw;
    return ghost_stage(m, w, K, (uint32_t)n, PL_NEW_VOX, PL_NEW_ENT, &pw, s, err);
}

// ------------------------------------------------------------------ voxel capacity (IncNdt3d's LRU list)
// IncNdt3d keeps its voxels in a list: a point moves its voxel to the front, a new voxel is pushed to the front and the
// tail is popped when the size reaches capacity_.  After a cloud the list is ordered by the last point that touched each
// voxel and holds the capacity - 1 most recently touched ones.  Here every slot carries that order as a stamp,
// (cloud number << 32) | index of the cloud's last point in the voxel, and the cut is made once per merged cloud:
//   k_touch_points   every point of the batch looks its voxel up and raises the voxel's stamp (integer atomic max: the
//                    result does not depend on the order of the lanes);
//   k_sel_*          radix select, most significant byte first, of the k-th smallest stamp among the voxels that hold
//                    points (k = voxels beyond capacity - 1; stamps are distinct): 8 rounds of a 256-bin count over the
//                    table and a one-thread pick — no sort, no read-back;
//   k_evict          voxels at or below that stamp become empty cells again: key and neighbour links stay (the cell is what
//                    a ghost cell is — DESIGN.md §3 — so no link anywhere has to change), points, statistics and payload go.
// Evicted cells stay in the table as ghost cells; when those outnumber the voxels eight to one the table is rebuilt from
// the voxels alone, on the device (map_compact: k_compact_move + the ghost stage; stamps carried).

__global__ __launch_bounds__(256) void k_touch_points(const unsigned char* raw, const float* soa_y, const float* soa_z, size_t n, size_t stride, int has_T, RtArg rt,
                                                      KeyParams kp, const int64_t* keys, uint64_t mask, unsigned long long* touch, unsigned long long stamp0) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float x, y, z;
        int32_t c[3];
        if (!load_target_point(raw, soa_y, soa_z, i, stride, has_T, rt.v, kp, x, y, z, c)) continue;
        const uint32_t slot = table_find(keys, mask, ndt_pack(c[0], c[1], c[2]));
        if (slot != NDT_LINK_NONE) atomicMax(&touch[slot], stamp0 | (unsigned long long)i);
    }
}
// sel[0] = the stamp's bytes fixed so far (from the top), sel[1] = rank still looked for among the stamps that share them;
// hist = sel + 2 viewed as 256 counters
__global__ void k_sel_init(unsigned long long* sel, unsigned long long k) {
    if (threadIdx.x == 0) { sel[0] = 0ull; sel[1] = k; }
    ((uint32_t*)(sel + 2))[threadIdx.x] = 0u;
}
__global__ __launch_bounds__(256) void k_sel_hist(const int32_t* counts, const unsigned long long* touch, size_t cap, const unsigned long long* sel, int shift,
                                                  uint32_t* hist) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long prefix = sel[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (size_t)gridDim.x * blockDim.x) {
        if (counts[i] <= 0) continue;
        const unsigned long long t = touch[i];
        if (shift < 56 && (t >> (shift + 8)) != (prefix >> (shift + 8))) continue;
        atomicAdd(&h[(unsigned)(t >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
__global__ void k_sel_pick(unsigned long long* sel, int shift) {          // <<<1, 256>>>
    __shared__ uint32_t h[256];
    uint32_t* hist = (uint32_t*)(sel + 2);
    h[threadIdx.x] = hist[threadIdx.x];
    hist[threadIdx.x] = 0u;                                               // ready for the next round
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long k = sel[1], below = 0;
        unsigned d = 255;
        for (unsigned b = 0; b < 256; ++b) {
            if (below + h[b] >= k) { d = b; break; }
            below += h[b];
        }
        sel[0] |= (unsigned long long)d << shift;
        sel[1] = k - below;
    }
}
__global__ __launch_bounds__(256) void k_evict(VoxSlot* slots, int32_t* counts, double* mean, double* cov, unsigned long long* touch, size_t cap,
                                               const unsigned long long* sel) {
    const unsigned long long thr = sel[0];
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (size_t)gridDim.x * blockDim.x) {
        if (counts[i] <= 0 || touch[i] > thr) continue;
        int32_t* p = (int32_t*)&slots[i];
#pragma unroll
        for (int j = 0; j < 10; ++j) p[j] = 0;                            // the 40-byte payload; the six links behind it stay
        counts[i] = 0; touch[i] = 0ull;
        for (int j = 0; j < 3; ++j) mean[i * 3 + j] = 0.0;
        for (int j = 0; j < 6; ++j) cov[i * 6 + j] = 0.0;
    }
}

// The table rebuilt from its voxels alone, ON THE DEVICE (no host round trip of the map, no host sort, no allocation beyond
// the new table itself): every entry that holds points moves into a fresh table — key, payload, count, statistics, LRU stamp
// — and leaves its key in a list; the ghost stage then makes the neighbours and links of that list as it does for the new
// voxels of any batch.  Runs inside a keyframe merge when dropped cells outnumber the voxels eight to one.
__global__ void k_compact_move(const int64_t* ok, const VoxSlot* os, const int32_t* on, const double* om, const double* oc, const unsigned long long* ot,
                               size_t ocap, VoxSlot* ns, int64_t* nk, int32_t* nn, double* nm, double* nc, unsigned long long* nt, uint64_t nmask,
                               int64_t* list, uint32_t* n_list) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ocap) return;
    const int64_t key = ok[i];
    if (key == NDT_KEY_EMPTY || on[i] <= 0) return;       // empty, a ghost cell, or a dropped voxel
    uint64_t slot = ndt_hash_key(key) & nmask;
    for (;;) {
        unsigned long long prev = atomicCAS((unsigned long long*)&nk[slot], (unsigned long long)NDT_KEY_EMPTY, (unsigned long long)key);
        if ((int64_t)prev == NDT_KEY_EMPTY) break;
        slot = (slot + 1) & nmask;
    }
    if (slot < 4) nk[nmask + 1 + slot] = key;
    VoxSlot v = os[i];
#pragma unroll
    for (int j = 0; j < 6; ++j) v.link[j] = NDT_LINK_NONE;      // the ghost stage links the new table
    ns[slot] = v;
    nn[slot] = on[i];
    for (int j = 0; j < 3; ++j) nm[slot * 3 + j] = om[i * 3 + j];
    for (int j = 0; j < 6; ++j) nc[slot * 6 + j] = oc[i * 6 + j];
    if (ot) nt[slot] = ot[i];
    list[atomicAdd(n_list, 1u)] = key;
}
static int map_compact(VoxelMap& m, MapWork& w, const MapBuildCfg& cfg, hipStream_t s, std::string& err) {
    (void)cfg;
    const size_t nv = (size_t)m.n_vox;
    if (getenv("NDT_DEBUG_LRU")) fprintf(stderr, "[ndt] table rebuilt on the device from %zu voxels (%lld entries before)\n", nv, (long long)m.n_entries);
    VoxelMap nm;
    nm.track_touch = m.track_touch; nm.cloud_no = m.cloud_no;
    HIP_TRY(map_alloc(nm, pow2_at_least(std::max<size_t>(nv, 1) * 5), s));      // voxels + their ghost cells at the usual ratio, load <= 1/2
    HIP_TRY(w.misc.ensure(256));
    HIP_TRY(w.keys_a.ensure(std::max<size_t>(nv, 1) * 8));
    int32_t* plan = w.misc.as<int32_t>();
    HIP_TRY(hipMemsetAsync((char*)w.misc.p + 160, 0, 32, s));
    HIP_TRY(hipMemsetAsync((char*)w.misc.p + 4 * PL_NSEG, 0, 28, s));       // nseg .. sort error
    k_compact_move<<<(unsigned)((m.cap + 255) / 256), 256, 0, s>>>(m.keys, m.slots, m.counts, m.mean, m.cov, m.touch, m.cap, nm.slots, nm.keys, nm.counts, nm.mean,
                                                                  nm.cov, nm.touch, (uint64_t)(nm.cap - 1), w.keys_a.as<int64_t>(), (uint32_t*)(plan + PL_NSEG));
    // the host's voxel count is exact, so the list's length is known without a read-back
    k_set_word<<<1, 1, 0, s>>>(plan + PL_NEW_VOX, (int32_t)nv);
    k_set_word<<<1, 1, 0, s>>>(plan + PL_NEW_ENT, (int32_t)nv);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s));        // the old table is read until here
    map_free(m);
    m = nm;                                  // n_vox = n_entries = 0: the ghost stage adds the plan's counts
    SegKeys K{};
    K.plain = w.keys_a.as<int64_t>(); K.d_count = plan + PL_NSEG;
    PlanWords pw;
    int rc = ghost_stage(m, w, K, (uint32_t)nv, PL_NEW_VOX, PL_NEW_ENT, &pw, s, err);
    if (rc == NDT_OK && (size_t)pw.nseg != nv) { err = "table rebuild: the device moved " + std::to_string(pw.nseg) + " voxels, the host counted " + std::to_string(nv); return NDT_ERR_HIP; }
    return rc;
}

int map_lru_step(VoxelMap& m, MapWork& w, const MapBuildCfg& cfg, hipStream_t s, std::string& err) {
    if (cfg.capacity <= 0 || !m.touch || !m.cap) return NDT_OK;
    m.cloud_no += 1;
    if (w.last_n) {
        RtArg rt{};
        for (int i = 0; i < 12; ++i) rt.v[i] = w.last_rt[i];
        const unsigned nb = (unsigned)std::min<size_t>(2048, (w.last_n + 255) / 256);
        k_touch_points<<<nb, 256, 0, s>>>(w.last_in, w.last_y, w.last_z, w.last_n, w.last_stride, w.last_has_T, rt, cfg.kp, m.keys, (uint64_t)(m.cap - 1), m.touch,
                                          (unsigned long long)m.cloud_no << 32);
        HIP_TRY(hipGetLastError());
    }
    if (m.n_vox < (int64_t)cfg.capacity) return NDT_OK;
    const unsigned long long k = (unsigned long long)(m.n_vox - ((int64_t)cfg.capacity - 1));
    HIP_TRY(w.sel.ensure(16 + 256 * 4));
    unsigned long long* sel = w.sel.as<unsigned long long>();
    const unsigned nb = (unsigned)std::min<size_t>(1024, (m.cap + 255) / 256);
    k_sel_init<<<1, 256, 0, s>>>(sel, k);
    for (int shift = 56; shift >= 0; shift -= 8) {
        k_sel_hist<<<nb, 256, 0, s>>>(m.counts, m.touch, m.cap, sel, shift, (uint32_t*)(sel + 2));
        k_sel_pick<<<1, 256, 0, s>>>(sel, shift);
    }
    k_evict<<<nb, 256, 0, s>>>(m.slots, m.counts, m.mean, m.cov, m.touch, m.cap, sel);
    HIP_TRY(hipGetLastError());
    m.n_vox -= (int64_t)k;                        // stamps are distinct (a point lies in one voxel): exactly k voxels went
    if (m.n_entries > 8 * std::max<int64_t>(m.n_vox, 4096)) return map_compact(m, w, cfg, s, err);
    return NDT_OK;
}
