// ndt_internal.h — host-side declarations shared by the engine's translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/ndt_mi355.h"
#include "ndt_device.h"

struct DevBuf {   // grow-only device allocation
    void* p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need) {
        if (need <= bytes) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); if (e != hipSuccess) return e; p = nullptr; bytes = 0; }
        size_t want = need + need / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) bytes = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <class T> T* as() const { return (T*)p; }
};

struct VoxelMap {          // the GPU-resident voxel hash
    VoxSlot* slots = nullptr;
    int64_t* keys = nullptr;   // [cap+4] dense copy of the slot keys (what a probe reads: 8 B per slot);
                               // keys[cap..cap+3] mirror keys[0..3] so a 32-byte four-key read never wraps
    int32_t* counts = nullptr; // [cap] points merged into the voxel (0: ghost cell)
    double* mean = nullptr;    // [cap*3] running mean offset from the voxel origin
    double* cov = nullptr;     // [cap*6] running covariance, xx xy xz yy yz zz
    size_t cap = 0;            // power of two (0 = no map)
    int64_t n_vox = 0;         // voxels that hold points
    int64_t n_entries = 0;     // occupied slots: voxels + ghost cells (what the load factor counts)
    // ndt_config.capacity > 0 only: [cap] per slot, (cloud number << 32) | index of the cloud's last point in the voxel
    unsigned long long* touch = nullptr;
    bool track_touch = false;  // allocate / carry `touch` (set before the table is made)
    uint32_t cloud_no = 0;     // clouds merged into this map so far
};

struct SortKeyMap {        // packed key <-> compact radix-sort key for one batch; entry 0 is the most significant axis
    int32_t mn[3];
    int32_t bits[3];
    int32_t axis[3];       // which cell coordinate (0 x, 1 y, 2 z) each entry is
};
struct SortedCells {       // result of sort_points_by_cell
    size_t n_valid = 0;
    uint32_t nseg = 0;              // 0 when the count was left on the device (d_nseg)
    const uint32_t* d_nseg = nullptr;
    uint64_t bbox_cells = 0;        // cells in the bounding box (an upper bound of nseg known without a read-back)
    SortKeyMap km = {};
    int passes = 0;
    const uint64_t* sorted_keys = nullptr;
    const uint32_t* sorted_vals = nullptr;  // index of each sorted point in the unsorted arrays (w.tx/ty/tz)
};

struct MapBuildCfg {
    KeyParams kp;
    int32_t min_pts;
    double eig_floor_ratio;
    int32_t max_pts;         // > 0: voxels holding more points than this are frozen (IncNdt3d)
    int32_t info_rule;       // NDT_INFO_*
    int32_t capacity;        // ndt_config.capacity (host side only: map_lru_step)
};

// ---- ndt_map.hip
hipError_t map_alloc(VoxelMap& m, size_t cap, hipStream_t s);
void map_free(VoxelMap& m);
hipError_t map_clear(VoxelMap& m, hipStream_t s);
struct MapWork {
    DevBuf raw, tx, ty, tz, block_rows, keys_a, keys_b, vals_a, vals_b, hist, flags, flags2, segid, seg_start, scan_tmp, misc, big_pos, big_list, big_sums;
    int32_t* h_plan = nullptr;       // pinned: the 256-byte plan block (misc) as last read back
    int32_t* h_plan_dev = nullptr;   // the same memory as the device addresses it (kernels leave their results there)
    int32_t plan_seq = 0;            // sequence number of the last result requested through the pinned block
    unsigned os_resident = 0;        // tiles of the one-sweep radix pass the device holds at once (0: not asked yet)
    // the batch map_merge_points saw last, as the device reads it (map_lru_step walks it once more, inside the same call)
    const unsigned char* last_in = nullptr; const float* last_y = nullptr; const float* last_z = nullptr;
    size_t last_n = 0, last_stride = 0; int last_has_T = 0; double last_rt[12] = {0};
    DevBuf sel;                      // radix-select state of map_lru_step: {prefix, k} + 256 digit counts
    void release() {
        if (h_plan) { (void)hipHostFree(h_plan); h_plan = nullptr; h_plan_dev = nullptr; }
        DevBuf* all[] = {&raw, &tx, &ty, &tz, &block_rows, &keys_a, &keys_b, &vals_a, &vals_b, &hist, &flags, &flags2, &segid, &seg_start, &scan_tmp, &misc, &big_pos, &big_list, &big_sums, &sel};
        for (DevBuf* b : all) b->release();
    }
};
// Merges n points (raw records, xyz float32 at offset 0 of each `stride`-byte record, host or device
// memory) into the map; T (device-independent, row-major R | t) may be null.  Returns the number of
// radix passes executed through *passes.
int map_merge_points(VoxelMap& m, MapWork& w, const MapBuildCfg& cfg, const void* pts, const float* soa_y, const float* soa_z, size_t n,
                     size_t stride, const double* R, const double* t, hipStream_t s, int* passes, std::string& err);
// Queues the work and leaves the number of centroids on the device (*d_n_out; nullptr if the cloud was empty): the
// caller reads it back with its own final synchronisation instead of a second one in the middle.
int downsample_points(MapWork& w, const KeyParams& leaf_kp, const void* pts, size_t n, size_t stride, const double* R, const double* t,
                      DevBuf& ox, DevBuf& oy, DevBuf& oz, const uint32_t** d_n_out, hipStream_t s, std::string& err);
int map_export(const VoxelMap& m, const KeyParams& kp, std::vector<int64_t>& keys, std::vector<int32_t>& counts,
               std::vector<double>& mean, std::vector<double>& cov, std::vector<double>& info,
               std::vector<uint8_t>& valid, hipStream_t s, std::string& err);
int map_dump(const VoxelMap& m, std::vector<int64_t>& keys, std::vector<int32_t>& counts, std::vector<double>& mean, std::vector<double>& cov,
             hipStream_t s, std::string& err, std::vector<unsigned long long>* touch = nullptr);
int map_restore(VoxelMap& m, MapWork& w, const MapBuildCfg& cfg, const int64_t* keys, const int32_t* counts, const double* mean, const double* cov,
                size_t n, hipStream_t s, std::string& err, const unsigned long long* touch = nullptr);
// ndt_config.capacity (IncNdt3d's LRU voxel list): stamps the voxels the last merged batch touched and, once the map holds
// `capacity` voxels or more, turns the least recently touched ones back into empty cells until capacity - 1 are left
int map_lru_step(VoxelMap& m, MapWork& w, const MapBuildCfg& cfg, hipStream_t s, std::string& err);
hipError_t map_track_touch(VoxelMap& m, bool on, hipStream_t s);

// ---- PCL compatibility mode (ndt_pcl.hip)
struct PclEngine {            // what a pass needs from the handle
    hipStream_t stream;
    const int64_t* keys; const VoxSlot* slots; uint32_t mask;
    const float *sx, *sy, *sz; size_t n;
    KeyParams kp;
    double* partials;         // pcl_grid(n) rows of ACC_N doubles
    double* d_row;            // ACC_N doubles, device
    double* h_row;            // ACC_N doubles, pinned host
    int use_links;            // 9 hashed cells + x-links instead of 27 hashed cells
    int n_cu;                 // compute units of the handle's device
};
int pcl_grid(size_t n, int n_cu);
void pcl_matrix_from_vector(const double p[6], double T[16]);
void pcl_vector_from_matrix(const double T[16], double p[6]);
int pcl_eval(const PclEngine& e, const ndt_pcl_params& prm, const double p[6], double* score, double g[6], double* H, int64_t* n_nb, std::string& err);
int pcl_time_pass(const PclEngine& e, const ndt_pcl_params& prm, const double p[6], bool with_hessian, int reps, double* us_per_launch,
                  double* voxels_per_point, double* neighbours_per_point, std::string& err);
int pcl_align(const PclEngine& e, const ndt_pcl_params& prm, int max_iterations, double trans_eps, const double guess[16], ndt_pcl_result* out,
              std::string& err);
hipError_t unpack_points_soa(const void* d_raw, size_t n, size_t stride, float* x, float* y, float* z, hipStream_t s);

// ---- ndt_residual.hip
// n_cu: compute units of the handle's device (256 on MI355X; read at ndt_create and kept in the handle): sizes every grid
int residual_grid(size_t n, int n_cu, size_t problems = 1, int nearby = 7);
// chained single-scan iteration: previous launch's reduction + step in every block's prologue, then the residual pass
// Peer exchange of the sharded loop (ndt_comm_p2p_*): every rank owns a mailbox in its own HBM that the other ranks write
// into over xGMI — [2 parities][NDT_MAX_PEERS ranks][ACC_N] float64 rows, then [2][NDT_MAX_PEERS] 64-bit sequence flags.
#define NDT_MAX_PEERS 16
#define PX_ROWS_BYTES (2 * NDT_MAX_PEERS * ACC_N * 8)
#define PX_BYTES (PX_ROWS_BYTES + 2 * NDT_MAX_PEERS * 8)
// Records still to be unpacked (ndt_set_source_async): the first launch that reads the source takes them from here and
// leaves the three-array copy behind for everything after it.  p == nullptr: the arrays are current.
struct AosIn { const unsigned char* p; uint32_t stride; float *ox, *oy, *oz; };
struct PeerWait { const unsigned long long* flags; int n; unsigned long long seq; };   // flags == nullptr: nothing to wait for
struct PeerRows { double* rows[NDT_MAX_PEERS]; unsigned long long* flags[NDT_MAX_PEERS]; };   // every rank's mailbox at the parity in use
void launch_residual_chain(const float* x, const float* y, const float* z, int n, const int64_t* keys, const VoxSlot* slots, uint32_t mask,
                           const GnState* st_in, GnState* st_out, const double* prev_partials, int n_prev, const SolveParams& sp,
                           const EvalParams& ep, double* partials, int grid, int linked, const GnState* init_state, const PeerWait& pw,
                           const AosIn& aos, hipStream_t s);
void launch_reduce_rows_p2p(const double* partials, int n_partials, const GnState* st, const PeerRows& peers, int n_ranks, int rank,
                            unsigned long long seq, int force, hipStream_t s);
// (init_state: host pointer or nullptr; given, the launch starts from a fresh state at that pose instead of reading st_in)
void launch_residual(const float* x, const float* y, const float* z, int n, const int64_t* keys, const VoxSlot* slots, uint32_t mask,
                     const GnState* st, const EvalParams& ep, double* partials, int grid, int force,
                     int64_t* keys_out, uint8_t* mask_out, int linked, hipStream_t s);
void launch_reduce_solve(const double* partials, int n_partials, GnState* st, const SolveParams& sp, int do_solve,
                         int force, GnState* mirror, unsigned long long mirror_seq, const PeerWait& pw, hipStream_t s);
void launch_reduce_rows(const double* partials, int n_partials, const GnState* st, double* row_out, int force, hipStream_t s);
void launch_solve_only(GnState* st, const double* reduced, const SolveParams& sp, int do_solve, int force, hipStream_t s);
void launch_residual_batch(const float* x, const float* y, const float* z, const int64_t* scan_off, const int32_t* pair_scan,
                           const int64_t* keys, const VoxSlot* slots, uint32_t mask, const GnState* states, const EvalParams& ep,
                           double* partials, int rows_per_pair, int pair0, int n_pairs, int force, int linked, hipStream_t s);
void launch_reduce_solve_batch(const double* partials, int rows_per_pair, GnState* states, int n_pairs, const SolveParams& sp,
                               int do_solve, int force, int32_t* n_done, hipStream_t s);
