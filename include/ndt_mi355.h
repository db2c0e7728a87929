/*
 * ndt_mi355.h — C ABI of the MI355X-native NDT scan-matching engine.
 *
 * This is the drop-in boundary for the point-cloud-matching step of
 * ergevv/lio-slam.  Every entry point names the reference call it replaces
 * (paths relative to the reference checkout).  The reference does not reach NDT
 * through an FFI: it holds a `pcl::NormalDistributionsTransform<Point,Point>`
 * by value (src/slam/include/process.h:86).  A maintainer swaps that member for
 * the C++ facade in ndt_mi355.hpp, which calls exactly the functions below.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++ or torch types cross this boundary;
 *   - every function returns an int status (NDT_OK == 0, negative on error) and
 *     never throws; ndt_last_error() gives the text of the last failure;
 *   - point buffers are arrays of records whose first 12 bytes are x,y,z as
 *     float32 (stride 32 for slam_czc::Point, src/slam/include/utility.h:54-61;
 *     12 or 16 for packed clouds).  Non-finite points are skipped, as the PCL
 *     filters in front of the reference NDT do (src/slam/src/process.cc:68-73);
 *   - point buffers may live in host memory or in device (HBM) memory: the
 *     engine copies them with hipMemcpyDefault, and the caller's buffer is free
 *     again when the call returns;
 *   - 4x4 transforms are double[16], column-major (Eigen's default layout,
 *     src/slam/src/utility.cc:22-35 builds the guess that way);
 *   - the 6x6 Hessian is double[36] (symmetric, so row- == column-major) in the
 *     order [rotation(3), translation(3)] with a right-multiplicative rotation
 *     perturbation: the coordinates of PointCloudFactor's residual
 *     (src/slam/include/factor.h:243-251), so it can be passed as `ndt_info`
 *     (src/slam/src/process.cc:205-211);
 *   - a handle is not thread-safe; distinct handles are independent.
 */
#ifndef NDT_MI355_H
#define NDT_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ndt_handle ndt_handle;

enum {
    NDT_OK = 0,
    NDT_ERR_INVALID_ARG = -1,
    NDT_ERR_NO_TARGET = -2,
    NDT_ERR_NO_SOURCE = -3,
    NDT_ERR_HIP = -4,
    NDT_ERR_TOO_FEW_POINTS = -5, /* fewer accepted residuals than min_effective_pts */
    NDT_ERR_SINGULAR = -6,       /* H not positive definite even after regularisation */
    NDT_ERR_KEY_RANGE = -7,      /* a coordinate falls outside the 21-bit-per-axis key range */
    NDT_ERR_COMM = -8,           /* RCCL not available / communicator failure */
    NDT_ERR_CAPACITY = -9        /* caller-provided output buffer too small */
};

enum { NDT_KEY_FLOOR_F32 = 0, NDT_KEY_ROUND_F64 = 1 };

/* Replaces the four hard-coded setters at src/slam/src/process.cc:373-376
 * (setResolution 1.0, setMaximumIterations 35, setTransformationEpsilon 0.01,
 * setStepSize 0.1 — the last has no Gauss-Newton counterpart) plus the PCL /
 * SLAM-book defaults listed in SURVEY.md §8(a) row A1/A6. */
typedef struct ndt_config {
    float resolution;        /* voxel edge, metres (process.cc:373)                        */
    int32_t max_iterations;  /* Gauss-Newton iteration cap (process.cc:374)                */
    double eps;              /* stop when |dx| < eps (process.cc:375); 0 = run all iters   */
    int32_t nearby;          /* 1 = centre voxel only, 7 = centre + 6 face neighbours      */
    int32_t min_pts_in_voxel;/* voxel is matched only if it holds >= this many points (6)  */
    double res_outlier_th;   /* reject a residual when its chi^2 exceeds this (20)         */
    int32_t min_effective_pts;/* align fails below this many accepted residuals (10)       */
    int32_t key_rule;        /* NDT_KEY_FLOOR_F32 (PCL voxel grid) | NDT_KEY_ROUND_F64     */
    double eig_floor_ratio;  /* covariance eigenvalues are raised to ratio*lambda_max (.01)*/
    int32_t final_eval;      /* 1: re-evaluate H,b,chi2 at the returned pose               */
    int32_t device;          /* HIP device ordinal                                          */
    /* Options of the SLAM-book IncNdt3d the reference's intended form was written against (src/slam/reference/
     * lio_preinteg.cc:37-38 sets its options).  Both default to 0 = off / PCL's rule. */
    int32_t max_pts_in_voxel;/* > 0: a voxel that already holds more points than this ignores further merges (IncNdt3d freezes at 50) */
    int32_t info_rule;       /* NDT_INFO_EIG_CLAMP (PCL: eigenvalue floor) | NDT_INFO_ADD_1E3 (IncNdt3d: (Sigma + 1e-3 I)^-1; one point: 100 I) */
    int32_t capacity;        /* > 0: at most capacity - 1 voxels; after a merge the least recently touched ones are dropped
                              * (IncNdt3d keeps its voxels in an LRU list of capacity_ = 100 000: a point moves its voxel to
                              * the front, a new voxel evicts the tail).  "Touched" orders voxels by the last cloud that put
                              * a point into them, then by that point's index in the cloud: the list order IncNdt3d ends a
                              * cloud with, as long as a cloud does not re-enter a voxel that the same cloud evicted. */
    int32_t reserved0;       /* keeps the struct's size a multiple of 8 */
} ndt_config;
enum { NDT_INFO_EIG_CLAMP = 0, NDT_INFO_ADD_1E3 = 1 };

/* Output of ndt_align.  Replaces getFinalTransformation() (process.cc:95) and
 * adds what the reference ignores (hasConverged, iteration count) and what the
 * north star asks for (the Hessian for the Ceres factor). */
typedef struct ndt_result {
    double T[16];        /* final pose, column-major 4x4                                   */
    double H[36];        /* sum J^T Sigma^-1 J, [rot, trans] order                          */
    double b[6];         /* sum -J^T Sigma^-1 e                                             */
    double chi2;         /* sum of accepted chi^2                                           */
    double last_dx_norm; /* |dx| of the last Gauss-Newton step                              */
    int64_t n_effective; /* accepted residuals in the reported evaluation                   */
    int32_t iterations;  /* Gauss-Newton steps taken                                        */
    int32_t converged;   /* 1 if |dx| < eps was reached                                     */
    int32_t status;      /* NDT_OK or the error of this alignment (what ndt_align returns)  */
    int32_t reserved;
} ndt_result;

/* Fills *cfg with the defaults chosen in SURVEY.md Appendix B. */
void ndt_default_config(ndt_config* cfg);

/* Constructor / destructor of the `ndt_` member (process.h:86). */
ndt_handle* ndt_create(const ndt_config* cfg);
void ndt_destroy(ndt_handle* h);

/* Re-applies the scalar parameters (the process.cc:373-376 setters) without
 * touching the resident map or source.  resolution/key_rule/min_pts/eig floor
 * changes take effect at the next ndt_set_target. */
int ndt_configure(ndt_handle* h, const ndt_config* cfg);

/* ndt_.setInputTarget(pc_key_) (process.cc:78, 147): discard the resident voxel
 * map and rebuild it from `n` points. */
int ndt_set_target(ndt_handle* h, const void* pts, size_t n, size_t stride_bytes);

/* The keyframe merge of process.cc:134-147 (transformPointCloud + `+=` +
 * setInputTarget) done incrementally, IncNdt3d::AddCloud style
 * (src/slam/reference/lio_preinteg.cc:108,133): transform the points by T (NULL
 * = identity), round to float32, and merge them into the resident voxel hash
 * with the Gaussian-merge rule of src/slam/reference/math_utils.h:87-98. */
int ndt_add_target(ndt_handle* h, const void* pts, size_t n, size_t stride_bytes,
                   const double T[16]);

/* ndt_.setInputSource(pc_trans_filter) (process.cc:87).  The records are copied: the caller's buffer is free on return. */
int ndt_set_source(ndt_handle* h, const void* pts, size_t n, size_t stride_bytes);
/* The same for records that already lie in this device's memory (a LiDAR driver or a pre-processing kernel left them
 * there), stream-ordered: the call returns at once; the records are read on the handle's stream by the first kernel that
 * needs the source (the first launch of ndt_align's loop unpacks them by the way; any other reader runs an unpack kernel
 * first).  d_pts must stay valid and unchanged until a later call on this handle that reads the source has returned
 * (ndt_align, ndt_eval, ndt_add_target_from_source, ndt_get_source, ...).  A host pointer is refused (NDT_ERR_INVALID_ARG). */
int ndt_set_source_async(ndt_handle* h, const void* d_pts, size_t n, size_t stride_bytes);

/* Scan pre-processing on the device, so a LiDAR frame crosses PCIe once:
 * pcl::transformPointCloud(LiDAR->IMU extrinsic) + pcl::VoxelGrid(leaf).filter() of src/slam/src/process.cc:62-73,
 * then setInputSource (process.cc:87).  T may be NULL (no transform).  One centroid (xyz only) per occupied leaf
 * cell, cells in PCL's index order (x fastest, z slowest); non-finite points are skipped as PCL does.
 * Differences from the reference, on purpose: pcl::VoxelGrid also averages the other fields of slam_czc::Point
 * (utility.h:54-61: intensity, time, ring); no later stage of this path reads them, so they are not carried.  And the
 * reference re-downsamples the WHOLE keyframe map in place after every keyframe (process.cc:137-146) before rebuilding
 * the NDT target from it; here a keyframe is merged into the voxel statistics instead (ndt_add_target*, below), which
 * keeps every point's contribution rather than one centroid per 0.5 m cell — the map's voxel means and covariances
 * therefore differ from a rebuilt-from-downsampled map's by the downsampling error, not by construction. */
int ndt_set_source_downsampled(ndt_handle* h, const void* pts, size_t n, size_t stride_bytes, float leaf,
                               const double T[16]);
/* Reads the resident source back as packed xyz (xyz may be NULL to ask for the size only). */
int ndt_get_source(ndt_handle* h, float* xyz, size_t cap_points, size_t* n_out);
/* The keyframe step of process.cc:134-147 without a host round trip: the resident source, transformed by T, is
 * merged into the resident map (what `*pc_key_ += *output; setInputTarget(pc_key_)` does with pc_trans_filter;
 * without the in-place re-downsample of process.cc:137-146, see above). */
int ndt_add_target_from_source(ndt_handle* h, const double T[16]);

/* ndt_.align(*output, guess) + getFinalTransformation() (process.cc:94-95).
 * Returns NDT_OK also when the iteration cap was hit (out->converged == 0),
 * as the reference never checks convergence. */
int ndt_align(ndt_handle* h, const double guess[16], ndt_result* out);

/* ndt_align for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE (which has no device memory at all): it is
 * for callers that may not call hipMalloc — tight serving loops, harnesses that forbid allocation after start-up, code
 * that runs between the captures of its own HIP graphs.  (It is NOT itself capturable: like ndt_align it returns its result
 * on the host, so it waits for the handle's stream — hipMemsetAsync + kernels + a bounded poll / hipStreamSynchronize —
 * before it returns.)  The source cloud is handed over as three float32 arrays the caller holds in device memory (x[n],
 * y[n], z[n]; non-finite points are skipped), the map is the handle's resident one, the result goes to *out as in ndt_align.
 * The loop's scratch — two buffers of per-block partial rows and two Gauss-Newton states, 132 096 bytes — lives in a
 * file-scope `static __device__` array of the library instead of allocated buffers (every GPU the library runs on holds
 * its own copy until the process ends), so this call makes no hipMalloc / hipFree / allocating library call
 * (tests/test_zz_no_alloc.py: hipMemGetInfo reads the same free bytes before and after).
 *   - n may be at most NDT_STATIC_MAX_POINTS; more is rejected with NDT_ERR_CAPACITY (use ndt_set_source + ndt_align);
 *   - not available on a handle with a communicator (ndt_comm_init) or with profiling enabled: NDT_ERR_INVALID_ARG;
 *   - all calls of a process on one GPU share that GPU's array (also across handles): the library serialises them with a
 *     per-device lock held for the duration of the call, so handles on several host threads take turns instead of racing.
 *     ndt_align and every other entry point keep using the handle's own buffers and are not affected;
 *   - on MI355X the launch geometry equals ndt_align's for the same n, so both return the same bits. */
#define NDT_STATIC_MAX_POINTS (1u << 22)
int ndt_align_static(ndt_handle* h, const float* d_x, const float* d_y, const float* d_z, size_t n, const double guess[16],
                     ndt_result* out);

/* One residual/accumulation pass at pose T: the unit of work inside align's
 * loop; no reference counterpart is callable on its
 * own.  Any output pointer may be NULL. */
int ndt_eval(ndt_handle* h, const double T[16], double H[36], double b[6],
             double* chi2, int64_t* n_effective);

/* Cell assignment of the resident source at pose T, for bit-exact parity
 * checks: keys[i] is the packed key of point i's centre voxel (or INT64_MIN for
 * a skipped point), hit_mask[i] has bit j set when probe j found a valid voxel
 * and its residual passed the chi^2 gate. */
int ndt_source_cells(ndt_handle* h, const double T[16], int64_t* keys, uint8_t* hit_mask,
                     size_t cap);

/* Number of source points resident (after non-finite points were dropped). */
int64_t ndt_source_size(const ndt_handle* h);
/* Number of voxels resident in the map (valid or not). */
int64_t ndt_num_voxels(const ndt_handle* h);

/* Dump of the voxel map for key/statistics parity checks; rows are sorted by
 * packed key.  keys[cap], counts[cap], mean[cap*3], cov[cap*9], info[cap*9]
 * (info = what the residual kernel reads, zero for voxels that are not valid),
 * valid[cap].  Any array may be NULL. */
int ndt_export_voxels(ndt_handle* h, int64_t* keys, int32_t* counts, double* mean,
                      double* cov, double* info, uint8_t* valid, size_t cap, size_t* n_out);

/* SURVEY.md §8(f) row N3 (map serialisation) — the reference keeps pc_key_ only in RAM and rebuilds the voxel grid from
 * it (src/slam/src/process.cc:78,147); a resident map that was grown by incremental merges has no such cloud to
 * rebuild from, so it can be saved and restored.  The blob is position-independent, little-endian:
 *   bytes 0-7 "NDTMI355", u32 version (1), f32 resolution, i32 key_rule, i32 reserved, u64 n_voxels, 32 reserved bytes
 *   (64-byte header), then int64 keys[n] (ascending), float64 mean_offset[n][3] (relative to key*resolution),
 *   float64 cov[n][6] (upper triangle), int32 counts[n].
 * ndt_map_save: *n_bytes = size needed; writes only when buf != NULL and cap_bytes is large enough (else
 * NDT_ERR_CAPACITY when buf != NULL).  ndt_map_load replaces the resident map; the blob's resolution and key_rule
 * must equal the handle's; validity and the float32 payload are derived again under the handle's current
 * min_pts_in_voxel / eig_floor_ratio, so a map loaded under the configuration it was saved with answers bit for bit
 * like the original and later ndt_add_target calls continue from the same statistics.  The LRU order of
 * ndt_config.capacity is not stored: a loaded map counts as one cloud whose points came in file (ascending key) order,
 * and nothing is dropped until the next merge. */
int ndt_map_save(ndt_handle* h, void* buf, size_t cap_bytes, size_t* n_bytes);
int ndt_map_load(ndt_handle* h, const void* buf, size_t n_bytes);

/* SURVEY.md §8(f) row N2 — the NDT Hessian as PointCloudFactor's information matrix in place of the constant
 * diagonal at src/slam/src/process.cc:205-208.  info = scale * H with H symmetrised and its eigenvalues raised to
 * rel_floor * lambda_max, so that it is positive definite even in a degenerate scene (a corridor leaves H
 * rank-deficient) and `Eigen::LLT(info)` at src/slam/include/factor.h:224 cannot fail.  Host-only. */
int ndt_information_from_hessian(const double H[36], double scale, double rel_floor, double info[36]);

/* ---- SURVEY.md §8(f) row N4 — PCL compatibility mode ------------------------------------------------------
 * The objective, derivatives and optimiser the reference's live build executes behind ndt_.align()
 * (src/slam/src/process.cc:94; parameters process.cc:373-376): pcl::NormalDistributionsTransform of PCL >= 1.12
 * (README.md:52 names PCL 1.14) as restated in SURVEY.md Appendix A — the PCL sources are not in this image, so
 * this mode is "parity unpinned" like the rest:
 *   - transform vector p = [tx ty tz, rx ry rz], R = Rx(rx) Ry(ry) Rz(rz) (Eigen eulerAngles(0,1,2));
 *   - the source cloud is transformed in float32 (PCL's Scalar = float);
 *   - neighbours of a transformed point = voxels with >= min_pts_in_voxel points whose centroid lies within
 *     `resolution` of it (radius search on the centroid kd-tree; here: the 27 cells around the point's cell + the
 *     distance test, which gives the same set);
 *   - score = sum -d1 exp(-d2/2 x'^T C x'), gradient and Hessian of Magnusson 2009 eq. 6.12 / 6.13 with the
 *     Gaussian constants d1, d2 from outlier_ratio and resolution;
 *   - Newton direction from an SVD solve, More-Thuente line search (mu 1e-4, nu 0.9, <= 10 trials) between
 *     eps/2 and step_size, convergence when iterations >= max_iterations or |dt|^2 <= eps (PCL compares the
 *     SQUARED translation with eps) and, if rotation_eps > 0, cos(angle) >= rotation_eps.
 * Uses the handle's resolution, max_iterations, eps and the resident map / source; `nearby`, `res_outlier_th` and
 * `min_effective_pts` do not apply. */
typedef struct ndt_pcl_params {
    double outlier_ratio;        /* 0.55, PCL's default (the reference never sets it)                        */
    double step_size;            /* 0.1, ndt_.setStepSize(0.1) at process.cc:376: More-Thuente maximum step  */
    double rotation_eps;         /* 0: transformation_rotation_epsilon_ is left unset by the reference       */
    int32_t hessian_d1_as_pcl;   /* 1: row d1 of the angular Hessian as PCL has it (+sin ry in its third
                                    entry); 0: Magnusson eq. 6.21 (-sin ry)                                  */
    int32_t reserved;
} ndt_pcl_params;

typedef struct ndt_pcl_result {
    double T[16];                /* final transform, column-major                                            */
    double p[6];                 /* final transform vector                                                   */
    double score;                /* PCL's score at the final transform                                       */
    double trans_probability;    /* getTransformationProbability(): score / number of source points          */
    double gradient[6];          /* score gradient and Hessian at the final transform (row-major 6x6)        */
    double hessian[36];
    int64_t n_neighbours;        /* (point, voxel) pairs in the last evaluation                              */
    int32_t iterations;          /* getFinalNumIteration()                                                   */
    int32_t converged;           /* hasConverged()                                                           */
    int32_t status;
    int32_t evaluations;         /* passes over the source cloud (derivative + Hessian-only evaluations)     */
} ndt_pcl_result;

void ndt_pcl_default_params(ndt_pcl_params* prm);
/* One computeDerivatives pass at transform vector p: score, gradient g[6], Hessian H[36] (row-major; skipped when
 * H == NULL), number of (point, voxel) pairs.  The unit the parity tests call. */
int ndt_pcl_eval(ndt_handle* h, const ndt_pcl_params* prm, const double p[6], double* score, double g[6], double H[36],
                 int64_t* n_neighbours);
/* computeTransformation from `guess` (column-major 4x4).  NDT_OK also when the iteration cap ended the loop. */
int ndt_pcl_align(ndt_handle* h, const ndt_pcl_params* prm, const double guess[16], ndt_pcl_result* out);
/* Measurement helper (bench.py): mean microseconds of the derivative pass alone at p, HIP events around `reps`
 * back-to-back launches; valid voxels met in the 27 cells and neighbours accepted, per source point. */
int ndt_pcl_time_pass(ndt_handle* h, const ndt_pcl_params* prm, const double p[6], int with_hessian, int reps,
                      double* us_per_launch, double* voxels_per_point, double* neighbours_per_point);
/* Host helpers of the mode: p <-> matrix (Eigen's eulerAngles(0,1,2) convention, first angle in [0, pi]). */
void ndt_pcl_vector_from_matrix(const double T[16], double p[6]);
void ndt_pcl_matrix_from_vector(const double p[6], double T[16]);

/* Packed-key helpers (21 bits per axis, biased by 2^20). */
int64_t ndt_pack_key(int32_t ix, int32_t iy, int32_t iz);
void ndt_unpack_key(int64_t key, int32_t* ix, int32_t* iy, int32_t* iz);

/* Offline replay (BASELINE.json configs[4]): align `n_pairs` hypotheses against
 * the resident map in one batched Gauss-Newton loop.  `pts` holds all scans
 * back to back; scan s is points [scan_offsets[s], scan_offsets[s+1]);
 * pair p uses scan pair_scan[p] (NULL: pair p uses scan p) from guesses[16*p].
 * Returns NDT_OK when the batch ran; results[p].status tells how pair p ended. */
int ndt_align_batch(ndt_handle* h, const void* pts, size_t stride_bytes,
                    const int64_t* scan_offsets, size_t n_scans, const int32_t* pair_scan,
                    size_t n_pairs, const double* guesses, ndt_result* results);

/* The same in two steps, so that the scans stay resident in HBM across many batched aligns. */
int ndt_set_batch_sources(ndt_handle* h, const void* pts, size_t stride_bytes, const int64_t* scan_offsets,
                          size_t n_scans);
int ndt_align_batch_resident(ndt_handle* h, const int32_t* pair_scan, size_t n_pairs, const double* guesses,
                             ndt_result* results);

/* Large clouds (BASELINE.json configs[3]): each rank holds a shard of the source
 * and a replica of the map; ndt_align/ndt_eval then all-reduce the 29 H/b/chi2/n
 * terms over RCCL on every Gauss-Newton iteration.  Rank 0 calls
 * ndt_comm_unique_id, the caller distributes the 128 bytes, every rank calls
 * ndt_comm_init. */
int ndt_comm_unique_id(char id_out[128]);
int ndt_comm_init(ndt_handle* h, int rank, int n_ranks, const char id[128]);
int ndt_comm_destroy(ndt_handle* h);
/* Ranks of the attached communicator as RCCL counts them, or the ranks of the peer exchange below (1 when neither is
 * attached): what bench.py reports as ranks_seen. */
int ndt_comm_size(const ndt_handle* h);

/* The same sharded loop without a collective library, for ranks of ONE node (the case §8(e) sizes: 256 bytes per
 * iteration, latency-bound): every rank owns a small mailbox in its HBM; on each Gauss-Newton iteration a rank stores its
 * reduced row straight into every rank's mailbox (xGMI peer stores, system-scope, sequence flags after the rows) and the
 * next launch sums all ranks' rows in rank order once they have arrived — two launches per iteration instead of three
 * plus ncclAllReduce, and identical bits on every rank.  ndt_comm_p2p_export allocates the mailbox and returns an opaque
 * handle (NDT_P2P_HANDLE_BYTES: the HIP IPC handle + the GPU's PCI bus id); the caller distributes the handles of all
 * ranks, in rank order, as it would RCCL's unique id; ndt_comm_p2p_init opens the peers' mailboxes (at most 16 ranks).
 * Ranks that share a GPU (rehearsals on a one-GPU box) are detected and wait for the rows in a one-block kernel instead of
 * in every block's prologue, so that the other rank can run.  Every rank must then make the same sequence of
 * ndt_align / ndt_eval calls, as with a collective.  A rank that never delivers makes the others return NDT_ERR_COMM
 * after a bounded wait instead of hanging.  ndt_comm_destroy detaches either kind.  No counterpart in the reference
 * (`process.cc:377`: its one NDT thread). */
#define NDT_P2P_HANDLE_BYTES 128
int ndt_comm_p2p_export(ndt_handle* h, unsigned char handle_out[NDT_P2P_HANDLE_BYTES]);
int ndt_comm_p2p_init(ndt_handle* h, int rank, int n_ranks, const unsigned char* handles /* n_ranks x NDT_P2P_HANDLE_BYTES */);

/* Several GPUs in ONE process (SURVEY.md §8(b) "multi-GPU: ndt_create_group(n_dev, dev_ids)"; the replay half of
 * §8(e)): a group is one member handle per entry of dev_ids (cfg->device is ignored; a device may be listed more than
 * once), every member with a replica of the map.  The group calls fan out to the members on one host thread each:
 * set / add target builds the same map everywhere, align_batch deals the pairs to the members in contiguous shares
 * (each member uploads only the scans its pairs use) and gathers the results in pair order; no collective.  Per-pair
 * results are those of ndt_align_batch on one handle, up to the summation order of a differently sized batch.
 * ndt_group_handle gives a member for per-device calls (ndt_align, ndt_export_voxels, ...).  A sharded source
 * (ndt_comm_*) stays one process per GPU.  No counterpart in the reference, which has one CPU thread. */
typedef struct ndt_group ndt_group;
ndt_group* ndt_create_group(const ndt_config* cfg, int n_dev, const int* dev_ids);
void ndt_destroy_group(ndt_group* g);
int ndt_group_size(const ndt_group* g);
ndt_handle* ndt_group_handle(ndt_group* g, int member);
int ndt_group_set_target(ndt_group* g, const void* pts, size_t n, size_t stride_bytes);
int ndt_group_add_target(ndt_group* g, const void* pts, size_t n, size_t stride_bytes, const double T[16]);
int ndt_group_align_batch(ndt_group* g, const void* pts, size_t stride_bytes, const int64_t* scan_offsets, size_t n_scans,
                          const int32_t* pair_scan, size_t n_pairs, const double* guesses, ndt_result* results);
const char* ndt_group_last_error(const ndt_group* g);

/* Device-time accounting for benchmarks: total microseconds and launches of the
 * residual kernel since the last reset, measured with HIP events on the
 * handle's own stream (enable with ndt_profile(h, 1)). */
int ndt_profile(ndt_handle* h, int enable);
int ndt_profile_read(ndt_handle* h, double* residual_kernel_us, int64_t* residual_launches,
                     double* mean_hits_per_point);

/* Mean duration of the residual kernel alone: `reps` back-to-back launches at pose T between two HIP events on
 * the handle's stream (what bench.py's roofline uses; agrees with rocprofv3 --kernel-trace). */
int ndt_time_residual(ndt_handle* h, const double T[16], int reps, double* us_per_launch);

/* The same for the launch the single-scan Gauss-Newton loop actually makes per iteration (k_residual_chain: the previous
 * launch's reduction + step in the prologue, then the residual pass), `reps` of them back to back from pose T. */
int ndt_time_chain(ndt_handle* h, const double T[16], int reps, double* us_per_launch);

/* The same for the batched residual kernel, on the batch left resident by the last ndt_align_batch (poses as
 * that call left them).  points_per_launch / hits_per_point may be NULL. */
int ndt_time_residual_batch(ndt_handle* h, int reps, double* us_per_launch, double* points_per_launch,
                            double* hits_per_point);

/* Radix-sort passes (8-bit digits) the last map build / merge executed: the P of SURVEY.md §8(d)'s
 * B_build = 12 + 12 + 24 P + 12 bytes per target point (measurement only). */
int ndt_last_radix_passes(const ndt_handle* h);

/* Blocks until all work queued on the handle's stream has finished. */
int ndt_synchronize(ndt_handle* h);

const char* ndt_last_error(const ndt_handle* h);
const char* ndt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NDT_MI355_H */
