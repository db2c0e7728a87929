// Walks the C ABI entry points that the other C++ drivers do not reach — voxel capacity with eviction and a table
// rebuild, stream-ordered source hand-over, ndt_align_static, downsampled source, save / load, p2p with one rank, group —
// on small random inputs, checking return codes and a few invariants.  Its job is to run under the host-side
// AddressSanitizer build (tools/build_asan.sh); results are checked against the oracle elsewhere (tests/test_gpu_*.py).
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "ndt_mi355.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != NDT_OK) { std::printf("FAIL %s -> %d (%s)\n", #x, rc_, h ? ndt_last_error(h) : ""); return 1; } } while (0)
#define REQUIRE(c) do { if (!(c)) { std::printf("FAIL %s (line %d)\n", #c, __LINE__); return 1; } } while (0)
#define HIPOK(x) do { if ((x) != hipSuccess) { std::printf("FAIL %s\n", #x); return 1; } } while (0)

static std::vector<float> box_cloud(std::mt19937& g, size_t n, float ox, float sx, float sy, float sz) {
    std::uniform_real_distribution<float> u(0.f, 1.f);
    std::vector<float> p(3 * n);
    for (size_t i = 0; i < n; ++i) { p[3 * i] = ox + sx * u(g); p[3 * i + 1] = sy * u(g); p[3 * i + 2] = sz * u(g); }
    return p;
}
static void identity(double T[16]) { for (int i = 0; i < 16; ++i) T[i] = (i % 5 == 0) ? 1.0 : 0.0; }

int main() {
    std::mt19937 gen(7);
    ndt_config cfg;
    ndt_default_config(&cfg);
    cfg.min_pts_in_voxel = 3; cfg.max_iterations = 6; cfg.eps = 0.0; cfg.capacity = 300;
    ndt_handle* h = ndt_create(&cfg);
    REQUIRE(h != nullptr);
    double I[16]; identity(I);

    // ---- capacity: 130 clouds, a whole 12 m box further each time (eviction every merge, one table rebuild on the way)
    for (int c = 0; c < 130; ++c) {
        std::vector<float> p = box_cloud(gen, 1500, 12.f * c, 12.f, 6.f, 2.f);
        if (c == 0) CHECK(ndt_set_target(h, p.data(), 1500, 12)); else CHECK(ndt_add_target(h, p.data(), 1500, 12, c % 2 ? I : nullptr));
        REQUIRE(ndt_num_voxels(h) <= 299);
    }
    REQUIRE(ndt_num_voxels(h) == 299);
    size_t nv = (size_t)ndt_num_voxels(h);
    std::vector<int64_t> keys(nv); std::vector<int32_t> counts(nv); std::vector<double> mean(nv * 3), cov(nv * 9), info(nv * 9); std::vector<uint8_t> valid(nv);
    CHECK(ndt_export_voxels(h, keys.data(), counts.data(), mean.data(), cov.data(), info.data(), valid.data(), nv, &nv));
    for (size_t i = 1; i < nv; ++i) REQUIRE(keys[i - 1] < keys[i]);

    // ---- save / load under the capacity, then one more merge
    size_t nb = 0;
    CHECK(ndt_map_save(h, nullptr, 0, &nb));
    std::vector<unsigned char> blob(nb);
    CHECK(ndt_map_save(h, blob.data(), nb, &nb));
    CHECK(ndt_map_load(h, blob.data(), nb));
    REQUIRE(ndt_num_voxels(h) == 299);
    std::vector<float> last = box_cloud(gen, 3000, 12.f * 129, 12.f, 6.f, 2.f);
    CHECK(ndt_add_target(h, last.data(), 3000, 12, nullptr));
    REQUIRE(ndt_num_voxels(h) == 299);

    // ---- source: host records (stride 16), stream-ordered device records, downsampled; align, static align, eval, cells
    const size_t ns = 2500;
    std::vector<float> rec(4 * ns);
    { std::vector<float> s = box_cloud(gen, ns, 12.f * 129 + 0.02f, 12.f, 6.f, 2.f); for (size_t i = 0; i < ns; ++i) { rec[4 * i] = s[3 * i]; rec[4 * i + 1] = s[3 * i + 1]; rec[4 * i + 2] = s[3 * i + 2]; rec[4 * i + 3] = 1.f; } }
    rec[4 * 7] = NAN;                                                   // one non-finite point
    ndt_result r0, r1, r2;
    CHECK(ndt_set_source(h, rec.data(), ns, 16));
    CHECK(ndt_align(h, I, &r0));
    float* d_rec = nullptr;
    HIPOK(hipMalloc((void**)&d_rec, rec.size() * 4));
    HIPOK(hipMemcpy(d_rec, rec.data(), rec.size() * 4, hipMemcpyHostToDevice));
    CHECK(ndt_set_source_async(h, d_rec, ns, 16));
    CHECK(ndt_align(h, I, &r1));
    for (int i = 0; i < 16; ++i) REQUIRE(r0.T[i] == r1.T[i]);          // same bits either way
    CHECK(ndt_set_source_async(h, d_rec, ns, 16));                      // pending records consumed by other entry points
    double H[36], b[6], chi2; int64_t neff;
    CHECK(ndt_eval(h, I, H, b, &chi2, &neff));
    std::vector<int64_t> ck(ns); std::vector<uint8_t> cm(ns);
    CHECK(ndt_source_cells(h, I, ck.data(), cm.data(), ns));
    size_t got = 0;
    std::vector<float> back(3 * ns);
    CHECK(ndt_get_source(h, back.data(), ns, &got));
    REQUIRE((int64_t)got == ndt_source_size(h));
    std::vector<float> sx(ns), sy(ns), sz(ns);
    for (size_t i = 0; i < ns; ++i) { sx[i] = rec[4 * i]; sy[i] = rec[4 * i + 1]; sz[i] = rec[4 * i + 2]; }
    float *dx = nullptr, *dy = nullptr, *dz = nullptr;
    HIPOK(hipMalloc((void**)&dx, ns * 4)); HIPOK(hipMalloc((void**)&dy, ns * 4)); HIPOK(hipMalloc((void**)&dz, ns * 4));
    HIPOK(hipMemcpy(dx, sx.data(), ns * 4, hipMemcpyHostToDevice)); HIPOK(hipMemcpy(dy, sy.data(), ns * 4, hipMemcpyHostToDevice));
    HIPOK(hipMemcpy(dz, sz.data(), ns * 4, hipMemcpyHostToDevice));
    CHECK(ndt_align_static(h, dx, dy, dz, ns, I, &r2));
    REQUIRE(r2.iterations == r0.iterations);
    CHECK(ndt_align_static(h, dx, dy, dz, ns / 3, I, &r2));
    CHECK(ndt_set_source_downsampled(h, rec.data(), ns, 16, 0.5f, I));
    CHECK(ndt_align(h, I, &r2));
    CHECK(ndt_add_target_from_source(h, I));
    REQUIRE(ndt_num_voxels(h) == 299);

    // ---- peer exchange with one rank, then plain again
    ndt_result r3;
    CHECK(ndt_set_source(h, rec.data(), ns, 16));
    CHECK(ndt_align(h, I, &r3));
    unsigned char hd[NDT_P2P_HANDLE_BYTES];
    CHECK(ndt_comm_p2p_export(h, hd));
    CHECK(ndt_comm_p2p_init(h, 0, 1, hd));
    REQUIRE(ndt_comm_size(h) == 1);
    CHECK(ndt_align(h, I, &r2));
    for (int i = 0; i < 16; ++i) REQUIRE(std::fabs(r2.T[i] - r3.T[i]) < 1e-9);          // one rank: the exchange adds nothing
    CHECK(ndt_comm_destroy(h));

    // ---- batched aligns on the handle and on a two-member group that shares device 0
    std::vector<float> scans; std::vector<int64_t> off{0};
    for (int s = 0; s < 3; ++s) { std::vector<float> p = box_cloud(gen, 800 + 100 * s, 12.f * 129, 12.f, 6.f, 2.f); scans.insert(scans.end(), p.begin(), p.end()); off.push_back(off.back() + 800 + 100 * s); }
    std::vector<double> guesses(16 * 5);
    for (int p = 0; p < 5; ++p) identity(&guesses[16 * p]);
    const int32_t pair_scan[5] = {0, 1, 2, 1, 0};
    std::vector<ndt_result> res(5);
    CHECK(ndt_align_batch(h, scans.data(), 12, off.data(), 3, pair_scan, 5, guesses.data(), res.data()));
    for (int i = 0; i < 16; ++i) REQUIRE(res[0].T[i] == res[4].T[i]);
    ndt_destroy(h); h = nullptr;
    cfg.capacity = 0;
    const int devs[2] = {0, 0};
    ndt_group* g = ndt_create_group(&cfg, 2, devs);
    REQUIRE(g != nullptr);
    std::vector<float> m = box_cloud(gen, 20000, 12.f * 129, 12.f, 6.f, 2.f);
    if (ndt_group_set_target(g, m.data(), 20000, 12) != NDT_OK) { std::printf("FAIL group target: %s\n", ndt_group_last_error(g)); return 1; }
    if (ndt_group_align_batch(g, scans.data(), 12, off.data(), 3, pair_scan, 5, guesses.data(), res.data()) != NDT_OK) { std::printf("FAIL group batch: %s\n", ndt_group_last_error(g)); return 1; }
    ndt_destroy_group(g);
    (void)hipFree(d_rec); (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(dz);
    std::printf("abi walk ok\n");
    std::fflush(stdout);
    _exit(0);          // (under ASan the HIP runtime's own exit handlers trip a sanitizer-internal check; nothing of ours is left to run)
}
