// ndt_group.hip — several GPUs in one process: one member handle per device, calls fanned out on host threads.
// Host code only, written against the public C ABI (no kernels here): a group adds no device state of its own.
#include "../../include/ndt_mi355.h"

#include <string.h>

#include <algorithm>
#include <new>
#include <string>
#include <thread>
#include <vector>

struct ndt_group {
    std::vector<ndt_handle*> members;
    std::string err;
};

namespace {
// runs f(member index) for every member, each on its own host thread (a handle is only ever touched by one thread at
// a time); returns the first non-zero status in member order and keeps that member's message
template <class F>
int fan_out(ndt_group* g, F f) {
    const int n = (int)g->members.size();
    std::vector<int> rc(n, NDT_OK);
    if (n == 1) rc[0] = f(0);
    else {
        std::vector<std::thread> th;
        th.reserve(n);
        for (int m = 0; m < n; ++m) th.emplace_back([&, m] { rc[m] = f(m); });
        for (auto& t : th) t.join();
    }
    for (int m = 0; m < n; ++m)
        if (rc[m] != NDT_OK) { g->err = "member " + std::to_string(m) + ": " + ndt_last_error(g->members[m]); return rc[m]; }
    return NDT_OK;
}
}  // namespace

extern "C" {

ndt_group* ndt_create_group(const ndt_config* cfg, int n_dev, const int* dev_ids) {
    if (!cfg || n_dev < 1 || n_dev > 64 || !dev_ids) return nullptr;
    ndt_group* g = new (std::nothrow) ndt_group();
    if (!g) return nullptr;
    for (int m = 0; m < n_dev; ++m) {
        ndt_config c = *cfg;
        c.device = dev_ids[m];
        ndt_handle* h = ndt_create(&c);
        if (!h) { ndt_destroy_group(g); return nullptr; }
        g->members.push_back(h);
    }
    return g;
}

void ndt_destroy_group(ndt_group* g) {
    if (!g) return;
    for (ndt_handle* h : g->members) ndt_destroy(h);
    delete g;
}

int ndt_group_size(const ndt_group* g) { return g ? (int)g->members.size() : 0; }
ndt_handle* ndt_group_handle(ndt_group* g, int member) {
    return (g && member >= 0 && member < (int)g->members.size()) ? g->members[member] : nullptr;
}
const char* ndt_group_last_error(const ndt_group* g) { return g ? g->err.c_str() : "null group"; }

int ndt_group_set_target(ndt_group* g, const void* pts, size_t n, size_t stride_bytes) {
    if (!g) return NDT_ERR_INVALID_ARG;
    return fan_out(g, [&](int m) { return ndt_set_target(g->members[m], pts, n, stride_bytes); });
}
int ndt_group_add_target(ndt_group* g, const void* pts, size_t n, size_t stride_bytes, const double T[16]) {
    if (!g) return NDT_ERR_INVALID_ARG;
    return fan_out(g, [&](int m) { return ndt_add_target(g->members[m], pts, n, stride_bytes, T); });
}

int ndt_group_align_batch(ndt_group* g, const void* pts, size_t stride_bytes, const int64_t* scan_offsets, size_t n_scans,
                          const int32_t* pair_scan, size_t n_pairs, const double* guesses, ndt_result* results) {
    if (!g) return NDT_ERR_INVALID_ARG;
    if (!scan_offsets || !n_scans || !n_pairs || !guesses || !results || stride_bytes < 12 || (stride_bytes & 3)) { g->err = "bad argument"; return NDT_ERR_INVALID_ARG; }
    if (!pair_scan && n_pairs != n_scans) { g->err = "without pair_scan there must be one scan per pair"; return NDT_ERR_INVALID_ARG; }
    if (scan_offsets[0] != 0) { g->err = "scan_offsets must start at 0"; return NDT_ERR_INVALID_ARG; }
    for (size_t i = 0; i < n_scans; ++i)
        if (scan_offsets[i + 1] < scan_offsets[i]) { g->err = "scan_offsets must not decrease"; return NDT_ERR_INVALID_ARG; }
    if (scan_offsets[n_scans] && !pts) { g->err = "null point buffer"; return NDT_ERR_INVALID_ARG; }
    if (pair_scan)
        for (size_t p = 0; p < n_pairs; ++p)
            if (pair_scan[p] < 0 || (size_t)pair_scan[p] >= n_scans) { g->err = "pair_scan out of range"; return NDT_ERR_INVALID_ARG; }
    const size_t n_mem = g->members.size();
    const size_t active = std::min(n_mem, n_pairs);              // members beyond the pair count sit this call out
    return fan_out(g, [&](int m) -> int {
        if ((size_t)m >= active) return NDT_OK;
        // contiguous share of the pairs (same cut as lio_slam_amd.shard.shard_range)
        const size_t lo = n_pairs * (size_t)m / active, hi = n_pairs * ((size_t)m + 1) / active;
        const size_t np = hi - lo;
        // the scans this share uses, in first-use order, gathered into one host buffer
        std::vector<int32_t> local_of(n_scans, -1), ps(np);
        std::vector<size_t> used;
        for (size_t p = 0; p < np; ++p) {
            const size_t sc = pair_scan ? (size_t)pair_scan[lo + p] : lo + p;
            if (local_of[sc] < 0) { local_of[sc] = (int32_t)used.size(); used.push_back(sc); }
            ps[p] = local_of[sc];
        }
        std::vector<int64_t> off(used.size() + 1, 0);
        for (size_t u = 0; u < used.size(); ++u) off[u + 1] = off[u] + (scan_offsets[used[u] + 1] - scan_offsets[used[u]]);
        std::vector<unsigned char> buf((size_t)off.back() * stride_bytes + 16);
        for (size_t u = 0; u < used.size(); ++u)
            memcpy(buf.data() + (size_t)off[u] * stride_bytes, (const unsigned char*)pts + (size_t)scan_offsets[used[u]] * stride_bytes,
                   (size_t)(off[u + 1] - off[u]) * stride_bytes);
        return ndt_align_batch(g->members[m], buf.data(), stride_bytes, off.data(), used.size(), ps.data(), np, guesses + 16 * lo, results + lo);
    });
}

}  // extern "C"
