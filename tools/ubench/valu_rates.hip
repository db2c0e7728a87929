// Issue cost of the vector instructions the residual kernels are made of, on one CU: N waves per SIMD each run a
// stream of 8 independent instructions x 2048 rounds; cycles per instruction per SIMD = elapsed / (rounds * 8 * waves).
// Diagnostic only (tools/): hipcc --offload-arch=gfx950 -O2 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define ROUNDS 2048
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define OP8_D_F(op)  /* f64 dst <- f32 src */                                                                   \
    asm volatile(op " %0, %8\n" op " %1, %9\n" op " %2, %10\n" op " %3, %11\n" op " %4, %8\n" op " %5, %9\n"   \
                 op " %6, %10\n" op " %7, %11\n"                                                                \
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7)               \
                 : "v"(f0), "v"(f1), "v"(f2), "v"(f3))
#define OP8_D_DD(op) /* f64 dst <- f64, f64 */                                                                  \
    asm volatile(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" \
                 op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"                                       \
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(e0))
#define OP8_D_DDD(op)                                                                                           \
    asm volatile(op " %0, %8, %9, %0\n" op " %1, %8, %9, %1\n" op " %2, %8, %9, %2\n" op " %3, %8, %9, %3\n"    \
                 op " %4, %8, %9, %4\n" op " %5, %8, %9, %5\n" op " %6, %8, %9, %6\n" op " %7, %8, %9, %7\n"    \
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(e0), "v"(e1))
#define OP8_D_D(op)                                                                                             \
    asm volatile(op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n"      \
                 op " %6, %6\n" op " %7, %7\n"                                                                  \
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7))
#define OP8_I_II(op)                                                                                            \
    asm volatile(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" \
                 op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"                                       \
                 : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(j0))
#define OP8_F_FF(op)                                                                                            \
    asm volatile(op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" \
                 op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"                                       \
                 : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3), "+v"(g4), "+v"(g5), "+v"(g6), "+v"(g7) : "v"(f0))
#define OP8_D_I(op)                                                                                             \
    asm volatile(op " %0, %8\n" op " %1, %9\n" op " %2, %8\n" op " %3, %9\n" op " %4, %8\n" op " %5, %9\n"      \
                 op " %6, %8\n" op " %7, %9\n"                                                                  \
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(i0), "v"(i1))
#define OP8_I_D(op)                                                                                             \
    asm volatile(op " %0, %8\n" op " %1, %9\n" op " %2, %8\n" op " %3, %9\n" op " %4, %8\n" op " %5, %9\n"      \
                 op " %6, %8\n" op " %7, %9\n"                                                                  \
                 : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(d0), "v"(d1))
#define OP8_CMP_D(op)  /* writes vcc */                                                                         \
    asm volatile(op " vcc, %0, %1\n" op " vcc, %1, %2\n" op " vcc, %2, %3\n" op " vcc, %3, %4\n"                \
                 op " vcc, %4, %5\n" op " vcc, %5, %6\n" op " vcc, %6, %7\n" op " vcc, %7, %0\n"                \
                 :: "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(d4), "v"(d5), "v"(d6), "v"(d7) : "vcc")
#define OP8_CND(op)                                                                                             \
    asm volatile(op " %0, %0, %8, vcc\n" op " %1, %1, %8, vcc\n" op " %2, %2, %8, vcc\n" op " %3, %3, %8, vcc\n"\
                 op " %4, %4, %8, vcc\n" op " %5, %5, %8, vcc\n" op " %6, %6, %8, vcc\n" op " %7, %7, %8, vcc\n"\
                 : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7) : "v"(j0) : "vcc")
#define OP8_U64(op)  /* v_lshl_add_u64 d, d, 3, e */                                                            \
    asm volatile(op " %0, %0, 3, %8\n" op " %1, %1, 3, %8\n" op " %2, %2, 3, %8\n" op " %3, %3, 3, %8\n"        \
                 op " %4, %4, 3, %8\n" op " %5, %5, 3, %8\n" op " %6, %6, 3, %8\n" op " %7, %7, 3, %8\n"        \
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(e0))

template <int WHICH>
__global__ void k(long long* out, double seed, float fseed, int iseed) {
    double d0 = seed, d1 = seed + 1, d2 = seed + 2, d3 = seed + 3, d4 = seed + 4, d5 = seed + 5, d6 = seed + 6, d7 = seed + 7;
    double e0 = seed * 0.5 + 1e-9, e1 = 1.0000001;
    float f0 = fseed, f1 = fseed + 1, f2 = fseed + 2, f3 = fseed + 3;
    float g0 = fseed, g1 = f1, g2 = f2, g3 = f3, g4 = f0 + 4, g5 = f0 + 5, g6 = f0 + 6, g7 = f0 + 7;
    int i0 = iseed, i1 = iseed + 1, i2 = iseed + 2, i3 = iseed + 3, i4 = iseed + 4, i5 = iseed + 5, i6 = iseed + 6, i7 = iseed + 7, j0 = iseed | 1;
    __syncthreads();
    const long long w0 = wall_clock64();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < ROUNDS; ++r) {
        if constexpr (WHICH == 0) OP8_D_DD("v_add_f64");
        if constexpr (WHICH == 1) OP8_D_DD("v_mul_f64");
        if constexpr (WHICH == 2) OP8_D_DDD("v_fma_f64");
        if constexpr (WHICH == 3) OP8_D_F("v_cvt_f64_f32");
        if constexpr (WHICH == 4) OP8_D_I("v_cvt_f64_i32");
        if constexpr (WHICH == 5) OP8_I_D("v_cvt_i32_f64");
        if constexpr (WHICH == 6) OP8_D_D("v_floor_f64");
        if constexpr (WHICH == 7) OP8_CMP_D("v_cmp_ge_f64");
        if constexpr (WHICH == 8) OP8_CND("v_cndmask_b32");
        if constexpr (WHICH == 9) OP8_I_II("v_add_u32");
        if constexpr (WHICH == 10) OP8_U64("v_lshl_add_u64");
        if constexpr (WHICH == 11) OP8_F_FF("v_add_f32");
        if constexpr (WHICH == 12) OP8_I_II("v_mul_lo_u32");
        if constexpr (WHICH == 13) OP8_D_D("v_rcp_f64");
        if constexpr (WHICH == 14) OP8_I_II("v_and_b32");
        if constexpr (WHICH == 15) OP8_D_D("v_mov_b64");
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    const long long w1 = wall_clock64();
    __syncthreads();
    double s = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + (double)(g0 + g1 + g2 + g3 + g4 + g5 + g6 + g7) + (double)(i0 ^ i1 ^ i2 ^ i3 ^ i4 ^ i5 ^ i6 ^ i7);
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)s; out[2] = w1 - w0; }
}

template <int W> void run(const char* name, long long* d, double ticks_to_cycles) {
    for (int waves = 1; waves <= 2; ++waves) {
        long long h[3], best = 1ll << 60, bestw = 1ll << 60;
        for (int rep = 0; rep < 5; ++rep) {
            k<W><<<1, 256 * waves>>>(d, 1.5, 2.5f, 7);
            CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
            if (h[0] < best) best = h[0];
            if (h[2] < bestw) bestw = h[2];
        }
        printf("%-16s %d wave(s)/SIMD: memtime %8lld  wall(100MHz) %8lld -> %6.2f cycles per instruction per SIMD at 2.4 GHz\n", name, waves,
               best, bestw, bestw * ticks_to_cycles / (ROUNDS * 8.0 * waves));
    }
}

int main() {
    long long* d; CK(hipMalloc(&d, 64));
    // s_memtime ticks at 100 MHz on this part; the shader clock is taken as 2.4 GHz
    const double t2c = 24.0;
    run<0>("v_add_f64", d, t2c); run<1>("v_mul_f64", d, t2c); run<2>("v_fma_f64", d, t2c); run<3>("v_cvt_f64_f32", d, t2c);
    run<4>("v_cvt_f64_i32", d, t2c); run<5>("v_cvt_i32_f64", d, t2c); run<6>("v_floor_f64", d, t2c); run<7>("v_cmp_ge_f64", d, t2c);
    run<8>("v_cndmask_b32", d, t2c); run<9>("v_add_u32", d, t2c); run<10>("v_lshl_add_u64", d, t2c); run<11>("v_add_f32", d, t2c);
    run<12>("v_mul_lo_u32", d, t2c); run<13>("v_rcp_f64", d, t2c); run<14>("v_and_b32", d, t2c); run<15>("v_mov_b64", d, t2c);
    CK(hipFree(d));
    return 0;
}
