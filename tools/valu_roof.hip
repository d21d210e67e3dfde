// Microbenchmark: how many wave64 vector instructions a SIMD of the CU issues per second, for streams of independent
// v_fma_f32 / v_cndmask_b32 / v_cvt_f32_u32 / v_min_f32 ... at 1 .. 8 waves per SIMD, and of what the shading kernel adds to
// the mix (double precision, the pieces of an IEEE division, 64-bit address arithmetic) at its four waves per SIMD.  The
// kernels are priced against these rates (bench.py, roofline.valu).
// Build and run:  hipcc -O3 --offload-arch=gfx950 -o build/valu_roof tools/valu_roof.hip && build/valu_roof
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                  \
    do {                                                                                          \
        hipError_t e = (x);                                                                       \
        if (e != hipSuccess) {                                                                    \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                           \
            std::exit(1);                                                                         \
        }                                                                                         \
    } while (0)

constexpr int kUnroll = 16;  // independent chains per thread

template <int MODE>
__global__ __launch_bounds__(256) void k_valu(float* out, int iters, float a, float b) {
    float v[kUnroll];
    double w[kUnroll];
    const double da = double(a), db = double(b);
    unsigned long long mask = 0x5555555555555555ull, m2 = 0;
    asm volatile("" : "+s"(mask));
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) {
        v[k] = float(threadIdx.x + k);
        w[k] = double(threadIdx.x + k);
    }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < kUnroll; ++k) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
            if (MODE == 1) asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[k]) : "v"(a));
            if (MODE == 2) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(v[k]));
            if (MODE == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[k]) : "v"(a));
            if (MODE == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(reinterpret_cast<double*>(&v[k & ~1]))) : "v"(*(reinterpret_cast<double*>(&v[(k & ~1) ^ 2]))));
            if (MODE == 5) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "s"(mask));
            if (MODE == 6) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m2) : "v"(v[k]), "v"(a));
            if (MODE == 7) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
            if (MODE == 8) asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(v[k]));
            if (MODE == 9) asm volatile("v_mov_b32 %0, %1" : "=v"(v[k]) : "v"(a));
            if (MODE == 10) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[k]) : "v"(a));
            if (MODE == 11) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[k]) : "v"(a));
            if (MODE == 12) asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(v[k]) : "v"(a));
            if (MODE == 13) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
            if (MODE == 14) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[k]) : "v"(a));
            if (MODE == 15) asm volatile("v_lshrrev_b32 %0, 16, %0" : "+v"(v[k]));
            if (MODE == 16) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(v[k]));
            if (MODE == 17) asm volatile("v_min_u32 %0, %0, %1" : "+v"(v[k]) : "v"(a));
            if (MODE == 18) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
            if (MODE == 19) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
            if (MODE == 20) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(v[k]));
            if (MODE == 21) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(v[k]) : "v"(a), "v"(b));
            if (MODE == 22) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[k]) : "v"(a));
            if (MODE == 23) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[k]) : "v"(a));
            if (MODE == 24) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
            if (MODE == 25) asm volatile("v_lshl_or_b32 %0, %0, 9, %1" : "+v"(v[k]) : "v"(a));
            if (MODE == 26) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[k]) : "v"(a));
            if (MODE == 27) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
            if (MODE == 28) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(v[k]), "v"(a) : "vcc");
            if (MODE == 29) asm volatile("v_fma_f32 %0, |%0|, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
            if (MODE == 30) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "s"(a), "v"(b));
            // what the shading kernel adds to the mix: double precision, the pieces of an IEEE division, 64-bit addresses
            if (MODE == 40) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(w[k]) : "v"(da), "v"(db));
            if (MODE == 41) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(w[k]) : "v"(da));
            if (MODE == 42) asm volatile("v_add_f64 %0, %0, %1" : "+v"(w[k]) : "v"(da));
            if (MODE == 43) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[k]));
            if (MODE == 44) asm volatile("v_sqrt_f32 %0, %0" : "+v"(v[k]));
            if (MODE == 45) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(v[k]) : "v"(a) : "vcc");
            if (MODE == 46) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b) : "vcc");
            if (MODE == 47) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
            if (MODE == 48) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(w[k]) : "v"(da));
            if (MODE == 49) asm volatile("v_rcp_f64 %0, %0" : "+v"(w[k]));
            if (MODE == 50) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(w[k]) : "v"(v[k]));
            if (MODE == 51) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(v[k]) : "v"(w[k]));
            if (MODE == 52) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(w[k]) : "v"(da) : "vcc");
            if (MODE == 53) asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(w[k]) : "v"(da), "v"(db) : "vcc");
            if (MODE == 54) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(w[k]) : "v"(da), "v"(db));
            if (MODE == 55) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(v[k]) : "v"(a));
            if (MODE == 56) asm volatile("v_cmp_class_f32 vcc, %0, %1" : : "v"(v[k]), "v"(a) : "vcc");
            if (MODE == 57) asm volatile("v_rsq_f32 %0, %0" : "+v"(v[k]));
            if (MODE == 58) asm volatile("v_frexp_mant_f64 %0, %0" : "+v"(w[k]));
            if (MODE == 59) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(w[k]) : "v"(a));
            if (MODE == 60) asm volatile("v_rndne_f64 %0, %0" : "+v"(w[k]));
            // a select after the comparison that feeds it (counted as two instructions), condition in vcc or in another pair
            if (MODE == 61) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[k]) : "v"(a) : "vcc");
            if (MODE == 62) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(v[k]) : "v"(a));
            if (MODE == 63) asm volatile("v_cmp_lt_f32_e64 %1, %0, %2\n\tv_cndmask_b32_e64 %0, %0, %2, %1" : "+v"(v[k]), "=&s"(m2) : "v"(a));
            if (MODE == 64) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(v[k]) : "v"(a) : "vcc");
        }
    }
    float s = float(m2 & 1ull);
#pragma unroll
    for (int k = 0; k < kUnroll; ++k) s += v[k] + float(w[k]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int waves_per_simd, float* d_out, int cus) {
    // one workgroup of 256 threads = 4 waves = one wave per SIMD of a CU; waves_per_simd workgroups per CU
    const int blocks = cus * waves_per_simd, iters = 32768;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float ms = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {  // the first launch also brings the clocks up
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_valu<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0001f, 0.5f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float t = 0;
        CHECK(hipEventElapsedTime(&t, e0, e1));
        if (rep > 0 && t < ms) ms = t;
    }
    const double inst_per_simd = double(iters) * kUnroll * waves_per_simd * (MODE == 4 ? 0.5 : (MODE == 61 || MODE == 63 || MODE == 64) ? 2.0 : 1.0);
    std::printf("{\"instruction\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"wave_instructions_per_simd_per_us\": %.1f}\n", name, waves_per_simd, ms,
                inst_per_simd / (ms * 1e3));
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    float* d_out = nullptr;
    CHECK(hipMalloc(&d_out, size_t(cus) * 8 * 256 * sizeof(float)));
    std::printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", p.name, cus, p.clockRate / 1000);
    run<0>("warm-up", 8, d_out, cus);
    run<0>("warm-up", 8, d_out, cus);
    for (int w : {6}) {
        run<12>("v_alignbit_b32", w, d_out, cus);
        run<13>("v_perm_b32", w, d_out, cus);
        run<14>("v_and_b32", w, d_out, cus);
        run<15>("v_lshrrev_b32", w, d_out, cus);
        run<16>("v_bfe_u32", w, d_out, cus);
        run<17>("v_min_u32", w, d_out, cus);
        run<18>("v_mad_u32_u24", w, d_out, cus);
        run<19>("v_med3_f32", w, d_out, cus);
        run<20>("v_cvt_f32_ubyte1", w, d_out, cus);
        run<21>("v_fma_mix_f32", w, d_out, cus);
        run<22>("v_add_f32", w, d_out, cus);
        run<23>("v_max_f32", w, d_out, cus);
        run<24>("v_bfi_b32", w, d_out, cus);
        run<25>("v_lshl_or_b32", w, d_out, cus);
        run<26>("v_xor_b32", w, d_out, cus);
        run<27>("v_fmac_f32", w, d_out, cus);
        run<28>("v_cmp_lt_f32 (vcc)", w, d_out, cus);
        run<29>("v_fma_f32 |src|", w, d_out, cus);
        run<30>("v_fma_f32 sgpr src", w, d_out, cus);
    }
    for (int w : {4}) {  // the shading kernel runs four waves per SIMD
        run<0>("v_fma_f32", w, d_out, cus);
        run<40>("v_fma_f64", w, d_out, cus);
        run<41>("v_mul_f64", w, d_out, cus);
        run<42>("v_add_f64", w, d_out, cus);
        run<43>("v_rcp_f32", w, d_out, cus);
        run<44>("v_sqrt_f32", w, d_out, cus);
        run<57>("v_rsq_f32", w, d_out, cus);
        run<45>("v_div_scale_f32", w, d_out, cus);
        run<46>("v_div_fmas_f32", w, d_out, cus);
        run<47>("v_div_fixup_f32", w, d_out, cus);
        run<48>("v_lshl_add_u64", w, d_out, cus);
        run<49>("v_rcp_f64", w, d_out, cus);
        run<50>("v_cvt_f64_f32", w, d_out, cus);
        run<51>("v_cvt_f32_f64", w, d_out, cus);
        run<52>("v_div_scale_f64", w, d_out, cus);
        run<53>("v_div_fmas_f64", w, d_out, cus);
        run<54>("v_div_fixup_f64", w, d_out, cus);
        run<55>("v_ldexp_f32", w, d_out, cus);
        run<56>("v_cmp_class_f32 (vcc)", w, d_out, cus);
        run<58>("v_frexp_mant_f64", w, d_out, cus);
        run<59>("v_ldexp_f64", w, d_out, cus);
        run<60>("v_rndne_f64", w, d_out, cus);
        run<61>("v_cmp_lt_f32 vcc + v_cndmask_b32 (vcc)", w, d_out, cus);
        run<64>("v_cmp_lt_f32 vcc + v_cndmask_b32_e64 (vcc)", w, d_out, cus);
        run<63>("v_cmp_lt_f32_e64 + v_cndmask_b32_e64 (sgpr pair)", w, d_out, cus);
        run<62>("v_cndmask_b32_e64 (vcc)", w, d_out, cus);
        run<9>("v_mov_b32", w, d_out, cus);
        run<11>("v_mul_f32", w, d_out, cus);
        run<22>("v_add_f32", w, d_out, cus);
        run<5>("v_cndmask_b32_e64 (sgpr pair)", w, d_out, cus);
    }
    for (int w : {1, 2, 4, 6, 8}) {
        run<0>("v_fma_f32", w, d_out, cus);
        run<11>("v_mul_f32", w, d_out, cus);
        run<1>("v_min_f32", w, d_out, cus);
        run<7>("v_max3_f32", w, d_out, cus);
        run<2>("v_cvt_f32_u32", w, d_out, cus);
        run<8>("v_cvt_f32_u32_sdwa", w, d_out, cus);
        run<3>("v_cndmask_b32 (vcc)", w, d_out, cus);
        run<5>("v_cndmask_b32_e64 (sgpr pair)", w, d_out, cus);
        run<6>("v_cmp_lt_f32_e64", w, d_out, cus);
        run<9>("v_mov_b32", w, d_out, cus);
        run<10>("v_add_u32", w, d_out, cus);
        run<4>("v_pk_fma_f32", w, d_out, cus);
    }
    return 0;
}
