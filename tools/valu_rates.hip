// valu_rates.hip — calibration of the "VALU issue" roofline on gfx950 (MI355X).
//
// Measures, per instruction kind, the cycles one SIMD needs per wave64 instruction when W waves per SIMD issue
// independent streams of it (s_memtime around the loop, one figure per wave, averaged).  The saturated figure of the
// plain VALU kinds is the denominator of bench.py's `roofline` (bound "valu-issue"): the dominant kernel of this repo
// is bound by VALU instruction issue, not by HBM or MFMA.  Run under rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
// SQ_BUSY_CYCLES GRBM_GUI_ACTIVE the saturated kernels also give the largest value the PMC-derived "VALU busy"
// fraction can reach (tools/roofline.py uses the same formula on the product kernel).
//
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rates.hip -o tools/_build/valu_rates && tools/_build/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

enum Kind { FMA_F32, PK_FMA_F32, FMA_F64, MUL_F64, ADD_F64, ADD_F32, CVT_F32_F64, CVT_F64_F32, CMP_CNDMASK, AND_B32, MAD_U32, RCP_F32, SQRT_F32, MIX_GOALSET, NKIND };
static const char* kind_name[NKIND] = {"v_fma_f32", "v_pk_fma_f32 (2 FMA/lane)", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_add_f32", "v_cvt_f32_f64",
                                       "v_cvt_f64_f32", "v_cmp+v_cndmask (2 instr)", "v_and_b32", "v_mad_u32_u24", "v_rcp_f32", "v_sqrt_f32",
                                       "mix 85% f32 / 15% f64 fma"};

#define ITER 512
#define UNROLL 8  // independent chains per lane

template <int K>
__global__ __launch_bounds__(256) void k_rate(float* out, unsigned long long* cyc, float seed) {
    float a[UNROLL];
    double d[UNROLL];
    float2v v[UNROLL];
    unsigned u[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) { a[i] = seed + i + threadIdx.x; d[i] = seed * 0.5 + i; v[i] = float2v{a[i], a[i] + 1.0f}; u[i] = threadIdx.x * 2654435761u + i; }
    const float b = seed * 0.999f, c = seed * 0.001f;
    const double bd = b, cd = c;
    const float2v b2 = float2v{b, b}, c2 = float2v{c, c};
    const unsigned m = 0xfffffff7u ^ (unsigned)seed;
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < UNROLL; ++i) {
            // inline asm: the compiler would otherwise pack neighbouring f32 chains into v_pk_* instructions
            if (K == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (K == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b2), "v"(c2));
            if (K == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(bd), "v"(cd));
            if (K == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(bd));
            if (K == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(cd));
            if (K == ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            if (K == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
            if (K == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
            if (K == CMP_CNDMASK) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
            if (K == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(m));
            if (K == MAD_U32) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[i]) : "v"(m), "v"(m));
            if (K == RCP_F32) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (K == SQRT_F32) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            if (K == MIX_GOALSET) {
                if (i == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(bd), "v"(cd));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) s += a[i] + (float)d[i] + v[i].x + v[i].y + (float)u[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int K>
static void run(int cus, int wgs_per_cu, float* d_out, unsigned long long* d_cyc) {
    const int grid = cus * wgs_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_rate<K>, dim3(grid), dim3(256), 0, 0, d_out, d_cyc, 1.0001f);  // warm-up
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_rate<K>, dim3(grid), dim3(256), 0, 0, d_out, d_cyc, 1.0001f);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)grid * 4);
    CHECK(hipMemcpy(h.data(), d_cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double sum = 0.0;
    for (auto x : h) sum += (double)x;
    const double per_wave = sum / h.size();                 // s_memtime ticks one wave spent in the loop
    const double instr = (double)ITER * UNROLL * (K == CMP_CNDMASK ? 2 : 1);
    // W waves share one SIMD (256-thread blocks put one wave on each of the 4 SIMDs): ticks per instruction per SIMD
    printf("%-28s W=%d  ticks/wave-instr %.2f  ticks/instr/SIMD %.2f  kernel %.1f us\n", kind_name[K], wgs_per_cu, per_wave / instr,
           per_wave / instr / wgs_per_cu, ms * 1e3);
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz (s_memtime ticks: see MI355X_MICROARCH.md, tick = shader cycle)\n", prop.gcnArchName, cus, prop.clockRate);
    float* d_out; unsigned long long* d_cyc;
    CHECK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * sizeof(float)));
    CHECK(hipMalloc(&d_cyc, (size_t)cus * 8 * 4 * sizeof(unsigned long long)));
    if (argc > 1) {  // "pmc": few dispatches for a rocprofv3 --pmc run (two launches per line: warm-up + timed)
        for (int w : {4, 6, 8}) { run<FMA_F32>(cus, w, d_out, d_cyc); run<FMA_F64>(cus, w, d_out, d_cyc); run<MIX_GOALSET>(cus, w, d_out, d_cyc); run<PK_FMA_F32>(cus, w, d_out, d_cyc); }
        return 0;
    }
    const int ws[] = {1, 2, 4, 6, 8};
    for (int w : ws) {
        run<FMA_F32>(cus, w, d_out, d_cyc);
        run<PK_FMA_F32>(cus, w, d_out, d_cyc);
        run<FMA_F64>(cus, w, d_out, d_cyc);
        run<MUL_F64>(cus, w, d_out, d_cyc);
        run<ADD_F64>(cus, w, d_out, d_cyc);
        run<ADD_F32>(cus, w, d_out, d_cyc);
        run<CVT_F32_F64>(cus, w, d_out, d_cyc);
        run<CVT_F64_F32>(cus, w, d_out, d_cyc);
        run<CMP_CNDMASK>(cus, w, d_out, d_cyc);
        run<AND_B32>(cus, w, d_out, d_cyc);
        run<MAD_U32>(cus, w, d_out, d_cyc);
        run<RCP_F32>(cus, w, d_out, d_cyc);
        run<SQRT_F32>(cus, w, d_out, d_cyc);
        run<MIX_GOALSET>(cus, w, d_out, d_cyc);
    }
    return 0;
}
