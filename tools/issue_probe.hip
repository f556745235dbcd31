// issue_probe.hip — what ONE instruction costs a wave and its SIMD on gfx950 when it is not an independent VALU stream
// (tools/valu_peak.hip measures those): scalar ALU, dependent chains, branches, lane reads, waits, LDS and scalar-cache
// round trips, and VALU / SALU streams mixed in one wave.  Same method as valu_peak.hip: a persistent grid of CUs x W
// workgroups of 256 threads (one wave per SIMD each, LDS-sized so that exactly W are resident per CU), every wave repeating a
// block of 16 instructions until a 2 ms window of the 100 MHz clock has passed; cycles come from s_memtime around the window.
//   per wave  = window cycles / instructions the wave issued        (the wave's own issue-to-issue time)
//   per SIMD  = window cycles / instructions all W waves issued     (what the instruction occupies of the SIMD)
//   hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o tools/_build/issue_probe && tools/_build/issue_probe [csv]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum Kind {
    S_ADD_INDEP, S_ADD_DEP, S_MOV, V_FMA32_DEP, V_FMA64_DEP, V_CVT_DEP, V_SQRT_DEP, V_CMP_CNDMASK_DEP, MIX_V_S, MIX_V64_S, MIX_V_S_S,
    VOTE_BRANCH_NT, BRANCH_TAKEN, READLANE, READFIRSTLANE_DEP, WRITELANE, S_NOP, S_WAITCNT_IDLE, LDS_TRIP, LDS_TRIP_B128, SMEM_TRIP, SMEM_TRIP_X16,
    V_FMA32_INDEP, V_FMA64_INDEP, SAVEEXEC, NKIND
};
static const char* kind_name[NKIND] = {
    "s_add_u32 x16 independent", "s_add_u32 dependent", "s_mov_b32", "v_fma_f32 dependent", "v_fma_f64 dependent", "v_cvt f32<->f64 dependent",
    "v_sqrt_f32 dependent", "v_cmp+v_cndmask dependent (per pair)", "v_fma_f32 + s_add alternating (per pair)", "v_fma_f64 + s_add alternating (per pair)",
    "v_fma_f32 + 2 s_add (per triple)", "v_cmp + s_cbranch_vccz not taken (per pair)", "s_cbranch taken over one instr (per branch)", "v_readlane_b32 independent",
    "v_readfirstlane -> v_mov dependent (per pair)", "v_writelane_b32", "s_nop 0", "s_waitcnt lgkmcnt(0), nothing pending", "ds_read_b32 -> wait -> address (per trip)",
    "ds_read_b128 -> wait -> address (per trip)", "s_load_dword -> wait -> address (per trip)", "s_load_dwordx16 -> wait -> address (per trip)",
    "v_fma_f32 x16 independent", "v_fma_f64 x16 independent", "s_and_saveexec + s_or exec (per pair)"};
static const int kind_units[NKIND] = {16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 16};  // units per block

struct Stamp { unsigned long long c0, c1, r0, r1, iters; };

#define R16(x) x x x x x x x x x x x x x x x x
#define R8(x) x x x x x x x x

template <int K>
__global__ __launch_bounds__(256, 8) void k_probe(float* out, Stamp* stamps, unsigned long long window_ticks, float seed, const unsigned* table) {
    extern __shared__ char lds[];
    float a[16];
    double d[16];
    unsigned s[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; d[i] = seed * 0.5 + i; s[i] = __builtin_amdgcn_readfirstlane(i * 7 + (int)seed); }
    const float b = seed * 0.999f, c = seed * 0.001f;
    const double bd = b, cd = c;
    unsigned laddr = (threadIdx.x & 63) * 16;  // LDS byte address; the loaded word (0) is added to it: a dependent trip
    for (int i = threadIdx.x; i < 4096; i += 256) reinterpret_cast<unsigned*>(lds)[i] = 0u;
    __syncthreads();
    unsigned long long saddr = (unsigned long long)table;  // table[] is all zero: address += loaded word
    const unsigned long long r0 = wall_clock64();
    const unsigned long long c0 = __builtin_readcyclecounter();
    unsigned long long iters = 0, r1;
    do {
#pragma unroll 1
        for (int it = 0; it < 32; ++it) {
            if (K == S_ADD_INDEP) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s[i]) : : "scc");
            }
            if (K == S_ADD_DEP) asm volatile(R16("s_add_u32 %0, %0, 3\n\t") : "+s"(s[0]) : : "scc");
            if (K == S_MOV) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("s_mov_b32 %0, %1" : "=s"(s[i]) : "s"(s[(i + 1) & 15]));
            }
            if (K == V_FMA32_DEP) asm volatile(R16("v_fma_f32 %0, %0, %1, %2\n\t") : "+v"(a[0]) : "v"(b), "v"(c));
            if (K == V_FMA64_DEP) asm volatile(R16("v_fma_f64 %0, %0, %1, %2\n\t") : "+v"(d[0]) : "v"(bd), "v"(cd));
            if (K == V_CVT_DEP) asm volatile(R8("v_cvt_f64_f32 %1, %0\n\tv_cvt_f32_f64 %0, %1\n\t") : "+v"(a[0]), "+v"(d[0]));
            if (K == V_SQRT_DEP) asm volatile(R16("v_sqrt_f32 %0, %0\n\t") : "+v"(a[0]));
            if (K == V_CMP_CNDMASK_DEP) asm volatile(R16("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc\n\t") : "+v"(a[0]) : "v"(b), "v"(c) : "vcc");
            if (K == MIX_V_S) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %2, %3\n\ts_add_u32 %1, %1, 3" : "+v"(a[i]), "+s"(s[i]) : "v"(b), "v"(c) : "scc");
            }
            if (K == MIX_V64_S) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %2, %3\n\ts_add_u32 %1, %1, 3" : "+v"(d[i]), "+s"(s[i]) : "v"(bd), "v"(cd) : "scc");
            }
            if (K == MIX_V_S_S) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    asm volatile("v_fma_f32 %0, %0, %3, %4\n\ts_add_u32 %1, %1, 3\n\ts_add_u32 %2, %2, 5" : "+v"(a[i]), "+s"(s[i]), "+s"(s[(i + 8) & 15]) : "v"(b), "v"(c) : "scc");
            }
            if (K == VOTE_BRANCH_NT) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_cmp_gt_f32 vcc, %0, %0\n\ts_cbranch_vccnz 1f\n\t1:" : : "v"(a[i]) : "vcc");  // a > a is false everywhere: never taken
            }
            if (K == BRANCH_TAKEN) asm volatile(R16("s_branch 1f\n\ts_nop 0\n\t1:\n\t"));
            if (K == READLANE) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s[i]) : "v"(a[i]));
            }
            if (K == READFIRSTLANE_DEP) asm volatile(R16("v_readfirstlane_b32 %1, %0\n\tv_mov_b32 %0, %1\n\t") : "+v"(a[0]), "+s"(s[0]));
            if (K == WRITELANE) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_writelane_b32 %0, %1, 5" : "+v"(a[i]) : "s"(s[i]));
            }
            if (K == S_NOP) asm volatile(R16("s_nop 0\n\t"));
            if (K == S_WAITCNT_IDLE) asm volatile(R16("s_waitcnt lgkmcnt(0)\n\t"));
            if (K == LDS_TRIP) {
                unsigned w;
                asm volatile(R16("ds_read_b32 %1, %0\n\ts_waitcnt lgkmcnt(0)\n\tv_add_u32 %0, %0, %1\n\t") : "+v"(laddr), "=&v"(w));
            }
            if (K == LDS_TRIP_B128) {
                typedef unsigned u4 __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    u4 q;
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(laddr));
                    asm volatile("v_add_u32 %0, %0, %1" : "+v"(laddr) : "v"(q.x));
                }
            }
            if (K == SMEM_TRIP) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    unsigned w;
                    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(w) : "s"(saddr));
                    asm volatile("s_add_u32 %0, %0, %1" : "+s"(*reinterpret_cast<unsigned*>(&saddr)) : "s"(w) : "scc");
                }
            }
            if (K == SMEM_TRIP_X16) {
                typedef unsigned u16v __attribute__((ext_vector_type(16)));
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    u16v w;
                    asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(w) : "s"(saddr));
                    asm volatile("s_add_u32 %0, %0, %1" : "+s"(*reinterpret_cast<unsigned*>(&saddr)) : "s"(w.s5) : "scc");
                }
            }
            if (K == V_FMA32_INDEP) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            }
            if (K == V_FMA64_INDEP) {
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(bd), "v"(cd));
            }
            if (K == SAVEEXEC) {
                unsigned long long sv;
                asm volatile(R16("s_and_saveexec_b64 %0, exec\n\ts_or_b64 exec, exec, %0\n\t") : "=&s"(sv) : : "scc");
            }
        }
        iters += 32;
        r1 = wall_clock64();
    } while (r1 - r0 < window_ticks);
    const unsigned long long c1 = __builtin_readcyclecounter();
    float sum = (float)laddr + (float)(unsigned)saddr;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += a[i] + (float)d[i] + (float)s[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0) stamps[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{c0, c1, r0, r1, iters};
}

struct Result { double per_wave, per_simd, ghz; };

template <int K>
static Result run(int cus, int W, float* d_out, Stamp* d_st, const unsigned* d_table) {
    static const int lds_for[9] = {0, 163840, 80896, 53248, 40960, 31744, 26624, 22528, 19456};
    const size_t lds = (size_t)lds_for[W];
    const int grid = cus * W;
    if (lds > 64 * 1024) CHECK(hipFuncSetAttribute((const void*)k_probe<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_probe<K>, dim3(grid), dim3(256), lds, 0, d_out, d_st, 10000ull, 1.0001f, d_table);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_probe<K>, dim3(grid), dim3(256), lds, 0, d_out, d_st, 200000ull, 1.0001f, d_table);  // 2 ms
    CHECK(hipDeviceSynchronize());
    std::vector<Stamp> h((size_t)grid * 4);
    CHECK(hipMemcpy(h.data(), d_st, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    double cyc = 0, units = 0, clk = 0;
    for (const Stamp& s : h) {
        cyc += (double)(s.c1 - s.c0);
        units += (double)s.iters * kind_units[K];
        clk += (double)(s.c1 - s.c0) / (double)(s.r1 - s.r0);
    }
    Result r;
    r.per_wave = cyc / units;                      // mean over waves of (cycles / units of that wave), weighted by units
    r.per_simd = r.per_wave / W;                   // W waves share the SIMD for the same window
    r.ghz = clk / h.size() / 10.0;
    return r;
}

typedef Result (*RunFn)(int, int, float*, Stamp*, const unsigned*);
template <int K> static void fill(RunFn* t) { t[K] = run<K>; fill<K + 1>(t); }
template <> void fill<NKIND>(RunFn*) {}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    float* d_out; Stamp* d_st; unsigned* d_table;
    CHECK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * sizeof(float)));
    CHECK(hipMalloc(&d_st, (size_t)cus * 8 * 4 * sizeof(Stamp)));
    CHECK(hipMalloc(&d_table, 4096));
    CHECK(hipMemset(d_table, 0, 4096));
    RunFn table[NKIND];
    fill<0>(table);
    FILE* f = argc > 1 ? fopen(argv[1], "w") : nullptr;
    if (f) fprintf(f, "unit,waves_per_simd,cycles_per_unit_per_wave,cycles_per_unit_per_simd,shader_clock_ghz\n");
    printf("device %s, %d CUs; cycles per unit (s_memtime)\n%-48s %2s %10s %10s %6s\n", prop.gcnArchName, cus, "unit", "W", "per wave", "per SIMD", "GHz");
    const int Ws[3] = {1, 2, 5};
    for (int k = 0; k < NKIND; ++k)
        for (int W : Ws) {
            const Result r = table[k](cus, W, d_out, d_st, d_table);
            printf("%-48s %2d %10.2f %10.2f %6.2f\n", kind_name[k], W, r.per_wave, r.per_simd, r.ghz);
            if (f) fprintf(f, "\"%s\",%d,%.3f,%.3f,%.3f\n", kind_name[k], W, r.per_wave, r.per_simd, r.ghz);
        }
    if (f) fclose(f);
    return 0;
}
