// valu_peak.hip — residency-verified calibration of the VALU issue rate of gfx950 (MI355X): the denominator of bench.py's
// `roofline` (bound "valu-issue").  Replaces tools/valu_rates.hip, whose 4096-instruction kernels were shorter than their own
// launch ramp.
//
// One persistent grid per measurement: CUs x W workgroups of 256 threads (one wave per SIMD each), each asking for the largest
// LDS size of which W fit a CU, so that exactly W workgroups = W waves per SIMD are resident on every CU and none waits for a
// slot.  The measurement is TIME-bounded, not count-bounded: every wave issues its instruction stream until a fixed window
// (3 ms of the 100 MHz clock) after its own start has passed and reports how many instructions it got through.  Count-bounded
// runs turned out useless on this chip: a SIMD serves its waves oldest-first, the favoured waves finish early and the rest of
// the run is a tail of under-occupied SIMDs (first version of this tool: per-wave durations of the same work 2.5-10 ms at
// W = 8).  With a window all waves are co-resident for the whole measurement by construction; the host still checks
//   * residency: every (XCC, SE, CU, SIMD) that ran anything ran exactly W waves,
//   * overlap  : (earliest end - latest start) / (latest end - earliest start) >= 0.95,
// and prints, per instruction kind and W,
//   cyc/instr/SIMD : window in shader cycles (s_memtime, per wave) / instructions the SIMD's W waves issued in it — the
//                    steady-state cost of one wave64 instruction to its SIMD; chip peak = 1024 SIMDs x clock / this;
//   min / max share: the least- and most-favoured wave's part of its SIMD's instructions (fair = 1 / W);
//   GHz            : shader cycles per 10 ns of the 100 MHz clock, i.e. the clock the CUs really ran at under this load;
//   G instr/s      : chip-wide wave-instructions per second actually retired in the window (all SIMDs).
// Under `rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU` (argument "pmc": fewer launches) the
// counters give the same ratio from the other side: SQ_INSTS_VALU / 1024 SIMDs against GRBM_GUI_ACTIVE cycles.
//
//   hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o tools/_build/valu_peak && tools/_build/valu_peak [pmc] [dump DIR] [csv-path]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

enum Kind {
    FMA_F32, ADD_F32, MUL_F32, PK_FMA_F32, FMA_F64, ADD_F64, MUL_F64, CVT_F32_F64, CVT_F64_F32, CVT_I32_F32, CMP_CNDMASK, AND_B32,
    MAD_U32, ADD_U32, LSHL_ADD, MOV_DPP, RCP_F32, SQRT_F32, MIX_GOALSET, NKIND
};
static const char* kind_name[NKIND] = {
    "v_fma_f32", "v_add_f32", "v_mul_f32", "v_pk_fma_f32", "v_fma_f64", "v_add_f64", "v_mul_f64", "v_cvt_f32_f64", "v_cvt_f64_f32",
    "v_cvt_i32_f32", "v_cmp_f32+v_cndmask", "v_and_b32", "v_mad_u32_u24", "v_add_u32", "v_lshl_add_u32", "v_mov_b32 dpp",
    "v_rcp_f32", "v_sqrt_f32", "mix goalset (see MIX)"};
static const int kind_instrs[NKIND] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 1, 1, 1};

#define UNROLL 16  // independent chains per lane: no instruction waits for its predecessor's result

struct Stamp { unsigned long long c0, c1, r0, r1, iters; unsigned hw, xcc; };

#define INNER 32  // loop bodies (16 instructions each) between two looks at the clock: ~1-2 us

// One instruction of kind K on chain i.  Only the register arrays a kind uses exist in its kernel (the others are never
// touched and fold away): every kernel stays below 64 VGPRs, so 8 waves per SIMD fit.
// MIX: the dynamic VALU mix of k_goalset_queue<2> by rocprofv3's SQ_INSTS_VALU_* counters (profiles/r03*_pmc_MIX.csv): of 16
// instructions about 9 plain f32 (fma / add / mul / compare / select), 3 int32, 2 f64, 1 conversion, 1 move.
template <int K>
__device__ __forceinline__ void body(float (&a)[UNROLL], double (&d)[4], float2v (&v)[UNROLL], unsigned (&u)[UNROLL], double (&dd)[UNROLL],
                                     const float b, const float c, const double bd, const double cd, const float2v b2, const float2v c2,
                                     const unsigned m) {
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
        // inline asm: the compiler would otherwise pack neighbouring f32 chains into v_pk_* instructions or fold them
        if (K == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (K == ADD_F32) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (K == MUL_F32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (K == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(b2), "v"(c2));
        if (K == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(dd[i]) : "v"(bd), "v"(cd));
        if (K == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(dd[i]) : "v"(cd));
        if (K == MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(dd[i]) : "v"(bd));
        if (K == CVT_F32_F64) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(dd[i]));
        if (K == CVT_F64_F32) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(dd[i]) : "v"(a[i]));
        if (K == CVT_I32_F32) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(u[i]) : "v"(a[i]));
        if (K == CMP_CNDMASK) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
        if (K == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[i]) : "v"(m));
        if (K == MAD_U32) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[i]) : "v"(m), "v"(m));
        if (K == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(m));
        if (K == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(m));
        if (K == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(u[i]));
        if (K == RCP_F32) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        if (K == SQRT_F32) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
        if (K == MIX_GOALSET) {
            if (i == 3 || i == 11) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i == 3 ? 0 : 1]) : "v"(bd), "v"(cd));
            else if (i == 7) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[2]));
            else if (i == 1 || i == 9 || i == 13) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[i]) : "v"(m), "v"(m));
            else if (i == 15) asm volatile("v_mov_b32 %0, %1" : "=v"(u[i]) : "v"(m));
            else if (i == 5) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
            else if (i == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(c) : "vcc");
            else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        }
    }
}

constexpr bool uses_a(int K) { return K == FMA_F32 || K == ADD_F32 || K == MUL_F32 || K == CVT_F32_F64 || K == CVT_F64_F32 || K == CVT_I32_F32 || K == CMP_CNDMASK || K == RCP_F32 || K == SQRT_F32 || K == MIX_GOALSET; }
constexpr bool uses_dd(int K) { return K == FMA_F64 || K == ADD_F64 || K == MUL_F64 || K == CVT_F32_F64 || K == CVT_F64_F32; }
constexpr bool uses_v(int K) { return K == PK_FMA_F32; }
constexpr bool uses_u(int K) { return K == CVT_I32_F32 || K == AND_B32 || K == MAD_U32 || K == ADD_U32 || K == LSHL_ADD || K == MOV_DPP || K == MIX_GOALSET; }

template <int K>
__global__ __launch_bounds__(256, 8) void k_peak(float* out, Stamp* stamps, unsigned long long window_ticks, float seed) {
    extern __shared__ char lds_pad[];  // only its size matters: it fixes the number of resident workgroups per CU
    float a[UNROLL];
    double d[4], dd[UNROLL];
    float2v v[UNROLL];
    unsigned u[UNROLL];
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
        if (uses_a(K)) a[i] = seed + i + threadIdx.x;
        if (uses_dd(K)) dd[i] = seed * 0.5 + i;
        if (uses_v(K)) v[i] = float2v{seed + i, seed + i + 1.0f};
        if (uses_u(K)) u[i] = threadIdx.x * 2654435761u + i;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = seed * 0.25 + i;
    const float b = seed * 0.999f, c = seed * 0.001f;
    const double bd = b, cd = c;
    const float2v b2 = float2v{b, b}, c2 = float2v{c, c};
    const unsigned m = 0xfffffff7u ^ (unsigned)seed;
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned long long r0 = wall_clock64();
    const unsigned long long c0 = __builtin_readcyclecounter();
    unsigned long long iters = 0, r1;
    do {
#pragma unroll 1
        for (int it = 0; it < INNER; ++it) body<K>(a, d, v, u, dd, b, c, bd, cd, b2, c2, m);
        iters += INNER;
        r1 = wall_clock64();
    } while (r1 - r0 < window_ticks);
    const unsigned long long c1 = __builtin_readcyclecounter();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < UNROLL; ++i) {
        if (uses_a(K)) s += a[i];
        if (uses_dd(K)) s += (float)dd[i];
        if (uses_v(K)) s += v[i].x + v[i].y;
        if (uses_u(K)) s += (float)u[i];
    }
    if (K == MIX_GOALSET) s += (float)(d[0] + d[1] + d[2]);
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) stamps[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{c0, c1, r0, r1, iters, hw, xcc};
}

static const char* g_dump_dir = nullptr;
struct Result { int kind, W; double cyc, share_min, share_max, clock_ghz, overlap, ms, ginstr; int simds, bad_simds; };

template <int K>
static Result run(int cus, int W, double window_ms, float* d_out, Stamp* d_st) {
    const int grid = cus * W;
    // the largest request of which W workgroups are resident on a CU at once (tools/lds_occupancy_probe.hip: the allocation has
    // a granule, 160 KiB / W is too much for W = 3, 5, 6): exactly W workgroups per CU, none waiting
    static const int lds_for[9] = {0, 163840, 80896, 53248, 40960, 31744, 26624, 22528, 19456};  // the LDS is handed out in 1280-byte granules, 128 per CU
    const size_t lds = (size_t)lds_for[W];
    if (lds > 64 * 1024) CHECK(hipFuncSetAttribute((const void*)k_peak<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned long long window = (unsigned long long)(window_ms * 1e5);  // 100 MHz ticks
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_peak<K>, dim3(grid), dim3(256), lds, 0, d_out, d_st, window / 16, 1.0001f);  // warm-up (code load, clocks)
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_peak<K>, dim3(grid), dim3(256), lds, 0, d_out, d_st, window, 1.0001f);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0.0f;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Stamp> h((size_t)grid * 4);
    CHECK(hipMemcpy(h.data(), d_st, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    if (g_dump_dir) {  // raw stamps for offline analysis
        char path[512];
        snprintf(path, sizeof path, "%s/stamps_k%d_w%d.bin", g_dump_dir, K, W);
        FILE* f = fopen(path, "wb");
        if (f) { fwrite(h.data(), sizeof(Stamp), h.size(), f); fclose(f); }
    }
    unsigned long long first_start = ~0ull, last_start = 0, first_end = ~0ull, last_end = 0;
    double clk_sum = 0.0, cyc_sum = 0.0, instr_all = 0.0, sec_sum = 0.0;
    struct PerSimd { int waves = 0; double instr = 0, lo = 1e300, hi = 0; };
    std::map<unsigned long long, PerSimd> per_simd;
    const double per_iter = (double)UNROLL * kind_instrs[K];
    for (const Stamp& s : h) {
        first_start = std::min(first_start, s.r0); last_start = std::max(last_start, s.r0);
        first_end = std::min(first_end, s.r1); last_end = std::max(last_end, s.r1);
        clk_sum += (double)(s.c1 - s.c0) / (double)(s.r1 - s.r0);  // shader cycles per 10 ns
        cyc_sum += (double)(s.c1 - s.c0);
        sec_sum += (double)(s.r1 - s.r0) * 1e-8;
        // HW_ID (gfx9): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID [3:0]
        PerSimd& ps = per_simd[((unsigned long long)(s.xcc & 0xf) << 32) | (s.hw & 0xff30u)];
        const double n = (double)s.iters * per_iter;
        ps.waves++; ps.instr += n; ps.lo = std::min(ps.lo, n); ps.hi = std::max(ps.hi, n);
        instr_all += n;
    }
    Result r{};
    r.kind = K; r.W = W; r.ms = ms;
    r.clock_ghz = clk_sum / h.size() / 10.0;
    // a SIMD's W waves issued instr_all / simds instructions during a window of cyc_sum / waves shader cycles
    r.cyc = (cyc_sum / h.size()) / (instr_all / per_simd.size());
    r.ginstr = instr_all / (sec_sum / h.size()) / 1e9;
    double smin = 1.0, smax = 0.0;
    for (auto& kv : per_simd) {
        r.bad_simds += kv.second.waves != W;
        smin = std::min(smin, kv.second.lo / kv.second.instr);
        smax = std::max(smax, kv.second.hi / kv.second.instr);
    }
    r.share_min = smin; r.share_max = smax;
    r.overlap = (double)((long long)first_end - (long long)last_start) / (double)(last_end - first_start);
    r.simds = (int)per_simd.size();
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return r;
}

typedef Result (*RunFn)(int, int, double, float*, Stamp*);
template <int K>
static void fill(RunFn* t) { t[K] = run<K>; fill<K + 1>(t); }
template <>
void fill<NKIND>(RunFn*) {}

int main(int argc, char** argv) {
    bool pmc = false;
    const char* csv = nullptr;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "pmc")) pmc = true;
        else if (!strcmp(argv[i], "dump") && i + 1 < argc) g_dump_dir = argv[++i];
        else csv = argv[i];
    }
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, nominal clock %d kHz; %d independent chains per lane, window 3 ms per kernel\n", prop.gcnArchName, cus, prop.clockRate, UNROLL);
    float* d_out; Stamp* d_st;
    CHECK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * sizeof(float)));
    CHECK(hipMalloc(&d_st, (size_t)cus * 8 * 4 * sizeof(Stamp)));
    RunFn table[NKIND];
    fill<0>(table);
    std::vector<Result> res;
    const std::vector<int> Ws = pmc ? std::vector<int>{2, 5, 6} : std::vector<int>{1, 2, 3, 4, 5, 6, 7};
    const std::vector<int> kinds_pmc = {FMA_F32, PK_FMA_F32, FMA_F64, CVT_F32_F64, MAD_U32, MIX_GOALSET};
    printf("%-22s %2s %14s %9s %9s %7s %10s %8s %9s %s\n", "instruction", "W", "cyc/instr/SIMD", "min share", "max share", "GHz", "G instr/s", "overlap", "kernel ms", "residency");
    for (int W : Ws)
        for (int k = 0; k < NKIND; ++k) {
            if (pmc && std::find(kinds_pmc.begin(), kinds_pmc.end(), k) == kinds_pmc.end()) continue;
            Result r{};
            for (int attempt = 0; attempt < 4; ++attempt) {  // a launch whose workgroups did not all start together is repeated
                r = table[k](cus, W, 3.0, d_out, d_st);
                if (r.overlap >= 0.95 && !r.bad_simds) break;
            }
            res.push_back(r);
            printf("%-22s %2d %14.3f %9.3f %9.3f %7.3f %10.1f %8.3f %9.3f %d SIMDs x %d waves%s\n", kind_name[k], W, r.cyc, r.share_min, r.share_max, r.clock_ghz,
                   r.ginstr, r.overlap, r.ms, r.simds, W, r.bad_simds ? " (UNEVEN)" : "");
        }
    if (csv) {
        FILE* f = fopen(csv, "w");
        fprintf(f, "instruction,waves_per_simd,cycles_per_instr_per_simd,min_wave_share,max_wave_share,shader_clock_ghz,chip_ginstr_per_s,overlap,kernel_ms,simds_seen,"
                   "simds_with_other_wave_count\n");
        for (const Result& r : res)
            fprintf(f, "%s,%d,%.4f,%.4f,%.4f,%.4f,%.2f,%.4f,%.4f,%d,%d\n", kind_name[r.kind], r.W, r.cyc, r.share_min, r.share_max, r.clock_ghz, r.ginstr, r.overlap,
                    r.ms, r.simds, r.bad_simds);
        fclose(f);
    }
    int bad = 0;
    for (const Result& r : res) bad += (r.overlap < 0.95) || r.bad_simds;
    printf("%d of %zu measurements fail the residency / overlap checks\n", bad, res.size());
    return 0;
}
