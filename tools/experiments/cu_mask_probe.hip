// Probe (GPU box): how hipExtStreamCreateWithCUMask's bit i maps to (XCC, SE, CU) on gfx950, and whether kernels on two
// streams with disjoint masks run side by side.  Build: hipcc --offload-arch=gfx950 -O2 tools/cu_mask_probe.hip -o tools/_build/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void where(uint32_t* out, int spin) {
    if (threadIdx.x == 0) {
        const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);    // HW_REG_HW_ID
        const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);  // HW_REG_XCC_ID
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc;
    }
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
}

static int run(const char* name, const std::vector<uint32_t>& mask, int nblocks) {
    hipStream_t st;
    CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()));
    uint32_t* d;
    CK(hipMalloc(&d, nblocks * 8));
    hipLaunchKernelGGL(where, dim3(nblocks), dim3(256), 100 * 1024, st, d, 20000);  // 100 KB LDS: one workgroup per CU
    CK(hipStreamSynchronize(st));
    std::vector<uint32_t> h(2 * nblocks);
    CK(hipMemcpy(h.data(), d, nblocks * 8, hipMemcpyDeviceToHost));
    std::map<uint32_t, int> seen;
    for (int i = 0; i < nblocks; ++i) {
        const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        seen[(xcc << 12) | (se << 8) | (sh << 4) | cu]++;
    }
    printf("%s: %zu distinct (xcc,se,sh,cu):", name, seen.size());
    for (auto& kv : seen) printf(" %u.%u.%u.%u", kv.first >> 12, (kv.first >> 8) & 0xf, (kv.first >> 4) & 0xf, kv.first & 0xf);
    printf("\n");
    uint32_t got[16] = {0};
    hipError_t e = hipExtStreamGetCUMask(st, 16, got);
    printf("  hipExtStreamGetCUMask -> %s %08x %08x %08x %08x %08x %08x %08x %08x\n", hipGetErrorString(e), got[0], got[1], got[2], got[3], got[4], got[5], got[6], got[7]);
    CK(hipFree(d));
    CK(hipStreamDestroy(st));
    return 0;
}

__global__ void busy(long long* out, int spin) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    if (threadIdx.x == 0) out[blockIdx.x] = t0;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("%s CUs %d\n", prop.gcnArchName, prop.multiProcessorCount);
    const int W = 8;  // 256 bits
    std::vector<uint32_t> m(W, 0);
    m[0] = 0xffffu;
    if (run("bits 0-15", m, 512)) return 1;
    m.assign(W, 0); m[0] = 0xffu;
    if (run("bits 0-7", m, 512)) return 1;
    m.assign(W, 0); m[0] = 0x1u;
    if (run("bit 0", m, 64)) return 1;
    m.assign(W, 0); m[0] = 0x100u;
    if (run("bit 8", m, 64)) return 1;
    m.assign(W, 0); m[1] = 0x1u;
    if (run("bit 32", m, 64)) return 1;
    m.assign(W, 0xffffffffu); m[0] = 0xffff0000u;
    if (run("all but 0-15", m, 2048)) return 1;
    m.assign(W, 0xffffffffu);
    if (run("all", m, 2048)) return 1;
    // concurrency: a long kernel on the big partition, then a short one on the small partition; does the short one start before the long one ends?
    std::vector<uint32_t> big(W, 0xffffffffu), small(W, 0);
    big[0] = 0xffff0000u; small[0] = 0xffffu;
    hipStream_t sb, ss;
    CK(hipExtStreamCreateWithCUMask(&sb, W, big.data()));
    CK(hipExtStreamCreateWithCUMask(&ss, W, small.data()));
    long long *db, *dsm;
    CK(hipMalloc(&db, 8192 * 8)); CK(hipMalloc(&dsm, 64 * 8));
    hipEvent_t e0, e1, e2, e3;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2)); CK(hipEventCreate(&e3));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, sb));
        hipLaunchKernelGGL(busy, dim3(8192), dim3(256), 26 * 1024, sb, db, 100000);   // ~1 ms per workgroup at 100 MHz wall clock? (printed below)
        CK(hipEventRecord(e1, sb));
        CK(hipEventRecord(e2, ss));
        hipLaunchKernelGGL(busy, dim3(16), dim3(256), 92 * 1024, ss, dsm, 1000);
        CK(hipEventRecord(e3, ss));
        CK(hipDeviceSynchronize());
        float tb, tsm, off;
        CK(hipEventElapsedTime(&tb, e0, e1)); CK(hipEventElapsedTime(&tsm, e2, e3)); CK(hipEventElapsedTime(&off, e0, e3));
        printf("rep %d: big kernel %.3f ms, small kernel %.3f ms, small finished %.3f ms after the big one started\n", rep, tb, tsm, off);
    }
    // (3) no masks: does a 92 KB-LDS kernel on another stream get onto the chip while a 26 KB-LDS kernel with thousands of
    //     pending workgroups is running?  With default and with high stream priority.
    for (int prio = 0; prio < 2; ++prio) {
        hipStream_t s1, s2;
        int lo = 0, hi = 0;
        CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, lo));
        CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, prio ? hi : lo));
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, s1));
            hipLaunchKernelGGL(busy, dim3(8192), dim3(256), 26 * 1024, s1, db, 100000);
            CK(hipEventRecord(e1, s1));
            CK(hipEventRecord(e2, s2));
            hipLaunchKernelGGL(busy, dim3(16), dim3(256), 92 * 1024, s2, dsm, 1000);
            CK(hipEventRecord(e3, s2));
            CK(hipDeviceSynchronize());
            float tb, tsm, off;
            CK(hipEventElapsedTime(&tb, e0, e1)); CK(hipEventElapsedTime(&tsm, e2, e3)); CK(hipEventElapsedTime(&off, e0, e3));
            printf("unmasked, small stream priority %d (range %d..%d) rep %d: big %.3f ms, small %.3f ms, small finished %.3f ms after the big one started\n",
                   prio ? hi : lo, lo, hi, rep, tb, tsm, off);
        }
    }
    return 0;
}
