// Where does workgroup b of a launch run?  Prints blockIdx, XCC_ID (hardware register 20) and HW_ID (register 4) fields for a launch of
// 1280 workgroups of 256 threads: the persistent planner kernel groups workgroups by XCD and elects update CUs from these numbers.
//   hipcc --offload-arch=gfx950 -O2 tools/experiments/xcc_probe.hip -o tools/_build/xcc_probe && tools/_build/xcc_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>

__global__ void probe(unsigned* out) {
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
    }
    // keep the workgroups resident together for a moment
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
}

int main() {
    const int N = 1280;
    unsigned* d;
    hipMalloc(&d, N * 2 * sizeof(unsigned));
    hipLaunchKernelGGL(probe, dim3(N), dim3(256), 30000, 0, d);
    std::vector<unsigned> h(2 * N);
    hipMemcpy(h.data(), d, h.size() * sizeof(unsigned), hipMemcpyDeviceToHost);
    int agree = 0;
    std::map<unsigned, std::set<unsigned>> cus;  // xcc -> distinct (hw_id >> 8 & 0xff)
    std::map<unsigned, int> per_key;
    for (int b = 0; b < N; ++b) {
        const unsigned xcc = h[2 * b] & 15u, hw = h[2 * b + 1];
        agree += (xcc == (unsigned)(b & 7));
        cus[xcc].insert((hw >> 8) & 0xffu);
        per_key[(xcc << 8) | ((hw >> 8) & 0xffu)]++;
        if (b < 24) printf("block %4d  xcc %u  hw_id 0x%08x  cu %u sh %u se %u  simd %u wave %u\n", b, xcc, hw, (hw >> 8) & 15u, (hw >> 12) & 1u, (hw >> 13) & 7u, (hw >> 4) & 3u, hw & 15u);
    }
    printf("xcc == block %% 8 for %d of %d workgroups\n", agree, N);
    for (auto& kv : cus) printf("xcc %u: %zu distinct CU keys\n", kv.first, kv.second.size());
    std::map<int, int> hist;
    for (auto& kv : per_key) hist[kv.second]++;
    for (auto& kv : hist) printf("%d CU keys hold %d workgroups\n", kv.second, kv.first);
    return 0;
}
