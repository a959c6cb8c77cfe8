// Which XCD does workgroup b of a launch run on?  (MI355X_MICROARCH.md: "blocks are dealt round-robin over the
// 8 XCDs ... read the id from HW_REG_XCC_ID".)  Launches the geometry of k_spmm_sweep -- 256 x T workgroups of
// 1024 threads with `lds` bytes of dynamic LDS -- and prints, per label b % 8, the XCC ids its workgroups saw.
//   hipcc --offload-arch=gfx950 -O2 tools/xcc_probe.hip -o /tmp/xcc_probe && /tmp/xcc_probe [tiles] [lds_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(1024) void k_probe(int *out) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) {
        lds[0] = 1.f;
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        out[blockIdx.y * gridDim.x + blockIdx.x] = static_cast<int>(xcc & 0xf);
    }
    // stay resident for a while so that the whole grid's placement is the steady-state one
    for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
}

int main(int argc, char **argv) {
    const int tiles = argc > 1 ? std::atoi(argv[1]) : 1;
    const int lds = argc > 2 ? std::atoi(argv[2]) : 32768;
    int *d = nullptr;
    const int n = 256 * tiles;
    hipMalloc(&d, sizeof(int) * n);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(d, 0xff, sizeof(int) * n);
        k_probe<<<dim3(256, tiles), 1024, lds>>>(d);
        std::vector<int> h(n);
        hipMemcpy(h.data(), d, sizeof(int) * n, hipMemcpyDeviceToHost);
        int consistent = 0;
        for (int label = 0; label < 8; ++label) {
            int hist[16] = {0};
            for (int b = label; b < n; b += 8) hist[h[b] & 15]++;
            std::printf("rep %d label %d:", rep, label);
            int best = 0;
            for (int x = 0; x < 16; ++x)
                if (hist[x]) {
                    std::printf(" xcc%d x%d", x, hist[x]);
                    best = hist[x] > best ? hist[x] : best;
                }
            std::printf("\n");
            consistent += best;
        }
        std::printf("rep %d: %d of %d workgroups on the XCD their label's majority runs on\n", rep, consistent, n);
    }
    hipFree(d);
    return 0;
}
