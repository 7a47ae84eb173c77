// Runs every kernel of the micro-benchmark code object (asm/microbench.py) on all CUs and prints cycles per MFMA gap.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>
#include <fstream>
int main(int argc, char** argv) {
    if (argc < 3) { printf("usage: mb_run mb.hsaco mb.s.names\n"); return 1; }
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<char> img((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    hipModule_t mod;
    if (hipModuleLoadData(&mod, img.data()) != hipSuccess) { printf("load failed\n"); return 2; }
    unsigned *out, *src;
    hipMalloc(&out, 4096 * 4); hipMalloc(&src, 1 << 20);
    hipMemset(src, 0, 1 << 20);
    std::ifstream nf(argv[2]);
    std::string name;
    struct { void* out; void* src; char pad[192 - 16]; } args;
    memset(&args, 0, sizeof(args));
    args.out = out; args.src = src;
    size_t size = sizeof(args);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    while (std::getline(nf, name)) {
        if (name.empty()) continue;
        hipFunction_t fn;
        if (hipModuleGetFunction(&fn, mod, name.c_str()) != hipSuccess) { printf("%s: not found\n", name.c_str()); continue; }
        std::vector<unsigned> h(1024);
        double best = 1e30;
        for (int rep = 0; rep < 3; ++rep) {
            hipModuleLaunchKernel(fn, 256, 1, 1, 256, 1, 1, 0, 0, nullptr, extra);
            if (hipDeviceSynchronize() != hipSuccess) { printf("%s: run failed\n", name.c_str()); return 3; }
            hipMemcpy(h.data(), out, 4096, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            best = std::min(best, (double)h[512]);
        }
        printf("%-16s %7.2f cycles per MFMA gap\n", name.c_str(), best / 512.0);
    }
    return 0;
}
