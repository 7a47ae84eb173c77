// Runs the power / clock probes of asm/powerprobe.py: for every kernel ~1.5 s of back-to-back launches (the chip settles at
// the clock it holds under that load), then 50 timed launches.  Prints cycles per unit (8 MFMAs 32x32x16 = 16 MFMAs 16x16x32),
// the in-kernel clock (s_memtime / s_memrealtime) and TFLOP/s of the MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <fstream>
#include <string>
#include <vector>
int main(int argc, char** argv) {
    if (argc < 3) { printf("usage: pw_run pw.hsaco pw.s.names [seconds]\n"); return 1; }
    const double settle = argc > 3 ? atof(argv[3]) : 1.5;
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<char> img((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    hipModule_t mod;
    if (hipModuleLoadData(&mod, img.data()) != hipSuccess) { printf("load failed\n"); return 2; }
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    unsigned *out, *src;
    hipMalloc(&out, 8192 * 4); hipMalloc(&src, 1 << 20);
    hipMemset(src, 0, 1 << 20);
    struct { void* out; void* src; unsigned iters; char pad[192 - 20]; } args;
    memset(&args, 0, sizeof(args));
    args.out = out; args.src = src;
    size_t size = sizeof(args);
    void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
    std::ifstream nf(argv[2]);
    std::string name;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double unit_flops = 8.0 * 32 * 32 * 16 * 2;
    while (std::getline(nf, name)) {
        if (name.empty()) continue;
        hipFunction_t fn;
        if (hipModuleGetFunction(&fn, mod, name.c_str()) != hipSuccess) { printf("%s: not found\n", name.c_str()); continue; }
        args.iters = 4000;   // ~ 4000 x 256..330 cycles = ~0.6 ms
        auto launch = [&]() { return hipModuleLaunchKernel(fn, cus, 1, 1, 256, 1, 1, 0, 0, nullptr, extra); };
        auto t0 = std::chrono::steady_clock::now();
        int n = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < settle) {
            for (int k = 0; k < 20; ++k) launch();
            if (hipDeviceSynchronize() != hipSuccess) { printf("%s: run failed\n", name.c_str()); return 3; }
            n += 20;
        }
        hipEventRecord(e0, 0);
        for (int k = 0; k < 50; ++k) launch();
        hipEventRecord(e1, 0);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: run failed\n", name.c_str()); return 3; }
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 50;
        std::vector<unsigned> h(8192);
        hipMemcpy(h.data(), out, (size_t)cus * 4 * 8, hipMemcpyDeviceToHost);
        std::vector<double> cyc, ghz, us;
        for (int w = 0; w < cus * 4; ++w) { cyc.push_back(h[2 * w]); ghz.push_back(h[2 * w] / (h[2 * w + 1] * 10.0)); us.push_back(h[2 * w + 1] * 0.01); }
        std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end()); std::sort(us.begin(), us.end());
        const double tf = unit_flops * args.iters * cus * 4 / (ms * 1e-3) / 1e12;
        printf("%-16s cycles/unit %6.1f (min %6.1f max %6.1f)  clock %.3f GHz (min %.3f max %.3f)  loop us %6.1f (min %6.1f max %6.1f)  %7.1f TFLOP/s  launch %.4f ms\n",
               name.c_str(), cyc[cyc.size() / 2] / args.iters, cyc.front() / args.iters, cyc.back() / args.iters, ghz[ghz.size() / 2], ghz.front(),
               ghz.back(), us[us.size() / 2], us.front(), us.back(), tf, ms);
        fflush(stdout);
    }
    return 0;
}
