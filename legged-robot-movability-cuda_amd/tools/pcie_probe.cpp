// pcie_probe.cpp -- what the host boundary can get out of PCIe on this box (not part of the product):
//   hipcc -O2 -o build/pcie_probe pcie_probe.cpp -lpthread && ./build/pcie_probe
// pageable copies one at a time / both directions from two threads, hipHostRegister cost, registered (pinned in place)
// copies both directions at once.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main() {
    const size_t nin = 120u << 20, nout = 130u << 20;
    char *hin = (char*)malloc(nin), *hout = (char*)malloc(nout);
    memset(hin, 1, nin);
    memset(hout, 2, nout); // touched: no first-touch faults inside the timed copies
    void *din, *dout;
    CK(hipMalloc(&din, nin)); CK(hipMalloc(&dout, nout));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now(); CK(hipMemcpy(din, hin, nin, hipMemcpyHostToDevice)); double t1 = now();
        CK(hipMemcpy(hout, dout, nout, hipMemcpyDeviceToHost)); double t2 = now();
        printf("pageable, one after the other: H2D %.2f ms (%.1f GB/s), D2H %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, nin / (t1 - t0) / 1e9, (t2 - t1) * 1e3, nout / (t2 - t1) / 1e9);
        t0 = now();
        std::thread a([&] { CK(hipMemcpyAsync(din, hin, nin, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); });
        std::thread b([&] { CK(hipMemcpyAsync(hout, dout, nout, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)); });
        a.join(); b.join();
        t1 = now();
        printf("pageable, both directions from two threads: %.2f ms (%.1f GB/s in + out)\n", (t1 - t0) * 1e3, (nin + nout) / (t1 - t0) / 1e9);
        // chunked, two threads
        t0 = now();
        const size_t ch = 12u << 20;
        std::thread c([&] { for (size_t o = 0; o < nin; o += ch) CK(hipMemcpyAsync((char*)din + o, hin + o, std::min(ch, nin - o), hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1)); });
        std::thread d([&] { for (size_t o = 0; o < nout; o += ch) CK(hipMemcpyAsync(hout + o, (char*)dout + o, std::min(ch, nout - o), hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2)); });
        c.join(); d.join();
        t1 = now();
        printf("pageable, 12 MB chunks, two threads: %.2f ms (%.1f GB/s in + out)\n", (t1 - t0) * 1e3, (nin + nout) / (t1 - t0) / 1e9);
    }
    for (int rep = 0; rep < 2; rep++) {
        double t0 = now();
        CK(hipHostRegister(hin, nin, hipHostRegisterDefault)); CK(hipHostRegister(hout, nout, hipHostRegisterDefault));
        double t1 = now();
        CK(hipMemcpyAsync(din, hin, nin, hipMemcpyHostToDevice, s1)); CK(hipStreamSynchronize(s1));
        double t2 = now();
        CK(hipMemcpyAsync(hout, dout, nout, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s2));
        double t3 = now();
        CK(hipMemcpyAsync(din, hin, nin, hipMemcpyHostToDevice, s1)); CK(hipMemcpyAsync(hout, dout, nout, hipMemcpyDeviceToHost, s2));
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        double t4 = now();
        CK(hipHostUnregister(hin)); CK(hipHostUnregister(hout));
        double t5 = now();
        printf("registered in place: register %.2f ms, H2D %.2f ms (%.1f GB/s), D2H %.2f ms (%.1f GB/s), both at once %.2f ms (%.1f GB/s), unregister %.2f ms\n",
               (t1 - t0) * 1e3, (t2 - t1) * 1e3, nin / (t2 - t1) / 1e9, (t3 - t2) * 1e3, nout / (t3 - t2) / 1e9, (t4 - t3) * 1e3, (nin + nout) / (t4 - t3) / 1e9, (t5 - t4) * 1e3);
    }
    // pinned staging + memcpy
    {
        char* pin; CK(hipHostMalloc((void**)&pin, nin, hipHostMallocDefault));
        double t0 = now(); memcpy(pin, hin, nin); double t1 = now();
        printf("memcpy user -> pinned staging, one thread: %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, nin / (t1 - t0) / 1e9);
        t0 = now();
        std::vector<std::thread> ts;
        for (int k = 0; k < 8; k++) ts.emplace_back([&, k] { const size_t part = nin / 8; memcpy(pin + k * part, hin + k * part, part); });
        for (auto& t : ts) t.join();
        t1 = now();
        printf("memcpy user -> pinned staging, 8 threads: %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, nin / (t1 - t0) / 1e9);
        char* fresh = (char*)malloc(nout);
        t0 = now(); CK(hipMemcpy(fresh, dout, nout, hipMemcpyDeviceToHost)); t1 = now();
        printf("pageable D2H into UNTOUCHED malloc memory: %.2f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, nout / (t1 - t0) / 1e9);
    }
    return 0;
}
