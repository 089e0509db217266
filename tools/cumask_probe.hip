// cumask_probe.hip -- which CUs does a stream created with hipExtStreamCreateWithCUMask use on MI355X?
// Each block records (XCC_ID, SE_ID, CU_ID); the host prints the set of distinct CUs per mask pattern.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/cumask_probe tools/cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>

__global__ void probe(uint32_t *out, int spin)
{
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) out[blockIdx.x] = (xcc & 0xf) << 16 | ((hw >> 13) & 0x7) << 8 | ((hw >> 8) & 0xf) | ((hw >> 12) & 1) << 4;
}

static void run(const char *name, const std::vector<uint32_t> &mask)
{
    hipStream_t st;
    hipError_t e = mask.empty() ? hipStreamCreate(&st) : hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data());
    if (e != hipSuccess) { printf("%s: create failed: %s\n", name, hipGetErrorString(e)); return; }
    const int nb = 8192;
    uint32_t *d; hipMalloc(&d, nb * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, st);
    hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, st, d, 20000);
    hipEventRecord(b, st);
    hipStreamSynchronize(st);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    std::vector<uint32_t> h(nb); hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost);
    std::set<uint32_t> cus; int perx[16] = {0};
    for (uint32_t v : h) cus.insert(v);
    for (uint32_t v : cus) perx[(v >> 16) & 0xf]++;
    printf("%-28s distinct CUs %3zu  time %.2f ms  per XCC:", name, cus.size(), ms);
    for (int i = 0; i < 8; ++i) printf(" %d", perx[i]);
    printf("\n");
    hipFree(d); hipStreamDestroy(st);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s CUs %d\n", p.gcnArchName, p.multiProcessorCount);
    run("no mask", {});
    run("all 256 bits", std::vector<uint32_t>(8, 0xffffffffu));
    run("low 32 bits", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0});
    run("low 8 bits", {0xffu, 0, 0, 0, 0, 0, 0, 0});
    run("bits 0..15", {0xffffu, 0, 0, 0, 0, 0, 0, 0});
    run("word 7 only", {0, 0, 0, 0, 0, 0, 0, 0xffffffffu});
    run("all but low 16", {0xffff0000u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u});
    run("all but low 32", {0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u});
    run("every 8th bit", std::vector<uint32_t>(8, 0x01010101u));
    run("1-word mask 0xffff", {0xffffu});
    return 0;
}
