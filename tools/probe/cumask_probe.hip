// Which XCD / shader engine / compute unit does each bit of a HIP stream's CU mask select on gfx950?
// Build: hipcc --offload-arch=gfx950 -O2 -o cumask_probe cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <vector>
__global__ void k_where(uint32_t* out)
{
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    // spin a little so that blocks spread over every allowed compute unit
    uint64_t t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 200000) {}
    if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xf) << 16) | (hw & 0xffff);
}
static void run(const char* name, const std::vector<int>& bits)
{
    uint32_t mask[8] = {0};
    for (int b : bits) mask[b >> 5] |= 1u << (b & 31);
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, 8, mask) != hipSuccess) { printf("%s: stream failed\n", name); return; }
    const int nb = 4096;
    uint32_t* d; hipMalloc(&d, nb * 4);
    k_where<<<nb, 64, 0, s>>>(d);
    hipStreamSynchronize(s);
    std::vector<uint32_t> h(nb);
    hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost);
    std::set<int> xccs; std::set<uint32_t> cus;
    for (uint32_t v : h) { xccs.insert(v >> 16); cus.insert(((v >> 16) << 16) | (v & 0xff00)); }   // cu_id 11:8, sh 12, se 15:13
    printf("%-28s: %zu distinct (xcc,se,sh,cu); xcc ids {", name, cus.size());
    for (int x : xccs) printf(" %d", x);
    printf(" }\n");
    hipFree(d); hipStreamDestroy(s);
}
int main()
{
    std::vector<int> v;
    v.clear(); for (int i = 0; i < 256; ++i) v.push_back(i);            run("all 256 bits", v);
    v.clear(); for (int i = 0; i < 128; ++i) v.push_back(i);            run("bits 0-127", v);
    v.clear(); for (int i = 128; i < 256; ++i) v.push_back(i);          run("bits 128-255", v);
    v.clear(); for (int i = 0; i < 256; ++i) if (i % 8 < 4) v.push_back(i);  run("bits i%8<4", v);
    v.clear(); for (int i = 0; i < 256; ++i) if (i % 8 == 0) v.push_back(i); run("bits i%8==0", v);
    v.clear(); for (int i = 0; i < 32; ++i) v.push_back(i);             run("bits 0-31", v);
    v.clear(); for (int i = 0; i < 8; ++i) v.push_back(i);              run("bits 0-7", v);
    v.clear(); v.push_back(0);                                          run("bit 0", v);
    v.clear(); v.push_back(1);                                          run("bit 1", v);
    v.clear(); v.push_back(8);                                          run("bit 8", v);
    return 0;
}
