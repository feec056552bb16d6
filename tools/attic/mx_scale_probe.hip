// Probe of v_mfma_scale_f32_32x32x64_f8f6f4's scale operands: which rows / K blocks does lane L's scale byte apply to?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
template <int OPA, int OPB>
__global__ void k(const int* sa, const int* sb, float* d) {
    const int lane = threadIdx.x;
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0x38383838; }   // e4m3 1.0
    f32x16 acc; for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, OPA, sa[lane], OPB, sb[lane]);
    for (int e = 0; e < 16; ++e) d[lane * 16 + e] = acc[e];
}
int main() {
    int *dsa, *dsb; float* dd; hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dd, 4096);
    std::vector<float> hd(1024);
    auto show = [&](const char* what) {
        hipMemcpy(hd.data(), dd, 4096, hipMemcpyDeviceToHost);
        // D[row][col]: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
        float D[32][32];
        for (int lane = 0; lane < 64; ++lane) for (int reg = 0; reg < 16; ++reg) D[(reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)][lane & 31] = hd[lane * 16 + reg];
        printf("%s: ", what);
        int shown = 0;
        for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) if (D[r][c] != 64.f && shown < 6) { printf("D[%d][%d]=%g ", r, c, D[r][c]); ++shown; }
        int cnt = 0; for (int r = 0; r < 32; ++r) for (int c = 0; c < 32; ++c) cnt += D[r][c] != 64.f;
        printf(" (%d entries != 64; D[0][0]=%g)\n", cnt, D[0][0]);
    };
    const int Ls[] = {0, 1, 5, 31, 32, 33, 63};
    for (int which = 0; which < 2; ++which)
        for (int L : Ls) {
            std::vector<int> hsa(64, 127), hsb(64, 127);
            (which ? hsb : hsa)[L] = 128;
            hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice);
            hipLaunchKernelGGL((k<0, 0>), dim3(1), dim3(64), 0, 0, dsa, dsb, dd);
            char buf[64]; snprintf(buf, 64, "scale_%c[lane %d]=128", which ? 'b' : 'a', L); show(buf);
        }
    {   // opsel: bytes of the dword
        std::vector<int> hsa(64, 0x7f7f7f7f), hsb(64, 0x7f7f7f7f);
        for (int l = 0; l < 64; ++l) hsa[l] = 0x82818080;      // byte0 = 128 (x2), byte1 = 128, byte2 = 129 (x4), byte3 = 130 (x8)
        hsa[3] = 0x82818085;
        hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL((k<0, 0>), dim3(1), dim3(64), 0, 0, dsa, dsb, dd); show("opsel_a=0 bytes {80,80,81,82}, lane3 byte0=85");
        hipLaunchKernelGGL((k<1, 0>), dim3(1), dim3(64), 0, 0, dsa, dsb, dd); show("opsel_a=1");
        hipLaunchKernelGGL((k<2, 0>), dim3(1), dim3(64), 0, 0, dsa, dsb, dd); show("opsel_a=2");
        hipLaunchKernelGGL((k<3, 0>), dim3(1), dim3(64), 0, 0, dsa, dsb, dd); show("opsel_a=3");
    }
    return 0;
}
