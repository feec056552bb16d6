// conv_mx_ar3.hip — conv3x3_mx_kernel instantiations of arithmetic AR = 3 (f16 + fp6x2); see conv_mx_kernel.h
#include "conv_mx_kernel.h"
namespace disco { template int dispatch_mx_ar<3>(const ConvMxArgs&, hipStream_t); }
#if MX_TIMELINE
// diagnostic builds only (tools/conv_timeline.py; not part of the ABI): read back / clear the phase stamps of workgroup (0, 0)
extern "C" int disco_diag_conv_timeline(unsigned long long* h_dst /* [2][8192] */, int clear) {
    if (clear) {
        static unsigned long long zeros[2][disco::MX_TL_EVENTS];
        return hipMemcpyToSymbol(HIP_SYMBOL(disco::g_mx_tl), zeros, sizeof(zeros)) == hipSuccess ? 0 : -1;
    }
    return hipMemcpyFromSymbol(h_dst, HIP_SYMBOL(disco::g_mx_tl), sizeof(unsigned long long) * 2 * disco::MX_TL_EVENTS) == hipSuccess ? 0 : -1;
}
#endif
