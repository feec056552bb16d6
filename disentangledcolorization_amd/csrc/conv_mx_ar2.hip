// conv_mx_ar2.hip — conv3x3_mx_kernel instantiations of arithmetic AR = 2 (f16x3); see conv_mx_kernel.h
#include "conv_mx_kernel.h"
namespace disco { template int dispatch_mx_ar<2>(const ConvMxArgs&, hipStream_t); }
#if MX_TIMELINE
MX_TIMELINE_EXPORT(disco_diag_conv_timeline_x3)     // diagnostic builds only (tools/conv_timeline.py --x3; not part of the ABI)
#endif
