// conv_mx_ar3.hip — conv3x3_mx_kernel instantiations of arithmetic AR = 3 (f16 + fp6x2); see conv_mx_kernel.h
#include "conv_mx_kernel.h"
namespace disco { template int dispatch_mx_ar<3>(const ConvMxArgs&, hipStream_t); }
#if MX_TIMELINE
MX_TIMELINE_EXPORT(disco_diag_conv_timeline)        // diagnostic builds only (tools/conv_timeline.py; not part of the ABI)
#endif
