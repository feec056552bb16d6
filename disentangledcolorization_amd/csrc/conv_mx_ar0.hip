// conv_mx_ar0.hip — conv3x3_mx_kernel instantiations of arithmetic AR = 0 (f16 + fp8x2); see conv_mx_kernel.h
#include "conv_mx_kernel.h"
namespace disco { template int dispatch_mx_ar<0>(const ConvMxArgs&, hipStream_t); }
