// conv_mx_ar1.hip — conv3x3_mx_kernel instantiations of arithmetic AR = 1 (f16x2 + fp8 (x2q)); see conv_mx_kernel.h
#include "conv_mx_kernel.h"
namespace disco { template int dispatch_mx_ar<1>(const ConvMxArgs&, hipStream_t); }
