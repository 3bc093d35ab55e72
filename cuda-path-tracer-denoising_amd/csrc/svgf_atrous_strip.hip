// svgf_atrous_strip.hip — LDS strip-marching a-trous kernel (fast path).  Placeholder until the kernel lands:
// reports "unsupported" so the host falls back to the strict gather kernel.
#include "svgf_kernels.h"

bool atrous_strip_supported(const AtrousArgs &) { return false; }
hipError_t launch_atrous_strip(const AtrousArgs &, hipStream_t) { return hipErrorNotSupported; }
