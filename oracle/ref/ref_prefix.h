// Force-included ahead of the hipified reference sources (see Makefile, step 3).
// rocThrust is pulled in first with its normal HIP configuration (include guards make the reference's own
// `#include <thrust/...>` lines no-ops afterwards); then __CUDACC__ is defined so that the reference's vendored
// glm 0.9.6.3 (external/include/glm/detail/setup.hpp:198,809-811) declares its functions __host__ __device__.
#include <hip/hip_runtime.h>
#include <thrust/execution_policy.h>
#include <thrust/random.h>
#include <thrust/remove.h>
#include <thrust/partition.h>
#define __CUDACC__ 1
#define CUDA_VERSION 10000
