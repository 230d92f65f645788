// Launchers shared between the fitters' translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
// fit_algebra.hip: the ordered chunk sum of the moments' per-chunk partials AND the per-instance algebra in one launch
// (M[B*K,52] is written as well: the backward pass and the parameter pack read it).
int cpfn_launch_reduce_algebra_fwd(const double *partial, int chunks, int B, int K, double *M, double *out, float *apex_axis32,
                                   hipStream_t stream);
