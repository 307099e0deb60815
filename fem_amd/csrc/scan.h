// fem_amd/csrc/scan.h -- exclusive prefix sum of int32 on the device
// (symbolic phase only: node->element offsets, slice offsets).
#pragma once
#include "common.h"

namespace fem {

// out[i] = sum_{k<i} in[k] for i in [0, n]; out has n+1 entries (out[n] = total).
// `in` and `out` may not alias.  tmp: at least cdiv(n, 1024) + 1 ints.
void exclusive_scan_i32(const int* in, int* out, size_t n, int* tmp, hipStream_t s);
size_t exclusive_scan_tmp_ints(size_t n);

}  // namespace fem
