// freddie_seg_sort.hip -- the one library sort of the segmentation library, in its own translation unit (rocPRIM's
// radix sort is a plain device sort; compiling it apart keeps the kernels' file quick to rebuild).
//
// fseg_upload() orders the read reps of every partition by their first position (freddie_seg.hip, "lanes"): the keys
// are (partition << 32 | first position), so one batch-wide stable sort gives the per-partition orders.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>

// tmp == nullptr: *tmp_bytes receives the temporary storage the sort of n pairs needs.
hipError_t fseg_sort_pairs(void *tmp, size_t *tmp_bytes, const unsigned long long *keys_in, unsigned long long *keys_out,
                           const int *vals_in, int *vals_out, size_t n, unsigned end_bit, hipStream_t stream) {
    return rocprim::radix_sort_pairs(tmp, *tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0u, end_bit, stream);
}
