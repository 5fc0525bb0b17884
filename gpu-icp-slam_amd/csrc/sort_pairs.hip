// sort_pairs.hip -- 30-bit key / int value radix sort (hipCUB/rocPRIM) used to order particles along a
// space-filling curve before scoring.  Kept in its own translation unit: the rocPRIM headers dominate compile time.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

extern "C" int pfslam_sort_pairs_u32(void *tmp, size_t *tmp_bytes, const unsigned *keys_in, unsigned *keys_out,
                                     const int *vals_in, int *vals_out, int n, int end_bit, void *stream)
{
    return (int)hipcub::DeviceRadixSort::SortPairs(tmp, *tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, end_bit,
                                                   (hipStream_t)stream);
}
