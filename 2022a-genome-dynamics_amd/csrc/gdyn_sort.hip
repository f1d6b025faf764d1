// gdyn_sort.hip -- the one library call of libgdyn: rocPRIM's radix sort, used when a contact map is dumped
// (gd_contacts_fetch: contact_map::accumulate() lists the count matrix in row-major order,
// simulation_interphase/contact_map.cc:77-91; here: the occupied table slots sorted by the key i << 32 | j).
// A translation unit of its own so that the template-heavy header stays out of the stepping kernels' build.
#include <hip/hip_runtime.h>

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "gdyn_types.h"

hipError_t gd_sort_contacts(void *tmp, size_t *tmp_bytes, const unsigned long long *kin, unsigned long long *kout, const unsigned *vin,
                            unsigned *vout, size_t n, unsigned key_bits, hipStream_t st)
{
    return rocprim::radix_sort_pairs(tmp, *tmp_bytes, kin, kout, vin, vout, n, 0u, key_bits, st);
}
