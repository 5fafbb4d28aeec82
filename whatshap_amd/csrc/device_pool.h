// device_pool.h -- device buffers kept between calls (per process, every device): a create / solve used to hipMalloc 20 - 30 arrays
// and hipFree them again, each a driver round trip behind one driver lock (tables created on several host threads at once waited for
// each other there).  Blocks are taken from and given back to a pool, rounded to size classes (eight per power of two: at most 12.5 %
// over) so that the tables of one run reuse each other's.  whamd_release_caches() empties it.  Implemented in dp_device.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

namespace whamd {

// *got = the block's real size (the class): give exactly that back.
hipError_t devpool_take(int device, size_t bytes, void** out, size_t* got);
// The caller has made sure nothing on the device still uses the block (stream synchronised).
void devpool_give(int device, void* ptr, size_t bytes);
void devpool_release();

}  // namespace whamd
