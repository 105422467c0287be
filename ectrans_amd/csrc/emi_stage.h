// emi_stage.h -- host arrays through device memory (mem_space == EMI_MEM_HOST): what the Fortran / transi callers of
// the reference's CPU interface hand over (dir_trans.h / inv_trans.h: plain host arrays).  The device-side staging
// buffers are kept between calls (a pool, best fit): hipMalloc / hipFree of several GiB per call cost twice what the
// copies themselves do (TCo1279, 128 fields: 1023 -> 331 ms per pair, 51 GB/s over the link of 57 / 55 GB/s).  The
// copies are plain asynchronous hipMemcpy calls on the caller's stream: the runtime pins pageable memory in place and
// reaches the same rate as on pinned memory (measured: 335 against 331 ms; a hand-made pipeline of pinned bounce
// buffers filled by 8 / 16 host threads was slower, 358 / 435 ms, and was dropped).
// EMI_STAGE_POOL=0 frees the device buffers after every call.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "emi_rt.h"

namespace emi_stage {

struct Buf {
  void *p;
  size_t cap;
  bool busy;
};
inline std::vector<Buf> &pool() {
  static std::vector<Buf> v;
  return v;
}
inline bool pool_enabled() {
  static const bool on = !(getenv("EMI_STAGE_POOL") && atoi(getenv("EMI_STAGE_POOL")) == 0);
  return on;
}
inline size_t idle_bytes() {
  size_t s = 0;
  for (auto &b : pool())
    if (!b.busy) s += b.cap;
  return s;
}
// frees the idle buffers (before a large allocation elsewhere, at TRANS_END)
inline void trim() {
  auto &v = pool();
  for (size_t i = 0; i < v.size();) {
    if (!v[i].busy) {
      emi_dev_free(v[i].p);
      v.erase(v.begin() + (long)i);
    } else {
      i++;
    }
  }
}
inline void *acquire(size_t bytes) {
  auto &v = pool();
  const size_t need = std::max(bytes, (size_t)256);
  const size_t cap = (need + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);  // capacities are multiples of 2 MiB
  int best = -1;
  for (int i = 0; i < (int)v.size(); i++)
    if (!v[i].busy && v[i].cap >= need && (best < 0 || v[i].cap < v[best].cap)) best = i;
  if (best >= 0 && v[best].cap <= 2 * cap) {  // not a buffer of a much larger call
    v[best].busy = true;
    return v[best].p;
  }
  void *p = nullptr;
  if (emi_dev_malloc(&p, cap)) return nullptr;  // (emi_dev_malloc itself retries once after trim())
  v.push_back(Buf{p, cap, true});
  return p;
}
inline void release(void *p) {
  auto &v = pool();
  for (size_t i = 0; i < v.size(); i++)
    if (v[i].p == p) {
      if (pool_enabled()) {
        v[i].busy = false;
      } else {
        emi_dev_free(p);
        v.erase(v.begin() + (long)i);
      }
      return;
    }
}

}  // namespace emi_stage
