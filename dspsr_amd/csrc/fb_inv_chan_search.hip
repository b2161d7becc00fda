// k_inv_chan with the search-mode epilogue: square-law detection + time scrunch inside the inverse pass (FbOut kind 5)
#include "fb_inv_chan.h"

namespace dspsr_amd {

template <int... I> static k3_t pick3s(int logf, bool full, iseq<I...>)
{
  static const k3_t t[] = {k_inv_chan<I, 2, -1>...};
  static const k3_t f[] = {k_inv_chan<I, 2, full_logt(I)>...};
  return full ? f[logf] : t[logf];
}
k3_t fb_pick3s(int logf, bool full) { return pick3s(logf, full, seq_t()); }

}  // namespace dspsr_amd
