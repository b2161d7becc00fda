// k_inv_chan with the fused fold
#include "fb_inv_chan.h"

namespace dspsr_amd {

template <int... I> static k3_t pick3f(int logf, bool full, iseq<I...>)
{
  static const k3_t t[] = {k_inv_chan<I, 1, -1>...};
  static const k3_t f[] = {k_inv_chan<I, 1, full_logt(I)>...};
  return full ? f[logf] : t[logf];
}
k3_t fb_pick3f(int logf, bool full) { return pick3f(logf, full, seq_t()); }

}  // namespace dspsr_amd

FB_ST_READER(inv_chan_fold)
