// step_two_64.hip -- k_sweep_two's instantiations for 64 slots per lane (step_two_impl.h; step_two.hip dispatches)
#include "step_two_impl.h"

namespace sb {

int sweep_two_run64(const Dev &d, hipStream_t stream, bool prepare) {
  return d.T > 0 ? launch_v<64, true>(d, stream, prepare) : launch_v<64, false>(d, stream, prepare);
}

} // namespace sb
