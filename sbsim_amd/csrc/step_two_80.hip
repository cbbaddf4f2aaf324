// step_two_80.hip -- k_sweep_two's instantiations for 80 slots per lane (step_two_impl.h; step_two.hip dispatches)
#include "step_two_impl.h"

namespace sb {

int sweep_two_run80(const Dev &d, hipStream_t stream, bool prepare) {
  return d.T > 0 ? launch_v<80, true>(d, stream, prepare) : launch_v<80, false>(d, stream, prepare);
}

} // namespace sb
