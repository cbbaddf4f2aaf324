// step_band_68.hip -- k_sweep_band's instantiation for 68 slots per lane (step_band_impl.h; step_band.hip dispatches)
#include "step_band_impl.h"

namespace sb {

int sweep_band_run68(const Dev &d, hipStream_t stream, bool prepare) { return launch<68>(d, stream, prepare); }

} // namespace sb
