// step_band.hip -- mode 5, one row per lane on two to four wavefronts per building: what the planner asks and the
// dispatch to the instantiations, which live in one translation unit per slot count (step_band_68.hip ..
// step_band_96.hip <- step_band_impl.h) so that they compile in parallel.
#include "sb_device.h"
#include "step_band_cfg.h"

namespace sb {

int sweep_band_run68(const Dev &d, hipStream_t stream, bool prepare);
int sweep_band_run72(const Dev &d, hipStream_t stream, bool prepare);
int sweep_band_run76(const Dev &d, hipStream_t stream, bool prepare);
int sweep_band_run80(const Dev &d, hipStream_t stream, bool prepare);
int sweep_band_run84(const Dev &d, hipStream_t stream, bool prepare);
int sweep_band_run88(const Dev &d, hipStream_t stream, bool prepare);
int sweep_band_run92(const Dev &d, hipStream_t stream, bool prepare);
int sweep_band_run96(const Dev &d, hipStream_t stream, bool prepare);

namespace {
using namespace band;

int dispatch(const Dev &d, hipStream_t stream, bool prepare) {
  if (d.RS < 128 || d.RS > 64 * kWMax || d.RS % 64) return (int)hipErrorInvalidValue;
  switch (d.NR) {
    case 68: return sweep_band_run68(d, stream, prepare);
    case 72: return sweep_band_run72(d, stream, prepare);
    case 76: return sweep_band_run76(d, stream, prepare);
    case 80: return sweep_band_run80(d, stream, prepare);
    case 84: return sweep_band_run84(d, stream, prepare);
    case 88: return sweep_band_run88(d, stream, prepare);
    case 92: return sweep_band_run92(d, stream, prepare);
    case 96: return sweep_band_run96(d, stream, prepare);
  }
  return (int)hipErrorInvalidValue;
}
} // namespace

bool sweep_band_supported(int NR) {
  for (int s : kSlotCounts)
    if (s == NR) return true;
  return false;
}
int sweep_band_max_waves() { return kWMax; }
int sweep_band_lds_slots(int NR) { return lds_slots(NR); }
int sweep_band_seam_doubles(int NR, int W) { return (2 * W + 2) * seam_region(NR) + W * (64 + NR + 8); }
int sweep_band_sync_doubles(int W) { return 32 * W + 2 * W * kHist + 8; }
int sweep_band_decision_lag(int NR, int W) { (void)NR; return W; } // periods until a sweep's max|delta| is known for certain
int sweep_band_set_table() { return kSets; }
int prepare_sweep_band(const Dev &d) { return dispatch(d, nullptr, true); }
int launch_sweep_band(const Dev &d, hipStream_t stream) { return dispatch(d, stream, false); }

} // namespace sb
