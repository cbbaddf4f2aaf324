// step_two.hip -- mode 4, the two-rows-per-lane sweep kernel: what the planner asks (slot counts, LDS levels) and
// the dispatch to the instantiations, which live in one translation unit per slot count (step_two_64.hip, step_two_76.hip,
// step_two_80.hip <- step_two_impl.h) so that they compile in parallel.
#include "sb_device.h"
#include "step_two_cfg.h"

namespace sb {

int sweep_two_run64(const Dev &d, hipStream_t stream, bool prepare);
int sweep_two_run76(const Dev &d, hipStream_t stream, bool prepare);
int sweep_two_run80(const Dev &d, hipStream_t stream, bool prepare);

namespace {
using namespace two;

int dispatch(const Dev &d, hipStream_t stream, bool prepare) {
  if (d.NR == 64) return sweep_two_run64(d, stream, prepare);
  if (d.NR == 76) return sweep_two_run76(d, stream, prepare);
  if (d.NR == 80) return sweep_two_run80(d, stream, prepare);
  return (int)hipErrorInvalidValue;
}
} // namespace

bool sweep_two_supported(int NR) { return NR == 64 || NR == 76 || NR == 80; }
int sweep_two_levels() { return kLevels; }
int sweep_two_lds_slots(int NR, int level) { return lds_slots(NR, level); }
int sweep_two_a_stride(int NR, int level) { return a_stride_of(lds_slots(NR, level)); }
int sweep_two_seam_doubles(int NR) { return tail_row(NR); }
int sweep_two_set_table() { return kSets; }
int prepare_sweep_two(const Dev &d) { return dispatch(d, nullptr, true); }
int launch_sweep_two(const Dev &d, hipStream_t stream) { return dispatch(d, stream, false); }

} // namespace sb
