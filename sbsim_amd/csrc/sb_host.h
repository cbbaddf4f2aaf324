// sb_host.h -- what the host-side translation units (sbsim_hip.hip, generators.hip) share: error
// reporting, device buffers, the handle behind the C ABI.
#ifndef SBSIM_AMD_SB_HOST_H_
#define SBSIM_AMD_SB_HOST_H_
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <vector>

#include "sb_device.h"

namespace sb {
namespace host {
inline thread_local std::string g_err; // sb_last_error()
} // namespace host
} // namespace sb

inline int fail(int code, const std::string &msg) {
  sb::host::g_err = msg;
  return code;
}

#define SB_HIP(call)                                                                     \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess)                                                                \
      return fail(SB_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_));        \
  } while (0)

// Entry points run on the handle's device and leave the calling thread's current device as
// they found it (a handle on device 1 must not redirect the caller's later torch allocations).
struct DeviceGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int device) {
    err = hipGetDevice(&prev);
    if (err == hipSuccess && prev != device) err = hipSetDevice(device);
    else if (err == hipSuccess) prev = -1; // nothing to restore
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define SB_ON_DEVICE(dev)                                                                    \
  DeviceGuard device_guard_(dev);                                                            \
  if (device_guard_.err != hipSuccess)                                                       \
    return fail(SB_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(device_guard_.err))

template <typename Tp>
struct DevBuf {
  Tp *p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
};

template <typename Tp>
inline int upload(DevBuf<Tp> &buf, const Tp *src, size_t n) {
  SB_HIP(hipMalloc((void **)&buf.p, std::max<size_t>(n, 1) * sizeof(Tp)));
  if (n) SB_HIP(hipMemcpy(buf.p, src, n * sizeof(Tp), hipMemcpyHostToDevice));
  return SB_OK;
}
template <typename Tp>
inline int alloc_zero(DevBuf<Tp> &buf, size_t n) {
  SB_HIP(hipMalloc((void **)&buf.p, std::max<size_t>(n, 1) * sizeof(Tp)));
  SB_HIP(hipMemset(buf.p, 0, std::max<size_t>(n, 1) * sizeof(Tp)));
  return SB_OK;
}


// A room cell of the convection shuffle (generators.hip).
struct ConvCell {
  int g0;                  // the cell's index in the caller's [H, W] grid: Philox counter, tie-break
  int gh;                  // ... in the handle's grid (which may be the transposed plan)
  int sidx;                // index into the building's state
  int pad;
  unsigned long long mask; // bit k: offset k of the table leads to a cell of the same room
};


struct sb_handle {
  sb::Dev d{};
  int device = 0, cus = 256;
  bool was_reset = false; // the first sb_reset also sets the construction-time device state
  int steps_since_reset = 0; // how far a reset rewinds the clock (the boiler's action age, scal[19])
  void *counters_zeroed_on = (void *)(uintptr_t)1; // the stream on which k_pre has zeroed the sweep kernel's draw counters since the last sweep launch (1: none)
  sb_launch_info info{};
  DevBuf<uint8_t> cls, tcls, tcset;
  DevBuf<double> abuf; // step_stream.hip: A = ap*Tprev + g of the buildings in flight
  DevBuf<double> ebuf; // step_stream_ms.hip: the scratch grid a pass writes when it reads the building's state (per resident workgroup)
  DevBuf<double> redo_scratch; // step_roll.hip: see Dev
  DevBuf<int> redo_ctr, redo_list, zs_off;
  DevBuf<int> act_kind, act_zone, zone_act; // the action vector's tables (sb_params.act_*) on the device
  DevBuf<double> act_lo, act_hi;
  DevBuf<double> tmul; // step_roll.hip: the tail scan's static multipliers
  DevBuf<double> ctab, csetab, temp, zmean, zair, damper, qz, scal, obs_mean, obs_sigma, ring, gtabg, zsum, gsum,
      hist_bins;
  DevBuf<int> czone, zone_off, zone_cells_l, mode, col_zone, zblk_zone, cell_state, nsw, next_b, src_dest,
      hist_col, hist_off;
  DevBuf<sb::Bld> bld;
  DevBuf<uint4> zl16;
  DevBuf<int4> sched;
  DevBuf<unsigned long long> smask, cmapS, amapS, zmapS;
  DevBuf<long long> dbg;
  // host copies for the optional generators
  std::vector<int> h_zone_off, h_zone_cells, h_state_index; // state index of every grid cell (< 0: exterior ring)
  // sb_convection_attach
  DevBuf<int> conv_local, conv_off; // per grid cell: index in its room's list; the offset table as linear steps
  DevBuf<int> conv_by_rank;         // whole-room shuffle: list index of the room's cell with raster rank r
  bool conv_whole_room = false;
  bool conv_wide = false, conv_transposed = false; // a window of more than 64 offsets: rejection sampling
  int conv_wide_d = 0;
  DevBuf<short> conv_room;          // per grid cell: its room (-1: none)
  DevBuf<ConvCell> conv_cells;
  DevBuf<unsigned short> conv_partner; // per room cell: its valid offsets' target cells (list indices), conv_pw entries per cell
  int conv_pw = 1;
  double conv_p = 0.0;
  int conv_n_off = 0, conv_max_room = 0;
  uint64_t conv_seed = 0;
  long long conv_first = 0;
  uint32_t conv_calls = 0;
  bool conv_attached = false;
  DevBuf<uint32_t> occ_state; // sb_occupancy_attach
  sb_occupancy_config occ{};
  uint32_t occ_queries = 0;
  bool occ_attached = false;
};


// generators.hip: the convection shuffle between the sweep and the reward (sb_step)
int sb_launch_convection(sb_handle *h, hipStream_t stream);

#endif // SBSIM_AMD_SB_HOST_H_
