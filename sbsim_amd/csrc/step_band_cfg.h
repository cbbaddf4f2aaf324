// step_band_cfg.h -- what step_band.hip (the planner's interface) and step_band_impl.h (the kernel) agree on.
#pragma once

namespace sb {
namespace band {

constexpr int kSets = 32;  // entries of the coefficient-set table (at LDS address 0)
#ifndef SB_BAND_GRP
#define SB_BAND_GRP 16 // (8: the sweeps 5-9 % slower, tools/bench_mid_plans.py: every check drains the wavefront's LDS queue)
#endif
constexpr int kGrp = SB_BAND_GRP; // steps between two checks of the neighbouring wavefronts' progress
constexpr int kHist = 16;  // ring of published max|delta| parts, by sweep number: wavefront 0 may be five sweeps ahead of the last one, whose decisions look back five more (8 entries: overwritten under a reader -- found by the spin limit's trap)
constexpr int kWMax = 4;   // wavefronts per building: one per SIMD
// Slots of A kept in LDS (the rest: registers) = A's row stride: even (the steps go in pairs, one
// ds_read_b128 per pair) and 2 mod 4 doubles (16-byte aligned rows, 16 lanes' ds_read_b128 cover all
// banks).  66: two wavefronts' A (67.6 KB) + tables + seam rows stay under 80 KB -- two buildings of two
// wavefronts per CU -- and four wavefronts' A (135 KB) under 160 KB.  A row holds slot j at position
// (j + 1) mod NR: the pair of an odd step is 16-byte aligned.
constexpr int lds_slots(int NR) { return 66; }
constexpr int seam_region(int NR) { return NR + 72; }                    // doubles per seam row: 64 finite ones in front (steps < 63)
constexpr int kSlotCounts[] = {68, 72, 76, 80, 84, 88, 92, 96};                  // the instantiations (step_band_NN.hip)

} // namespace band
} // namespace sb
