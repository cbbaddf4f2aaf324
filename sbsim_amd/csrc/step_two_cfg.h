// step_two_cfg.h -- what step_two.hip (the planner's interface) and step_two_impl.h (the kernel) agree on.
#pragma once

namespace sb {
namespace two {

constexpr int kSets = 32;  // entries of the coefficient-set table (at LDS address 0)
// Slots of A kept in LDS, by level (step_two_impl.h): level 0: two buildings per CU, 1: three, 2: four.
constexpr int kLevels = 3;
constexpr int lds_slots(int NR, int level) {
  return NR <= 64 ? (level == 0 ? 64 : level == 1 ? 46 : 34)
         : NR <= 76 ? (level == 0 ? 72 : level == 1 ? 46 : 34) : (level == 0 ? 74 : level == 1 ? 56 : 42);
}
constexpr int a_stride_of(int NL) { return 2 * NL + 2; } // 2 mod 4 doubles: conflict-free ds_read_b128 across 16 lanes
constexpr int tail_row(int NR) { return NR + 4; }        // first tail row in LDS: column c at [2 + c], zero guards

} // namespace two
} // namespace sb
