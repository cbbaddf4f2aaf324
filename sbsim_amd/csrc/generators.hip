// generators.hip -- optional per-building input generators on the device:
// RandomizedArrivalDepartureOccupancy and the stochastic convection shuffle, both driven by the
// counter-based Philox4x32-10 so that results do not depend on how a batch is sharded.
#include "sb_host.h"

#include <cstdlib>

using namespace sb;

namespace {

constexpr int kLdsCap = 160 * 1024;

// ---------------------------------------------------------------- randomized occupancy
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11): ten
// rounds of two 32x32 -> 64 multiplies on a 128-bit counter under a 64-bit key.
__device__ inline void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

struct OccArgs {
  uint32_t *state; // [B][Z]: bit i = occupant i of the zone is at WORK
  int B, Z, n_occ, hour, workday, e_arr, l_arr, e_dep;
  double p_arr, p_dep;
  uint64_t seed;
  long long first_building;
  uint32_t query;
  float *count, *total;
};

// ZoneOccupant.peek (randomized_arrival_departure_occupancy.py:138-160) for every occupant of
// every zone of one building per thread.  Uniform of occupant i: word i & 3 of the Philox block
// with counter (building lo, building hi, zone, query * 8 + (i >> 2)), u = (x >> 8) / 2^24.
__global__ void k_occupancy(OccArgs o) {
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < o.B; b += gridDim.x * blockDim.x) {
    const unsigned long long gb = (unsigned long long)(o.first_building + b);
    float tot = 0.0f;
    for (int z = 0; z < o.Z; ++z) {
      uint32_t st = o.state[(size_t)b * o.Z + z];
      if (!o.workday) st = 0;
      else {
        const bool arr_open = !(o.hour < o.e_arr || o.hour > o.l_arr), dep_open = !(o.hour < o.e_dep);
        if (arr_open || dep_open)
          for (int blk = 0; blk * 4 < o.n_occ; ++blk) {
            uint32_t c[4] = {(uint32_t)gb, (uint32_t)(gb >> 32), (uint32_t)z, o.query * 8u + (uint32_t)blk};
            philox4x32_10(c, (uint32_t)o.seed, (uint32_t)(o.seed >> 32));
            for (int k = 0; k < 4 && blk * 4 + k < o.n_occ; ++k) {
              const int i = blk * 4 + k;
              const double u = (double)(c[k] >> 8) * (1.0 / 16777216.0);
              const bool at_work = (st >> i) & 1u;
              if (!at_work && arr_open && u < o.p_arr) st |= 1u << i;
              else if (at_work && dep_open && u < o.p_dep) st &= ~(1u << i);
            }
          }
      }
      o.state[(size_t)b * o.Z + z] = st;
      const float n = (float)__popc(st);
      if (o.count) o.count[(size_t)b * o.Z + z] = n;
      tot += n;
    }
    if (o.total) o.total[b] = tot;
  }
}

// ---------------------------------------------------------------- stochastic convection
// StochasticConvectionSimulator._shuffle_max_dist (stochastic_convection_simulator.py:101-145),
// one workgroup per building, room by room.  Every air cell of a room is, with probability p,
// the first half of a swap whose second half is drawn uniformly from the room's cells within
// the offset table (dx^2 + dy^2 <= distance inside the [-distance, distance) window, :125-131);
// the swaps are applied one after the other in a uniformly random order (:137).  Random order:
// swap i gets the time stamp T_i = (32 random bits, i) and swaps run by increasing T.  Instead of
// replaying the sequence, every VALUE is followed through it: at cell c after time t the next
// swap touching c is the earliest of c's own swap and the swaps that chose c (a linked list per
// cell, built with LDS atomics); the value moves to that swap's other cell.  All values move
// in parallel, reads before writes.  Randomness: Philox4x32-10, key = seed, counter = (global
// building lo, hi, call number, grid cell): word 0 -> inclusion, 1 -> choice, 2 -> time stamp.
struct ConvArgs {
  double *temp;
  size_t stride; // doubles per building
  const int *zone_off, *local, *off;
  const ConvCell *cells; // every zone's cells, within a zone by increasing state index: lanes that
                         // read / write neighbouring cells of the list touch neighbouring memory
  int B, Z, W, n_off, max_room;
  // windows of more than 64 offsets (distance >= 20; distance = -1 with p < 1 is the reference's 1000):
  // the partner by rejection sampling inside the disc (oracle/convection_oracle.py wide_partner)
  int wide, wide_r, wide_d, H, transposed;
  const short *room; // [N] the cell's room in the handle's grid (-1: none)
  const unsigned short *partner; // [cells][pw]: a cell's valid offsets, in table order, as indices of the target cells in the room's list
  int pw;
  const int *by_rank; // by_rank[zone_off[z] + r]: index (in the zone's cell list) of the room's r-th cell in the CALLER's raster order
  double p;
  uint64_t seed;
  long long first_building;
  uint32_t call;
};

constexpr int kConvMaxThreads = 512, kConvMaxPerLane = 8;

// The whole-room shuffle (distance = -1, p = 1: stochastic_convection_simulator.py:78-99,
// _shuffle_no_max_dist): the values of a room's cells are permuted uniformly at random
// (random.shuffle).  Here: a keyed bijection on the room's cells in the CALLER's raster order
// (rank r of a cell = its position in the zone's cell list), dest = pi(r): three rounds of
// (odd multiply, add, xor-shift) on k = ceil(log2 n) bits, walked until the result is < n (cycle
// walking keeps it a bijection on [0, n)).  Keys: Philox4x32-10, key = seed, counters (global
// building lo, hi, call number, 0x80000000 | 2 * room [+1]).  Not the Fisher-Yates of the reference --
// no counter-based generator reproduces Python's global random -- but its displacement statistics
// are the reference's (tests/test_convection.py).
struct PermKeys { uint32_t m[3], a[3]; int k, s0, s1; uint32_t mask; };
__host__ __device__ inline uint32_t perm_apply(const PermKeys &K, uint32_t x, uint32_t n) {
  do {
    x = (x * K.m[0] + K.a[0]) & K.mask; x ^= x >> K.s0;
    x = (x * K.m[1] + K.a[1]) & K.mask; x ^= x >> K.s1;
    x = (x * K.m[2] + K.a[2]) & K.mask; x ^= x >> K.s0;
  } while (x >= n);
  return x;
}
constexpr int kConvMaxRoom = 2047; // a cell's index in its room fits 11 bits next to the 20-bit grid index

// The shuffle's random numbers: a counter-based mixer -- MurmurHash3's 32-bit finaliser over (stream, cell,
// word number) -- an eighth of Philox4x32-10's instructions (which was a quarter of this kernel's).  stream:
// (seed, global building, call number) folded once per building into TWO 32-bit words (s0, s1) by two independent
// functions (round 5, ADVICE r4: one word gave 65,536 buildings 0.5 colliding pairs per call -- whole buildings making
// identical draws; now each word is a bijection of the building's number); a cell's key (index g0 in the CALLER's grid) = (s0 ^ g0 * 0x9E3779B1) + s1 (an xor and an
// add: two buildings' keys agree on every cell only when both words do),
// word k of the cell = fmix32(key + k * 0x6C8E9CF5).  Word 0 -> inclusion (u = (x >> 8) / 2^24,
// included unless u > p; not formed when p >= 1), word 1 -> the swap's time stamp (its top 20 bits; ties by the cell's
// rank in the room) and, from its low 12 bits, the pick among the cell's partner list (round 5: one mixer round for
// both -- v_mul_lo_u32 is a quarter-rate instruction and the draws were a third of the kernel's VALU time), words 2.. ->
// partner candidates of the wide windows until one is accepted.  Known answers: tests/test_convection.py;
// restated in oracle/convection_oracle.py.  (Occupancy and the whole-room permutation keep Philox.)
__host__ __device__ inline uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}
struct ConvStream { uint32_t s0, s1; };
// What the two words take from (seed, call): the same for every building of a launch -- the compiler hoists both out of
// the building loop (two scalar registers).
struct ConvSalt { uint32_t a, b; };
__host__ __device__ inline ConvSalt conv_salt(uint64_t seed, uint32_t call) {
  uint32_t h = fmix32((uint32_t)seed ^ 0x9E3779B9u);
  h = fmix32(h ^ (uint32_t)(seed >> 32));
  h = fmix32(h ^ call);
  uint32_t g = fmix32(call ^ 0x7F4A7C15u);
  g = fmix32(g + (uint32_t)(seed >> 32));
  g = fmix32(g + (uint32_t)seed);
  return ConvSalt{h, g};
}
// Per building ONE round per word, side by side, over the building's number modulo 2^32 (sb_convection_attach refuses a
// batch that reaches beyond): each word is a bijection of it, so no two buildings of a launch share even one.  Every
// wavefront walks this scalar chain twice per room and building right after a barrier: at five rounds per word it was 13 %
// of k_convect's time, at five and one 3 %.
__host__ __device__ inline ConvStream conv_stream(ConvSalt salt, uint32_t gb) {
  return ConvStream{fmix32(gb ^ salt.a), fmix32(gb * 0x9E3779B1u + salt.b)};
}
__host__ __device__ inline ConvStream conv_stream(uint64_t seed, uint64_t gb, uint32_t call) { return conv_stream(conv_salt(seed, call), (uint32_t)gb); }
__host__ __device__ inline uint32_t conv_key(ConvStream st, uint32_t g0) { return (st.s0 ^ (g0 * 0x9E3779B1u)) + st.s1; }
__host__ __device__ inline uint32_t conv_word(uint32_t cell_key, uint32_t k) { return fmix32(cell_key + k * 0x6C8E9CF5u); }

// A cell's own swap in LDS, 8 bytes: its time stamp ((top 20 bits of word 1) << 11 | the cell's rank in the
// room) + 1 -- 0: the cell starts no swap; unique within a room, so the order of the swaps is total and does not
// depend on the state layout -- and the swap's other cell (low 16 bits) with the next swap that chose the same
// cell as this one (high 16 bits: a list link, 0xffff: end).  head[c]: the first swap that chose cell c.
constexpr uint32_t kConvEnd = 0xffffu;
constexpr int kConvMaxTries = 1 << 14; // partner candidates of a wide-window draw (the cell itself is one: accepted sooner or later)

// Q: cells per lane; the workgroup has ceil(largest room / Q) lanes rounded up to a wavefront.
// Room-major: the lane keeps its cells' table entries in registers while the workgroup walks
// through its share of the buildings.
template <int Q>
__global__ void __launch_bounds__(kConvMaxThreads) __attribute__((amdgpu_waves_per_eu(Q <= 2 ? 8 : Q == 3 ? 6 : Q == 4 ? 4 : 2)))
k_convect(ConvArgs o) {
  const int kConvThreads = blockDim.x;
  extern __shared__ __attribute__((aligned(16))) unsigned long long conv_lds[];
  uint2 *rec = (uint2 *)conv_lds;                // [max_room] {stamp, other | next << 16}
  double *vout = (double *)(rec + o.max_room);   // [max_room] the value that ends in this cell
  uint32_t *head = (uint32_t *)(vout + o.max_room); // [max_room] first swap that chose this cell (kConvEnd: none)
  __shared__ int soff[64];                       // the offset table as steps in the handle's grid
  const int tid = threadIdx.x;
  if (tid < 64) soff[tid] = tid < o.n_off ? o.off[tid] : 0;
  __syncthreads();
  for (int z = 0; z < o.Z; ++z) {
    const int c0 = o.zone_off[z], n = o.zone_off[z + 1] - c0;
    // the lane's cells: what every building needs of them, in registers (the rest of a ConvCell -- its index in the
    // handle's grid -- only the wide-window path reads, from memory: eight waves per SIMD need <= 64 registers)
    // two words per cell: its index in the caller's grid (20 bits) | its partner count << 20; its index in the state
    // (21 bits) | its rank in its room << 21 (sb_convection_attach checks the widths)
    uint32_t c_a[Q], c_b[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int i = tid + q * kConvThreads;
      const ConvCell c = o.cells[c0 + (i < n ? i : 0)];
      c_a[q] = (uint32_t)c.g0 | ((uint32_t)__popcll(c.mask) << 20); c_b[q] = (uint32_t)c.sidx | ((uint32_t)c.pad << 21);
    }
    for (int i = tid; i < n; i += kConvThreads) head[i] = kConvEnd; // (re-armed with every building's store phase)
    __syncthreads();
    // the draws of a building -- its values (HBM), the partner (a table read: L1 / L2), the time stamps: the Q reads of
    // a lane are in flight together, and the NEXT building's are issued before this building's follow phase (which
    // waits for LDS only), so a building's memory latency hides behind its predecessor's pointer chase
    auto draw = [&](int b, double (&val)[Q], int (&oth)[Q]) {
      const double *st = o.temp + (size_t)b * o.stride;
      const ConvStream stream = conv_stream(o.seed, (uint64_t)(o.first_building + b), o.call);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const int i = tid + q * kConvThreads;
          oth[q] = i;
          if (i < n) {
            val[q] = st[c_b[q] & 0x1fffffu];
            const uint32_t key = conv_key(stream, c_a[q] & 0xfffffu);
            bool in = true; // :119 (u < 1: with p >= 1 every cell starts a swap and word 0 is not formed)
            if (o.p < 1.0) in = !((double)(conv_word(key, 0) >> 8) * (1.0 / 16777216.0) > o.p);
            int other = i;
            if (in && o.wide) { // :119, a window too large for an offset table: the partner by rejection --
              // uniform over the reference's candidate list (:122-131; the cell itself is a candidate)
              const int span = 2 * o.wide_r + 1, W0 = o.transposed ? o.H : o.W; // W0: row length of the caller's grid
              if (n < span * span) { // the room has fewer cells than the window's box: a cell of the room (by rank), kept when inside the disc
                const int g0 = (int)(c_a[q] & 0xfffffu), x0 = g0 / W0, y0 = g0 - x0 * W0;
                for (int k = 2; k < kConvMaxTries; ++k) {
                  const int j = o.by_rank[c0 + (int)(((unsigned long long)conv_word(key, (uint32_t)k) * (unsigned long long)n) >> 32)];
                  const int gj = o.cells[c0 + j].g0, dx = gj / W0 - x0, dy = gj - (gj / W0) * W0 - y0;
                  if (dx * dx + dy * dy <= o.wide_d) { other = j; break; }
                }
              } else { // a square of the box, kept when inside the disc and in the room
                const int gh = o.cells[c0 + i].gh, xh = gh / o.W, yh = gh - xh * o.W;
                for (int k = 2; k < kConvMaxTries; ++k) {
                  const uint32_t w = conv_word(key, (uint32_t)k);
                  const int dx = (int)(((w & 0xffffu) * (uint32_t)span) >> 16) - o.wide_r;
                  const int dy = (int)(((w >> 16) * (uint32_t)span) >> 16) - o.wide_r;
                  const int hx = xh + (o.transposed ? dy : dx), hy = yh + (o.transposed ? dx : dy);
                  if (dx * dx + dy * dy <= o.wide_d && hx >= 0 && hx < o.H && hy >= 0 && hy < o.W && (int)o.room[hx * o.W + hy] == z) {
                    other = o.local[hx * o.W + hy];
                    break;
                  }
                }
              }
            } else if (in) { // :119: uniform over the room's cells inside the offset window -- the cell's own
              // partner list (the valid offsets in (dx, dy) raster order, as list indices; built by sb_convection_attach);
              // the pick from the low 12 bits of the word whose top 20 are the time stamp (at most 64 offsets)
              const int cnt = (int)(c_a[q] >> 20);
              const int pick = (int)(((conv_word(key, 1) & 0xfffu) * (uint32_t)cnt) >> 12);
#if defined(SB_CONV_ABL) && (SB_CONV_ABL & 2) // ... without the partner table's gather
              other = (i + 1 + pick) % n;
#else
              other = (int)o.partner[(size_t)(c0 + i) * (size_t)o.pw + (size_t)pick];
#endif
            }
            oth[q] = other;
          }
        }
    };
    double val[Q], val_n[Q];
    int oth[Q], oth_n[Q];
    if ((int)blockIdx.x < o.B) draw(blockIdx.x, val, oth);
#pragma unroll
    for (int q = 0; q < Q; ++q) asm volatile("" : "+v"(val[q]), "+v"(oth[q])); // waited for here, not at the loop's top (see the hand-over below)
    for (int b = blockIdx.x; b < o.B; b += gridDim.x) {
      double *st = o.temp + (size_t)b * o.stride;
#if defined(SB_CONV_ABL) && (SB_CONV_ABL & 8) // ... nothing but the loads and the stores (no barrier, no LDS): the memory pattern's own time
      {
        const bool more8 = b + (int)gridDim.x < o.B;
        if (more8) draw(b + (int)gridDim.x, val_n, oth_n);
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          const int i = tid + q * kConvThreads;
          if (i < n) st[c_b[q] & 0x1fffffu] = val[q] + (double)(oth[q] - i) * 1e-300;
        }
        if (more8) {
#pragma unroll
          for (int q = 0; q < Q; ++q) { val[q] = val_n[q]; oth[q] = oth_n[q]; }
        }
        continue;
      }
#endif
      const ConvStream stream = conv_stream(o.seed, (uint64_t)(o.first_building + b), o.call); // (scalar: cheaper formed again than carried from the draw)
      // the records and the list links (LDS atomics): the lane's Q exchanges first, their results read after the time
      // stamps are formed (an exchange whose result is used under its own branch is waited for there: Q round trips in a row)
      uint32_t nxt[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int i = tid + q * kConvThreads;
        nxt[q] = kConvEnd;
#if defined(SB_CONV_ABL) && (SB_CONV_ABL & 4) // ... without the list exchanges
        if (i < n && oth[q] != i) nxt[q] = (uint32_t)oth[q];
#else
        if (i < n && oth[q] != i) nxt[q] = atomicExch(&head[oth[q]], (uint32_t)i);
#endif
      }
#pragma unroll
      for (int q = 0; q < Q; ++q) asm volatile("" : "+v"(nxt[q])); // (keeps the compiler from sinking the result's first use into the branch)
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int i = tid + q * kConvThreads;
        if (i < n) {
          const int other = oth[q];
          const uint32_t key = conv_key(stream, c_a[q] & 0xfffffu);
          uint2 r;
          r.x = other != i ? (((conv_word(key, 1) >> 12) << 11) | (c_b[q] >> 21)) + 1u : 0u; // the time stamp; pad: the cell's rank
          r.y = (uint32_t)other | (nxt[q] << 16);
          rec[i] = r;
        }
      }
      __syncthreads();
      const bool more = b + (int)gridDim.x < o.B;
      if (more) draw(b + (int)gridDim.x, val_n, oth_n);
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int i = tid + q * kConvThreads;
        if (i < n) { // follow the value through the swaps that touch the cell it sits in
          int pos = i, to = -1;
          uint32_t t = 0, best = 0xffffffffu;
#if defined(SB_CONV_ABL) && (SB_CONV_ABL & 1) // developer experiment (tools/build_variant.sh): the kernel without the pointer chase
          if (o.B > 0) { vout[pos] = val[q]; continue; }
#endif
          uint2 R = rec[pos];
          uint32_t j = head[pos];
          if (R.x > t) { best = R.x; to = (int)(R.y & 0xffffu); }
          for (;;) { // (one loop over hops and list nodes alike, and a rejection loop over the offsets instead of the
            // partner list, were both slower: scalar branch overhead, lanes waiting for the longest loop; round 5: ONE loop
            // over the lane's Q chains, hops and nodes -- a third fewer record reads per wavefront by the model, 75
            // instructions per iteration as compiled: 5.4 against 3.6 ms.  The chase is bound by instruction issue.)
            while (j != kConvEnd) { // the swaps that chose this cell
              const uint2 N = rec[j];
              if (N.x > t && N.x < best) { best = N.x; to = (int)j; }
              j = N.y >> 16;
            }
            if (to < 0) break;
            t = best; pos = to;    // the earliest later swap moves the value
            R = rec[pos];
            j = head[pos];
            best = 0xffffffffu; to = -1;
            if (R.x > t) { best = R.x; to = (int)(R.y & 0xffffu); }
          }
          vout[pos] = val[q];
        }
      }
      // the next building's draw becomes this one's BEFORE the stores below are issued: the wait for its loads would
      // otherwise also wait for them (memory operations complete in order) -- an HBM write round trip per room and building
      if (more) {
#pragma unroll
        for (int q = 0; q < Q; ++q) { val[q] = val_n[q]; oth[q] = oth_n[q]; }
#pragma unroll
        for (int q = 0; q < Q; ++q) asm volatile("" : "+v"(val[q]), "+v"(oth[q]));
      }
      __syncthreads();
      double vo[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int i = tid + q * kConvThreads;
        vo[q] = vout[i < n ? i : 0];
      }
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        const int i = tid + q * kConvThreads;
        if (i < n) {
          st[c_b[q] & 0x1fffffu] = vo[q];
          head[i] = kConvEnd; // for the next building (the follow phase is over: every lane is past the barrier above)
        }
      }
      __syncthreads();
    }
  }
}

struct ConvAllArgs {
  double *temp;
  size_t stride;
  const int *zone_off, *by_rank; // by_rank[zone_off[z] + r]: index (in the zone's ConvCell list) of the cell with rank r
  const ConvCell *cells;         // .pad = the cell's rank
  int B, Z;
  uint64_t seed;
  long long first_building;
  uint32_t call;
};

// One workgroup per building at a time, room by room: read every value, barrier, write it to
// its destination cell.
__global__ void __launch_bounds__(256) k_convect_all(ConvAllArgs o) {
  for (int b = blockIdx.x; b < o.B; b += gridDim.x) {
    double *st = o.temp + (size_t)b * o.stride;
    const unsigned long long gb = (unsigned long long)(o.first_building + b);
    for (int z = 0; z < o.Z; ++z) {
      const int c0 = o.zone_off[z], n = o.zone_off[z + 1] - c0;
      if (n < 2) continue;
      PermKeys K;
      uint32_t c[4] = {(uint32_t)gb, (uint32_t)(gb >> 32), o.call, 0x80000000u | (uint32_t)(2 * z)};
      uint32_t d[4] = {(uint32_t)gb, (uint32_t)(gb >> 32), o.call, 0x80000000u | (uint32_t)(2 * z + 1)};
      philox4x32_10(c, (uint32_t)o.seed, (uint32_t)(o.seed >> 32));
      philox4x32_10(d, (uint32_t)o.seed, (uint32_t)(o.seed >> 32));
      K.k = 32 - __clz(n - 1);
      K.mask = K.k >= 32 ? 0xffffffffu : ((1u << K.k) - 1u);
      K.s0 = max(1, K.k / 2); K.s1 = max(1, (K.k + 1) / 3);
      K.m[0] = c[0] | 1u; K.m[1] = c[1] | 1u; K.m[2] = c[2] | 1u;
      K.a[0] = c[3]; K.a[1] = d[0]; K.a[2] = d[1];
      constexpr int kQ = 8; // cells per lane: rooms of up to 2048 cells (sb_convection_attach checks)
      double val[kQ];
      int dst[kQ];
#pragma unroll
      for (int q = 0; q < kQ; ++q) {
        const int i = threadIdx.x + q * blockDim.x;
        if (i < n) {
          const ConvCell cc = o.cells[c0 + i];
          val[q] = st[cc.sidx];
          const int j = o.by_rank[c0 + (int)perm_apply(K, (uint32_t)cc.pad, (uint32_t)n)];
          dst[q] = o.cells[c0 + j].sidx;
        }
      }
      __syncthreads(); // every value of the room is read before the first one is written
#pragma unroll
      for (int q = 0; q < kQ; ++q) {
        const int i = threadIdx.x + q * blockDim.x;
        if (i < n) st[dst[q]] = val[q];
      }
      __syncthreads();
    }
  }
}

} // namespace

extern "C" {

int sb_occupancy_attach(sb_handle *h, const sb_occupancy_config *cfg) {
  if (!h || !cfg) return fail(SB_ERR_INVALID, "sb_occupancy_attach: null argument");
  if (cfg->zone_assignment < 1 || cfg->zone_assignment > 32)
    return fail(SB_ERR_INVALID, "sb_occupancy_attach: zone_assignment must be 1..32 (one state word per zone)");
  if (!(cfg->earliest_arrival_hour < cfg->latest_arrival_hour &&
        cfg->latest_arrival_hour < cfg->earliest_departure_hour &&
        cfg->earliest_departure_hour < cfg->latest_departure_hour))
    return fail(SB_ERR_INVALID, "sb_occupancy_attach: hours must be strictly increasing "
                                "(randomized_arrival_departure_occupancy.py:68-73)");
  if (!(cfg->time_step_sec > 0) || cfg->first_building < 0)
    return fail(SB_ERR_INVALID, "sb_occupancy_attach: time_step_sec must be positive, first_building >= 0");
  SB_ON_DEVICE(h->device);
  if (!h->occ_state.p) {
    const int rc = alloc_zero(h->occ_state, (size_t)h->d.B * h->d.Z);
    if (rc != SB_OK) return rc;
  } else SB_HIP(hipMemset(h->occ_state.p, 0, (size_t)h->d.B * h->d.Z * sizeof(uint32_t)));
  h->occ = *cfg;
  h->occ_queries = 0;
  h->occ_attached = true;
  return SB_OK;
}

int sb_occupancy_peek(sb_handle *h, int32_t local_hour, int32_t is_work_day, float *count_dev,
                      float *total_dev, void *stream) {
  if (!h) return fail(SB_ERR_INVALID, "sb_occupancy_peek: null handle");
  if (!h->occ_attached) return fail(SB_ERR_INVALID, "sb_occupancy_peek: sb_occupancy_attach first");
  if (local_hour < 0 || local_hour > 23) return fail(SB_ERR_INVALID, "sb_occupancy_peek: hour must be 0..23");
  SB_ON_DEVICE(h->device);
  const sb_occupancy_config &c = h->occ;
  OccArgs o;
  o.state = h->occ_state.p; o.B = h->d.B; o.Z = h->d.Z; o.n_occ = c.zone_assignment;
  o.hour = local_hour; o.workday = is_work_day != 0;
  o.e_arr = c.earliest_arrival_hour; o.l_arr = c.latest_arrival_hour; o.e_dep = c.earliest_departure_hour;
  // _get_event_probability (:100-112): 1 / (half the window in time steps)
  o.p_arr = 1.0 / ((double)(c.latest_arrival_hour - c.earliest_arrival_hour) * 3600.0 / c.time_step_sec / 2.0);
  o.p_dep = 1.0 / ((double)(c.latest_departure_hour - c.earliest_departure_hour) * 3600.0 / c.time_step_sec / 2.0);
  o.seed = c.seed; o.first_building = c.first_building; o.query = h->occ_queries++;
  o.count = count_dev; o.total = total_dev;
  const int blocks = std::max(1, std::min((o.B + 63) / 64, 4096));
  hipLaunchKernelGGL(k_occupancy, dim3(blocks), dim3(64), 0, (hipStream_t)stream, o);
  SB_HIP(hipGetLastError());
  return SB_OK;
}

int sb_convection_attach(sb_handle *h, double p, int32_t distance, uint64_t seed, int64_t first_building,
                         int32_t transposed) {
  if (!h) return fail(SB_ERR_INVALID, "sb_convection_attach: null handle");
  if (!(p >= 0.0 && p <= 1.0)) return fail(SB_ERR_INVALID, "sb_convection_attach: p must be in [0, 1]");
  if (first_building < 0) return fail(SB_ERR_INVALID, "sb_convection_attach: first_building must be >= 0");
  if (p == 0.0 || distance == 0) { h->conv_attached = false; return SB_OK; } // stochastic_convection_simulator.py:70-71
  const bool whole_room = distance == -1 && p == 1.0; // stochastic_convection_simulator.py:78: the special case
  if (distance == -1 && !whole_room) distance = 1000; // :108-109: then the window is 1000 cells
  if (!whole_room && (distance < 0 || distance > 1000))
    return fail(SB_ERR_UNSUPPORTED, "sb_convection_attach: distance must be 1..1000, or -1 (p = 1: the whole-room shuffle, "
                                    "p < 1: the reference's 1000-cell window)");
  SB_ON_DEVICE(h->device);
  const Dev &d = h->d;
  if (whole_room) distance = 0; // no offset window
  // :125-131: window [-distance, distance) in both directions, squared distance <= distance
  // (in the order of the caller's grid: the handle may hold the transposed floor plan)
  if (d.N >= (1 << 20)) return fail(SB_ERR_UNSUPPORTED, "sb_convection_attach: more than 2^20 grid cells");
  if ((d.reg ? (size_t)d.state_doubles : (size_t)d.Np) >= ((size_t)1 << 21))
    return fail(SB_ERR_UNSUPPORTED, "sb_convection_attach: a building's state of more than 2^21 doubles");
  std::vector<int> odx, ody, offd; // offsets in the handle's coordinates, linear steps
  const int reach = std::min(distance, 32); // dx^2 <= distance <= 1000
  for (int dx = -reach; dx < reach; ++dx)
    for (int dy = -reach; dy < reach; ++dy)
      if (dx >= -distance && dy >= -distance && dx * dx + dy * dy <= distance) {
        odx.push_back(transposed ? dy : dx);
        ody.push_back(transposed ? dx : dy);
        offd.push_back(odx.back() * d.W + ody.back());
      }
  // more than 64 offsets (distance >= 20): no offset table, no per-cell mask -- the kernel samples the
  // partner by rejection inside the disc (the half-open box [-d, d) no longer cuts it: sqrt(d) < d)
  const bool wide = offd.size() > 64;
  if (wide) { odx.clear(); ody.clear(); offd.assign(1, 0); }
  std::vector<int> room((size_t)d.N, -1), local((size_t)d.N, -1);
  for (int z = 0; z < d.Z; ++z)
    for (int i = h->h_zone_off[z]; i < h->h_zone_off[z + 1]; ++i) room[h->h_zone_cells[i]] = z;
  std::vector<ConvCell> cells(h->h_zone_cells.size());
  const int pw = wide ? 1 : (int)((offd.size() + 1) & ~(size_t)1);
  std::vector<unsigned short> partner(h->h_zone_cells.size() * (size_t)pw, 0);
  std::vector<int> by_rank(h->h_zone_cells.size(), 0);
  int max_room = 1;
  for (int z = 0; z < d.Z; ++z) {
    const int c0 = h->h_zone_off[z], n = h->h_zone_off[z + 1] - c0;
    max_room = std::max(max_room, n);
    std::vector<int> order(h->h_zone_cells.begin() + c0, h->h_zone_cells.begin() + c0 + n);
    for (int g : order)
      if (h->h_state_index[g] < 0) return fail(SB_ERR_INVALID, "sb_convection_attach: a zone cell lies in the exterior ring");
    std::sort(order.begin(), order.end(), [&](int a, int b) { return h->h_state_index[a] < h->h_state_index[b]; });
    // rank of a cell in the CALLER's raster order (the whole-room shuffle permutes ranks)
    std::vector<int> by_g0(order);
    auto g0_of = [&](int g) { return transposed ? (g % d.W) * d.H + g / d.W : g; };
    std::sort(by_g0.begin(), by_g0.end(), [&](int a, int b) { return g0_of(a) < g0_of(b); });
    std::vector<int> rank_of((size_t)d.N, 0);
    for (int r = 0; r < n; ++r) rank_of[by_g0[r]] = r;
    for (int i = 0; i < n; ++i) {
      const int g = order[i], x = g / d.W, y = g % d.W;
      local[g] = i;
      by_rank[(size_t)c0 + rank_of[g]] = i;
      ConvCell &c = cells[(size_t)c0 + i];
      c.gh = g; c.g0 = transposed ? y * d.H + x : g; c.sidx = h->h_state_index[g]; c.pad = rank_of[g]; c.mask = 0;
      for (size_t k = 0; k < odx.size(); ++k) { // (wide: no offset table, no mask)
        const int xx = x + odx[k], yy = y + ody[k];
        if (xx >= 0 && xx < d.H && yy >= 0 && yy < d.W && room[xx * d.W + yy] == z) c.mask |= 1ull << k;
      }
    }
  }
  if (max_room > kConvMaxRoom)
    return fail(SB_ERR_UNSUPPORTED, "sb_convection_attach: a room has more than 2047 cells");
  if (first_building < 0 || (unsigned long long)first_building + (unsigned long long)d.B > (1ull << 32))
    return fail(SB_ERR_INVALID, "sb_convection_attach: the shuffle's streams number buildings modulo 2^32 -- first_building + B reaches beyond");
  if (!wide) // every cell's partner list: the targets of its valid offsets, in table order, as list indices
    for (int z = 0; z < d.Z; ++z) {
      const int c0 = h->h_zone_off[z], n = h->h_zone_off[z + 1] - c0;
      for (int i = 0; i < n; ++i) {
        const ConvCell &c = cells[(size_t)c0 + i];
        int cnt = 0;
        for (size_t k = 0; k < offd.size(); ++k)
          if ((c.mask >> k) & 1ull) partner[((size_t)c0 + i) * pw + cnt++] = (unsigned short)local[c.gh + offd[k]];
      }
    }
  {
    std::vector<short> room16(room.size());
    for (size_t i = 0; i < room.size(); ++i) room16[i] = (short)room[i];
    if (h->conv_room.p) { (void)hipFree(h->conv_room.p); h->conv_room.p = nullptr; }
    const int rc16 = upload(h->conv_room, room16.data(), room16.size());
    if (rc16 != SB_OK) return rc16;
  }
  h->conv_wide = wide; h->conv_wide_d = distance; h->conv_transposed = transposed != 0;
  for (DevBuf<int> *buf : {&h->conv_local, &h->conv_off, &h->conv_by_rank}) {
    if (buf->p) (void)hipFree(buf->p); // attached before: replace
    buf->p = nullptr;
  }
  if (h->conv_cells.p) { (void)hipFree(h->conv_cells.p); h->conv_cells.p = nullptr; }
  int rc;
  if ((rc = upload(h->conv_local, local.data(), local.size())) != SB_OK) return rc;
  if ((rc = upload(h->conv_off, offd.data(), offd.size())) != SB_OK) return rc;
  if ((rc = upload(h->conv_cells, cells.data(), cells.size())) != SB_OK) return rc;
  if ((rc = upload(h->conv_by_rank, by_rank.data(), by_rank.size())) != SB_OK) return rc;
  if (h->conv_partner.p) { (void)hipFree(h->conv_partner.p); h->conv_partner.p = nullptr; }
  if ((rc = upload(h->conv_partner, partner.data(), partner.size())) != SB_OK) return rc;
  h->conv_pw = pw;
  h->conv_whole_room = whole_room;
  h->conv_p = p; h->conv_n_off = (int)offd.size(); h->conv_max_room = max_room;
  h->conv_seed = seed; h->conv_first = first_building; h->conv_calls = 0;
  h->conv_attached = true;
  return SB_OK;
}

} // extern "C"

int sb_launch_convection(sb_handle *h, hipStream_t stream) {
  const Dev &d = h->d;
  if (h->conv_whole_room) {
    ConvAllArgs w;
    w.temp = d.temp; w.stride = d.reg ? (size_t)d.state_doubles : (size_t)d.Np;
    w.zone_off = h->zone_off.p; w.by_rank = h->conv_by_rank.p; w.cells = h->conv_cells.p;
    w.B = d.B; w.Z = d.Z; w.seed = h->conv_seed; w.first_building = h->conv_first; w.call = h->conv_calls++;
    hipLaunchKernelGGL(k_convect_all, dim3(std::max(1, std::min(d.B, h->cus * 8))), dim3(256), 0, stream, w);
    SB_HIP(hipGetLastError());
    return SB_OK;
  }
  ConvArgs o;
  o.temp = d.temp; o.stride = d.reg ? (size_t)d.state_doubles : (size_t)d.Np;
  o.zone_off = h->zone_off.p; o.local = h->conv_local.p; o.off = h->conv_off.p; o.cells = h->conv_cells.p;
  o.B = d.B; o.Z = d.Z; o.W = d.W; o.n_off = h->conv_n_off; o.max_room = h->conv_max_room;
  o.p = h->conv_p; o.seed = h->conv_seed; o.first_building = h->conv_first; o.call = h->conv_calls++;
  o.wide = h->conv_wide ? 1 : 0; o.wide_d = h->conv_wide_d; o.wide_r = (int)std::floor(std::sqrt((double)h->conv_wide_d));
  o.H = d.H; o.transposed = h->conv_transposed ? 1 : 0; o.room = h->conv_room.p; o.by_rank = h->conv_by_rank.p;
  o.partner = h->conv_partner.p; o.pw = h->conv_pw;
  const size_t lds = (size_t)o.max_room * (8 + 8 + 4);
  // cells per lane: workgroups of about 256 lanes; the grid is exactly what is resident at once
  // (the runtime's occupancy for this instantiation), so that every workgroup gets the same number
  // of buildings and there is no second round
  int q = std::max(1, (o.max_room + 255) / 256);
  if (const char *e = getenv("SBSIM_DEBUG_CONV_Q")) q = std::max(1, atoi(e)); // developer knob: cells per lane (too few for the largest room: the launch fails below)
  if (q > kConvMaxPerLane) q = kConvMaxPerLane;
  const int threads = ((o.max_room + q - 1) / q + 63) / 64 * 64;
  if (threads > kConvMaxThreads) return fail(SB_ERR_UNSUPPORTED, "convection: room too large for one workgroup");
  auto go = [&](auto kernel) -> int {
    int per_cu = 0;
    SB_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, lds));
    const int blocks = std::max(1, std::min(d.B, h->cus * std::max(per_cu, 1)));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(threads), lds, stream, o);
    return SB_OK;
  };
  int rc;
  switch (q) {
    case 1: rc = go(k_convect<1>); break;
    case 2: rc = go(k_convect<2>); break;
    case 3: rc = go(k_convect<3>); break;
    case 4: rc = go(k_convect<4>); break;
    case 5: rc = go(k_convect<5>); break;
    case 6: rc = go(k_convect<6>); break;
    case 7: rc = go(k_convect<7>); break;
    default: rc = go(k_convect<8>); break;
  }
  if (rc != SB_OK) return rc;
  SB_HIP(hipGetLastError());
  return SB_OK;
}

