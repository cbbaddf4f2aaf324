"""Stochastic convection shuffle: the device kernel's CPU restatement
against the reference's displacement statistics, and the device against the restatement."""
import numpy as np
import pytest

from oracle.convection_oracle import ConvectionOracle, offsets
from tests.golden_util import load


def _stats(grids, H, W):
  B = grids.shape[0]
  src = grids.astype(np.int64)
  sx, sy = np.divmod(src, W)
  x, y = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
  dx, dy = x[None] - sx, y[None] - sy
  room = np.zeros((H, W), bool)
  room[1:H - 1, 1:W - 1] = True
  inner = room[None] & (sx >= 5) & (sx <= H - 6) & (sy >= 5) & (sy <= W - 6)
  hist = np.zeros((17, 17))
  near = inner & (np.abs(dx) <= 8) & (np.abs(dy) <= 8)   # further moves (whole-room shuffle) stay outside the window
  np.add.at(hist, (dx[near] + 8, dy[near] + 8), 1)
  n_room = B * room.sum()
  fixed = ((dx == 0) & (dy == 0) & room[None]).sum() / n_room
  msd = ((dx * dx + dy * dy) * room[None]).sum() / n_room
  return hist / inner.sum(), fixed, msd


KAT_STREAM = (398727278, 1462956438)               # conv_stream(4242, 1000, 0): two 32-bit words since round 5
KAT_WORDS = [2148191377, 7270220, 921932424]     # conv_word(conv_stream(4242, 1000, 0), 777, 0..2)


def test_mixer_known_answers():
  """The shuffle's generator (generators.hip fmix32 / conv_stream / conv_word): MurmurHash3's finaliser --
  its published test values -- and the composition the kernel uses, pinned here so that the device, which is
  compared with this restatement cell for cell, cannot drift with it."""
  from oracle.convection_oracle import conv_stream, conv_word, fmix32
  assert fmix32(0) == 0 and fmix32(1) == 0x514E28B7 and fmix32(0xFFFFFFFF) == 0x81F16F39   # MurmurHash3 fmix32
  s = conv_stream(4242, 1000, 0)
  assert s == conv_stream(4242, 1000, 0) and s != conv_stream(4242, 1001, 0) and s != conv_stream(4242, 1000, 1)
  assert conv_stream(0x1_0000_0000 + 4242, 1000, 0) != s                                # the seed's high word counts
  assert s == KAT_STREAM and [conv_word(s, 777, k) for k in range(3)] == KAT_WORDS
  # a 32-bit stream gives a 65,536-building batch ~0.5 colliding pairs per call (whole buildings making identical
  # draws); with both words no two buildings of the batch share a stream -- here for three calls, and with a shard offset
  for call, first in ((0, 0), (1, 0), (287, 65536 * 7)):
    st = [conv_stream(1234, first + b, call) for b in range(65536)]
    assert len(set(st)) == 65536, (call, first)
  assert len({a for a, _ in [conv_stream(1234, b, 0) for b in range(65536)]}) <= 65536   # (the first word alone may collide)
  # uniformity of the top bits over cells and words (a smoke test, not a battery): 4,096 draws into 16 bins
  draws = np.array([conv_word(s, g, k) >> 28 for g in range(1024) for k in range(4)])
  assert np.abs(np.bincount(draws, minlength=16) - 256).max() < 70


def test_offset_window_is_the_reference_one():
  """stochastic_convection_simulator.py:125-131 with distance = 5: squared distance <= 5."""
  off = offsets(5)
  assert len(off) == 21 and (0, 0) in off and (2, 1) in off and (-1, -2) in off and (2, 2) not in off
  assert off == sorted(off)                       # (dx, dy) raster order = the reference's loop order
  assert len(offsets(2)) == 9 and offsets(1) == [(-1, 0), (0, -1), (0, 0)]    # the window is half-open: [-d, d)
  # from distance 20 on the window has more than 64 offsets (the device then samples by rejection); the
  # half-open box no longer cuts the disc: floor(sqrt(d)) < d
  assert len(offsets(19)) == 61 and len(offsets(20)) == 69
  assert all(max(abs(dx), abs(dy)) <= 4 for dx, dy in offsets(20)) and len(offsets(1000)) == 3149


def test_partner_pick_bias_bound():
  """since round 5 the partner is picked from the LOW 12 bits of the word whose top 20 bits are the swap's time
  stamp: `pick = ((w & 0xfff) * cnt) >> 12`.  For a count that does not divide 4,096 the picks are not exactly uniform (the
  reference's `random.choice` is): a partner gets floor(4096 / cnt) or ceil(4096 / cnt) of the 4,096 values -- a relative
  spread of at most 1 / floor(4096 / cnt), i.e. <= 1.6 % for the largest offset table (61 offsets at distance 19; 21 at SB1's
  distance 5: 0.52 %), far inside what the displacement statistics below resolve.  Pinned here so that a change of the pick's
  width is a conscious one."""
  for cnt in (1, 2, 3, 9, 21, 49, 61, 64):
    picks = (np.arange(4096, dtype=np.int64) * cnt) >> 12
    hist = np.bincount(picks, minlength=cnt)
    assert hist.min() == 4096 // cnt and hist.max() <= 4096 // cnt + 1 and picks.max() == cnt - 1
    assert (hist.max() - hist.min()) / hist.min() <= 1.0 / (4096 // cnt)
  assert 1.0 / (4096 // 61) < 0.016 and 1.0 / (4096 // 21) < 0.0052


@pytest.mark.parametrize("case", [0, 1, 2, 3, 4, 5])
def test_restatement_matches_reference_displacement_statistics(case):
  """24,000 tracked values on each side.  Bin probabilities <= 0.25 -> sigma of a difference
  <= sqrt(2 * 0.25 * 0.75 / 24000) = 0.004; 0.02 is five sigma."""
  g = load("convection_stats.npz")
  H, W = int(g["H"]), int(g["W"])
  p, dist = float(g["cases"][case][0]), int(g["cases"][case][1])
  room = [x * W + y for x in range(1, H - 1) for y in range(1, W - 1)]
  B = 400
  m = ConvectionOracle([room], H, W, p, dist, seed=77 + case)
  grids = np.tile(np.arange(H * W, dtype=np.float64).reshape(1, H, W), (B, 1, 1))
  m.apply(grids)
  assert np.array_equal(np.sort(grids.reshape(B, -1), axis=1), np.tile(np.arange(H * W), (B, 1)))   # permutations
  assert np.array_equal(grids[:, 0, :], np.tile(np.arange(W), (B, 1)))                              # walls stay
  hist, fixed, msd = _stats(grids, H, W)
  assert np.abs(hist - g[f"hist_{case}"]).max() < 0.02
  assert abs(fixed - float(g[f"fixed_{case}"])) < 0.01
  assert abs(msd - float(g[f"msd_{case}"])) < 0.03 * float(g[f"msd_{case}"])


@pytest.mark.parametrize("case", [0, 1])
def test_small_rooms_under_a_wide_window_mix_like_the_reference(case):
  """Rooms of 9, 60 and 100 cells under windows of more than 64 offsets (distance = -1 with p < 1 = the
  reference's 1000-cell window; distance 30).  The reference always draws the partner from the room's own
  candidate list (stochastic_convection_simulator.py:108-131), so a 9-cell room mixes as thoroughly as a
  large one: the restatement draws a cell of the room by rank and keeps it when it lies inside the disc.
  (Round 3's box sampling with a 16-block cap left 85 % of a 9-cell room's swaps undone: fixed fraction 0.9
  instead of 0.36.)  800 runs on each side: fixed fraction within 0.03, mean squared displacement within 6 %."""
  g = load("convection_stats.npz")
  H, W = (int(v) for v in g["small_shape"])
  p, dist = float(g["small_cases"][case][0]), int(g["small_cases"][case][1])
  rooms = [[x * W + y for x in range(x0, x0 + h) for y in range(y0, y0 + w)] for x0, y0, h, w in g["small_rooms"]]
  B = 800
  m = ConvectionOracle(rooms, H, W, p, dist, seed=91 + case)
  grids = np.tile(np.arange(H * W, dtype=np.float64).reshape(1, H, W), (B, 1, 1))
  m.apply(grids)
  flat = grids.reshape(B, -1).astype(np.int64)
  for ri, cells in enumerate(rooms):
    cells = np.asarray(cells)
    src = flat[:, cells]
    assert np.array_equal(np.sort(src, axis=1), np.tile(np.sort(cells), (B, 1)))     # values stay in their room
    sx, sy = np.divmod(src, W)
    x, y = np.divmod(cells, W)
    fixed = ((sx == x[None]) & (sy == y[None])).mean()
    msd = ((sx - x[None]) ** 2 + (sy - y[None]) ** 2).mean()
    assert abs(fixed - float(g[f"small_fixed_{case}"][ri])) < 0.03, (ri, fixed, g[f"small_fixed_{case}"][ri])
    assert abs(msd - float(g[f"small_msd_{case}"][ri])) < 0.06 * float(g[f"small_msd_{case}"][ri]), (ri, msd)


def test_restatement_does_not_depend_on_sharding_and_calls_differ():
  H, W = 10, 12
  rooms = [[x * W + y for x in range(1, 5) for y in range(1, 11)], [x * W + y for x in range(6, 9) for y in range(1, 11)]]
  base = np.arange(H * W, dtype=np.float64).reshape(1, H, W)
  whole = ConvectionOracle(rooms, H, W, 1.0, 5, seed=3)
  a = np.tile(base, (6, 1, 1)); whole.apply(a)
  parts = []
  for f in (0, 3):
    m = ConvectionOracle(rooms, H, W, 1.0, 5, seed=3, first_building=f)
    gpart = np.tile(base, (3, 1, 1)); m.apply(gpart); parts.append(gpart)
  assert np.array_equal(a, np.concatenate(parts))
  assert not np.array_equal(a[0], a[1])
  b = np.tile(base, (6, 1, 1)); whole.apply(b)        # second call: other draws
  assert not np.array_equal(a, b)
  for z, cells in enumerate(rooms):                   # values never leave their room
    assert np.array_equal(np.sort(a[:, :, :].reshape(6, -1)[:, cells], axis=1), np.tile(np.sort(cells), (6, 1)))


# ------------------------------------------------------------------------------------- GPU
def _need_gpu():
  import torch
  if not torch.cuda.is_available():
    pytest.skip("needs an MI355X")


@pytest.mark.gpu
@pytest.mark.parametrize("kernel,distance,prob", [("reg", 5, 1.0), ("lds", 5, 1.0), ("reg", -1, 1.0), ("lds-columns", -1, 1.0),
                                                  ("reg", -1, 0.5), ("lds-columns", -1, 0.5), ("reg", 20, 1.0), ("lds-columns", 20, 0.7)])
def test_device_shuffle_equals_its_restatement_inside_a_rollout(kernel, distance, prob, monkeypatch):
  """R9, SB1 physics, convection p = 1 / distance = 5 (sim_config.gin:36-39) -- or the whole-room
  shuffle, distance = -1 (stochastic_convection_simulator.py:78-99); or a window of more than 64 offsets
  (distance 20; distance = -1 with p < 1 = the reference's 1000-cell window, :108-109), where the device
  samples the partner by rejection -- after every FD update: sweep
  counts, zone temperatures and the final grid against oracle twins whose grids get the
  restatement's shuffle after every step."""
  _need_gpu()
  import torch
  from oracle import oracle as orc
  from sbsim_amd import _ffi
  from sbsim_amd.environment import BatchedSimulator, SimConfig
  from tests.golden_util import oracle_params, oracle_plan
  from tests.test_gpu_parity import T_TOL, _plan, _step_in
  if kernel.startswith("lds"):
    monkeypatch.setenv("SBSIM_FORCE_LDS_PATH", "1")
  g = load("h2_sb1_r9_random.npz")
  p = load("plan_r9_sb1.npz")
  B, T, first = 4, 10, 1000
  rs = np.random.RandomState(5)
  init = np.clip(294.0 + rs.randn(B, 1, 1) + 0.8 * rs.randn(B, 68, 98), 285.0, 305.0)
  acts = rs.uniform(-1, 1, size=(T, B, 2)).astype(np.float32)
  fp = _plan(p)
  sim = BatchedSimulator(fp, SimConfig.sb1(), B, float(g["h_conv"]),
                         orientation="columns" if kernel == "lds-columns" else "auto")
  assert sim.launch_info["path"] == (1 if kernel == "reg" else 0)
  assert sim.transposed or kernel != "lds-columns"     # the permutation is defined on the CALLER's raster order
  sim.convection_attach(prob, distance, seed=4242, first_building=first)
  sim.reset(temps=torch.tensor(init.reshape(B, -1), dtype=torch.float64, device="cuda"))
  conv = ConvectionOracle(fp.zone_cell_lists(), 68, 98, prob, distance, seed=4242, first_building=first)
  plan, prm = oracle_plan(p), oracle_params(g["params_json"])
  twins = [orc.OracleBuilding(plan, prm, 0.0, reset_temps=init[b].reshape(-1)) for b in range(B)]
  obs = torch.zeros((B, sim.O), dtype=torch.float32, device="cuda")
  rew = torch.zeros((B,), dtype=torch.float32, device="cuda")
  info = torch.zeros((B, _ffi.SB_INFO_STRIDE), dtype=torch.float32, device="cuda")
  for t in range(T):
    sim.step(torch.tensor(acts[t], device="cuda"), _step_in(g, t + 100), obs, rew, info)
    i = info.cpu().numpy().astype(np.float64)
    zt = sim.zone_temps().cpu().numpy()
    for b in range(B):
      a = acts[t, b]
      native = [np.float32((float(a[0]) + 1.0) / 2.0 * 45.0 + 310.0), np.float32((float(a[1]) + 1.0) / 2.0 * 15.0 + 285.0)]
      tt = t + 100
      o = twins[b].step(
          now_ts=300.0 * t, t_amb_now=float(g["t_amb_now"][tt]), h_conv=float(g["h_conv"]),
          t_amb_next=float(g["t_amb_next"][tt]), comfort_now=bool(g["comfort_now"][tt]),
          comfort_prev=g["comfort_prev"][tt] == 1, comfort_next=bool(g["comfort_next"][tt]),
          occupancy=float(g["occupancy"][tt]), e_price=float(g["e_price"][tt]),
          e_carbon=float(g["e_carbon"][tt]), g_price=float(g["g_price"][tt]),
          g_carbon=float(g["g_carbon"][tt]), action=native, observe=True)
      assert i[b, 4] == o["n_sweeps"], (t, b, i[b, 4], o["n_sweeps"])
      assert np.abs(zt[b] - o["zone_temp_post"]).max() < T_TOL, (t, b)
    grids = np.stack([tw.grid() for tw in twins])
    before = grids.copy()
    conv.apply(grids)                                   # apply_convection after the FD update
    assert not np.array_equal(before, grids)
    for b in range(B):
      twins[b].temp[:] = grids[b].reshape(-1)
    dev = sim.temps().cpu().numpy()
    for b in range(B):
      assert np.abs(dev[b] - grids[b]).max() < T_TOL, (t, b)
  sim.close()


@pytest.mark.gpu
def test_convection_attach_argument_checks():
  _need_gpu()
  from sbsim_amd import _ffi
  from sbsim_amd.environment import BatchedSimulator, SimConfig
  from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan
  fp = FloorPlan.from_file_input(rectangular_floor_plan((2, 2), (5, 9)), Materials.sb1(), 10.0, 300.0)
  sim = BatchedSimulator(fp, SimConfig.sb1(), 3, 100.0)
  sim.convection_attach(0.0, 5, seed=1)       # p == 0: the reference returns early -> detached
  sim.convection_attach(1.0, 0, seed=1)
  sim.convection_attach(1.0, -1, seed=1)      # the whole-room shuffle (stochastic_convection_simulator.py:78)
  sim.convection_attach(0.5, -1, seed=1)      # distance = -1 with p < 1: the reference's 1000-cell window (:108-109)
  with pytest.raises(_ffi.SbsimError, match="distance must be"):
    sim.convection_attach(0.5, 1001, seed=1)
  with pytest.raises(_ffi.SbsimError, match=r"p must be in \[0, 1\]"):
    sim.convection_attach(1.5, 5, seed=1)
  sim.close()


def test_seeded_host_convection_replays_the_reference_draw_for_draw():
  """host_convection.py against the reference's own seeded shuffles (oracle/gen_golden_convection_seeded.py ran
  `StochasticConvectionSimulator(p, distance, seed).apply_convection` three times per case): the array after every
  call is IDENTICAL -- rooms in room-dict order (first appearance in raster order), the windowed shuffle, the
  whole-room permutation, distance -1 with p < 1 and a window wider than the rooms."""
  from sbsim_amd.host_convection import SeededHostConvection
  g = load("convection_seeded.npz")
  H, W = int(g["H"]), int(g["W"])
  rooms = [[(x, y) for x in range(x0, x0 + h) for y in range(y0, y0 + w)] for x0, y0, h, w in g["rooms"]]
  rooms.sort(key=lambda cells: cells[0])     # building_utils.py:406-414: keys in order of first appearance
  for ci, (p, dist, seed) in enumerate(g["cases"]):
    sim = SeededHostConvection(p, int(dist), int(seed))
    temp = np.arange(H * W, dtype=np.float64).reshape(H, W)
    for k in range(g["after"].shape[1]):
      sim.apply(rooms, temp)
      assert np.array_equal(temp.astype(np.int32), g["after"][ci, k]), (ci, k)
  quiet = SeededHostConvection(0.0, 5, 1)    # p == 0 / distance == 0: the reference returns early (:68-70)
  temp = np.arange(H * W, dtype=np.float64).reshape(H, W)
  quiet.apply(rooms, temp)
  assert np.array_equal(temp, np.arange(H * W, dtype=np.float64).reshape(H, W))


@pytest.mark.gpu
@pytest.mark.parametrize("plan_rooms,room_shape,force_lds", [((3, 3), (20, 30), False), ((2, 2), (5, 9), False), ((8, 5), (12, 14), False),
                                                            ((9, 4), (16, 17), False), ((3, 3), (20, 30), True)])
def test_single_building_adapter_with_reproducible_convection(plan_rooms, room_shape, force_lds, monkeypatch):
  """HipSimulatorBuilding(convection_simulator=..., reproducible_convection=True): after a step the device grid is the
  finite-difference update's grid with the reference's seeded shuffle applied -- bit for bit (a twin adapter without
  convection gives the grid before the shuffle) -- the zone means follow it, every device state stays, and the next
  steps run from it.  sb_set_temps on its own: what goes in comes out, on five state layouts (k_sweep_roll, k_sweep_reg,
  k_sweep_two, k_sweep_band on three wavefronts, the LDS-grid kernel)."""
  _need_gpu()
  if force_lds:
    monkeypatch.setenv("SBSIM_FORCE_LDS_PATH", "1")
  import torch
  from sbsim_amd import building_adapter as ba
  from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan
  from sbsim_amd.host_convection import SeededHostConvection
  from sbsim_amd.host_inputs import StochasticConvectionSimulator
  plan = FloorPlan.from_file_input(rectangular_floor_plan(plan_rooms, room_shape), Materials.sb1(), 10.0, 300.0)
  conv = StochasticConvectionSimulator(1.0, 5, seed=17)
  a = ba.HipSimulatorBuilding(plan, holiday_calendar=None, convection_simulator=conv, reproducible_convection=True)
  b = ba.HipSimulatorBuilding(plan, holiday_calendar=None)
  rooms = [[(int(x), int(y)) for x, y in zip(*np.unravel_index(cells, plan.shape))] for cells in plan.zone_cell_lists()]
  twin = SeededHostConvection(1.0, 5, 17)
  # a non-uniform start, the same in both
  rs = np.random.RandomState(3)
  start = a.env.sim.temps()
  inside = torch.tensor(~np.asarray(plan.exterior_space, dtype=bool), device="cuda")
  start[0][inside] = torch.tensor(293.0 + rs.rand(int(inside.sum())), dtype=torch.float64, device="cuda")   # (exterior-space cells keep their values)
  for bld in (a, b):
    bld.env.sim.set_temps(start)
    assert torch.equal(bld.env.sim.temps(), start)
  a.wait_time()
  b.wait_time()
  before = b.env.sim.temps()[0].cpu().numpy()
  expect = before.copy()
  twin.apply(rooms, expect)
  got = a.env.sim.temps()[0].cpu().numpy()
  assert np.array_equal(got, expect) and not np.array_equal(got, before)
  zt = a.env.sim.zone_temps()[0].cpu().numpy()
  means = np.array([got.reshape(-1)[cells].mean() for cells in plan.zone_cell_lists()])
  assert np.abs(zt - means).max() < 1e-11
  sa, sb_ = a.env.sim.scalars().cpu().numpy(), b.env.sim.scalars().cpu().numpy()
  keep = [i for i in range(sa.shape[1]) if i != 11]
  assert np.array_equal(sa[:, keep], sb_[:, keep])                       # AHU / boiler state untouched by the shuffle ...
  # ... but for the cached grid mean (the next step's recirculation temperature, simulator.py:423-426): sb_set_temps takes it
  # from the NEW grid -- the shuffle keeps the sum, not the order it is added in
  assert abs(sa[0, 11] - got.mean()) < 1e-11 and abs(sa[0, 11] - sb_[0, 11]) < 1e-11
  for _ in range(3):   # the next steps start from the shuffled grid, the draws go on where the last call stopped
    a.wait_time()
  assert np.isfinite(a.env.sim.temps().cpu().numpy()).all()
  with pytest.raises(ValueError, match="n_replicas must be 1"):
    ba.HipSimulatorBuilding(plan, n_replicas=2, holiday_calendar=None, convection_simulator=conv, reproducible_convection=True)
  a.close()
  b.close()
