"""RandomizedArrivalDepartureOccupancy: the host mirror against the
reference's own call-by-call trace, the device generator's CPU restatement against published
Philox known answers and the reference's statistics, and the device against the restatement."""
import datetime as dt

import numpy as np
import pytest

from oracle.occupancy_oracle import DeviceOccupancyOracle, philox4x32_10
from sbsim_amd import host_inputs
from tests.golden_util import load

UTC = dt.timezone.utc


def _args(g):
  return dict(zip([str(k) for k in g["args_keys"]], [int(v) for v in g["args_vals"]]))


@pytest.mark.parametrize("tz_name,tz", [("utc", "UTC"), ("pacific", "US/Pacific")])
def test_host_mirror_reproduces_the_reference_trace(tz_name, tz):
  """Same RandomState stream, same call order (num_occupants' query after the reward's in this
  fixture), three days incl. a Saturday: bit-identical counts."""
  g = load("occupancy_randomized.npz")
  m = host_inputs.RandomizedArrivalDepartureOccupancy(seed=17321, time_zone=tz, holiday_calendar=None, **_args(g))
  t0 = dt.datetime(2023, 7, 6, tzinfo=UTC)
  zones = ["zone_%d" % i for i in range(3)]
  for s in range(864):
    ts = t0 + dt.timedelta(seconds=300 * s)
    r = [m.average_zone_occupancy(z, ts, ts + dt.timedelta(seconds=300)) for z in zones]
    n = int(sum(m.average_zone_occupancy(z, ts - dt.timedelta(minutes=5), ts) for z in zones))
    assert r == list(g["trace_reward_" + tz_name][s]), s
    assert n == g["trace_num_occupants_" + tz_name][s], s
  if tz_name == "utc":
    assert g["trace_reward_utc"][576:].max() == 0     # Saturday
  assert g["trace_reward_" + tz_name][:288].max() == 10


def test_philox_known_answers():
  """Random123 kat_vectors for philox4x32-10."""
  kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
         ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
         ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
          (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
  for ctr, key, want in kat:
    got = philox4x32_10(*[[c] for c in ctr], key[0], key[1])
    assert tuple(int(v[0]) for v in got) == want


def test_device_generator_statistics_match_the_reference():
  """Mean occupancy fraction per step of a working day: 4,000 occupants of the counter-based
  generator against 4,000 of the reference (400 seeds x 10).  Both are sample means of
  Bernoulli chains: sigma of the difference <= sqrt(2 * 0.25 / 4000) = 0.011."""
  g = load("occupancy_randomized.npz")
  a = _args(g)
  ref = g["mean_fraction_one_query_per_step"]
  m = DeviceOccupancyOracle(400, 1, a["zone_assignment"], a["earliest_expected_arrival_hour"],
                            a["latest_expected_arrival_hour"], a["earliest_expected_departure_hour"],
                            a["latest_expected_departure_hour"], float(a["time_step_sec"]), seed=99)
  mean = np.array([m.peek((s * 300) // 3600, True).sum() / 4000.0 for s in range(288)])
  assert np.abs(mean - ref).max() < 0.05
  assert abs(mean.mean() - ref.mean()) < 0.005
  assert mean[:36].max() == 0 and ref[:36].max() == 0          # nobody before the arrival window opens


def test_device_generator_does_not_depend_on_sharding():
  whole = DeviceOccupancyOracle(12, 3, 10, 3, 12, 13, 23, 300.0, seed=5)
  parts = [DeviceOccupancyOracle(6, 3, 10, 3, 12, 13, 23, 300.0, seed=5, first_building=f) for f in (0, 6)]
  for s in range(0, 288, 7):
    w = whole.peek((s * 300) // 3600, True)
    p = np.concatenate([q.peek((s * 300) // 3600, True) for q in parts])
    assert np.array_equal(w, p)
  assert whole.peek(12, True).max() > 0
  assert whole.peek(12, False).max() == 0     # holiday / weekend: everybody AWAY


# ------------------------------------------------------------------------------------- GPU
def _need_gpu():
  import torch
  if not torch.cuda.is_available():
    pytest.skip("needs an MI355X")


@pytest.mark.gpu
def test_device_occupancy_equals_its_restatement():
  _need_gpu()
  import torch
  from sbsim_amd.environment import BatchedSimulator, SimConfig
  from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan
  fp = FloorPlan.from_file_input(rectangular_floor_plan((2, 2), (5, 9)), Materials.sb1(), 10.0, 300.0)
  B = 300
  sim = BatchedSimulator(fp, SimConfig.sb1(), B, 100.0)
  for n_occ, first in ((10, 0), (32, 1 << 33), (1, 7)):
    sim.occupancy_attach(n_occ, (3, 12, 13, 23), 300.0, seed=0x1234567890abcdef, first_building=first)
    orc = DeviceOccupancyOracle(B, sim.Z, n_occ, 3, 12, 13, 23, 300.0, seed=0x1234567890abcdef, first_building=first)
    count = torch.zeros((B, sim.Z), dtype=torch.float32, device="cuda")
    total = torch.zeros((B,), dtype=torch.float32, device="cuda")
    seen = 0.0
    for s in range(0, 2 * 288, 3):
      hour, workday = ((s % 288) * 300) // 3600, s < 288 or (s % 5) != 0
      sim.occupancy_peek(hour, workday, count, total)
      want = orc.peek(hour, workday)
      assert np.array_equal(count.cpu().numpy().astype(np.float64), want), (n_occ, s)
      assert np.array_equal(total.cpu().numpy().astype(np.float64), want.sum(axis=1)), (n_occ, s)
      seen = max(seen, want.max())
    assert seen >= max(1, n_occ // 2)
  sim.close()


@pytest.mark.gpu
def test_environment_with_per_building_randomized_occupancy():
  """Per-building occupancy reaches the reward (productivity term) and the num_occupants feature;
  every building against its own oracle twin fed with the restatement's occupancy."""
  _need_gpu()
  import torch
  from oracle import oracle as orc
  from sbsim_amd.environment import BatchedEnvironment, SimConfig
  from tests.golden_util import oracle_params, oracle_plan
  from tests.test_gpu_parity import _plan
  g = load("h2_sb1_r9_random.npz")
  p = load("plan_r9_sb1.npz")
  B, T = 6, 40
  start = dt.datetime(2023, 7, 6, 8, 0, 0)
  occ = host_inputs.BatchedRandomizedArrivalDepartureOccupancy(10, 3, 12, 13, 23, 300, seed=77, holiday_calendar=None)
  env = BatchedEnvironment(_plan(p), B, config=SimConfig.sb1(), occupancy=occ, start_timestamp=start,
                           occupancy_normalization_constant=125.0, holiday_calendar=None, collect_info=True)
  ts0 = env.reset()
  model = DeviceOccupancyOracle(B, env.sim.Z, 10, 3, 12, 13, 23, 300.0, seed=77)
  col = env.sim.O - 1                                     # num_occupants is the last feature
  n0 = model.peek((start - dt.timedelta(minutes=5)).hour, True).sum(axis=1)
  assert np.allclose(ts0.observation[:, col].cpu().numpy(), ((n0 - 125.0) / 126.0).astype(np.float32))
  plan, prm = oracle_plan(p), oracle_params(g["params_json"])
  twins = [orc.OracleBuilding(plan, prm, float(env.config.initial_temp)) for _ in range(B)]
  rs = np.random.RandomState(2)
  now = start
  differ = False
  for t in range(T):
    act = rs.uniform(-1, 1, size=(B, 2)).astype(np.float32)
    si = env.make_step_in(now)          # what env.step() builds (it advances the device occupants:
    env._needs_reset = False            # rebuild the same inputs for the twins from the restatement)
    nxt = now + dt.timedelta(seconds=300)
    n_obs = model.peek((nxt - dt.timedelta(minutes=5)).hour, True).sum(axis=1)
    occ_rew = model.peek(nxt.hour, True)
    env.sim.step(torch.tensor(act, device="cuda"), si, env._obs, env._reward, env._info)
    r = env._reward.cpu().numpy()
    obs = env._obs.cpu().numpy()
    assert np.allclose(obs[:, col], ((n_obs - 125.0) / 126.0).astype(np.float32)), t
    for b in range(B):
      native = [np.float32((float(act[b, 0]) + 1.0) / 2.0 * 45.0 + 310.0),
                np.float32((float(act[b, 1]) + 1.0) / 2.0 * 15.0 + 285.0)]
      o = twins[b].step(
          now_ts=300.0 * t, t_amb_now=si.t_amb_now, h_conv=float(g["h_conv"]), t_amb_next=si.t_amb_next,
          comfort_now=bool(si.comfort_now), comfort_prev=(si.comfort_prev == 1), comfort_next=bool(si.comfort_next),
          occupancy=occ_rew[b], e_price=si.e_price, e_carbon=si.e_carbon, g_price=si.g_price,
          g_carbon=si.g_carbon, action=native, observe=True)
      assert abs(float(r[b]) - o["reward"]) < 1e-6, (t, b, r[b], o["reward"])
    differ = differ or len(set(occ_rew.sum(axis=1))) > 1
    now = nxt
  assert differ                         # the buildings really had different occupancies
  env.close()
