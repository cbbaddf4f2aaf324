"""Pins the CPU oracle (oracle/sb_oracle.c) to fixtures produced by the reference itself.

The fixtures in tests/golden were written by oracle/gen_golden.py, which imports and runs
the reference simulator (see that file).  Everything here is float64 and must agree
BIT-EXACTLY with the reference: temperatures, sweep counts, device state.  Quantities
that pass through a proto ``float`` field are compared as fp32 values, exactly equal,
except where the reference evaluates a numpy transcendental (np.exp / np.log) whose
last-ulp behaviour is library-specific: those use a 1-ulp(fp32) tolerance.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.golden_util import load, oracle_params, oracle_plan


def test_pairwise_sum_matches_numpy():
  rs = np.random.RandomState(0)
  for n in [1, 2, 7, 8, 9, 15, 16, 17, 100, 127, 128, 129, 130, 255, 256, 257, 600, 601, 1000,
            4097, 5400, 6664, 8191, 8192, 8193, 10000, 15756, 16385, 40000]:
    a = 280.0 + 30.0 * rs.rand(n)
    assert orc.np_mean(a) == np.mean(a), n
    assert orc.np_mean(a) == np.mean(list(a)), n
    b = a.reshape(1, -1) if n % 2 else a.reshape(-1, 2) if n % 2 == 0 and n > 1 else a
    assert orc.np_mean(np.ascontiguousarray(b)) == b.mean(), n


@pytest.mark.parametrize("name", ["r9_test", "small_test", "weird_test"])
def test_single_sweep_and_fd_timestep_bit_exact(name):
  g = load(f"sweep_{name}.npz")
  plan = oracle_plan(load(f"plan_{name}.npz"))
  est = g["est"].copy()
  md1 = orc.sweep(plan, g["prev"], est, g["q"], float(g["t_amb"]), float(g["h"]), float(g["dt"]))
  assert md1 == float(g["max_delta_1"])
  assert np.array_equal(est, g["est_after_1"])
  md2 = orc.sweep(plan, g["prev"], est, g["q"], float(g["t_amb"]), float(g["h"]), float(g["dt"]))
  assert md2 == float(g["max_delta_2"])
  assert np.array_equal(est, g["est_after_2"])
  grid, n, conv = orc.fd_timestep(plan, g["prev"], g["q"], float(g["t_amb"]), float(g["h"]),
                                  float(g["dt"]), float(g["thr"]), int(g["iter_limit"]))
  assert n == int(g["fd_sweeps"]) and conv == bool(g["fd_converged"])
  assert np.array_equal(grid, g["fd_grid"])


def test_neighbor_counts_match_reference():
  for name in ["r9_test", "small_test", "weird_test", "r9_sb1"]:
    p = load(f"plan_{name}.npz")
    plan = oracle_plan(p)
    assert np.array_equal(plan.nbr_cnt, p["len_neighbors"].astype(np.int32)), name


def _h1(name, plan_name):
  g = load(name)
  plan = oracle_plan(load(plan_name))
  prm = oracle_params(g["params_json"])
  ob = orc.OracleBuilding(plan, prm, float(g["initial_temp"]))
  comfort = g["comfort"]
  T = len(g["n_sweeps"])
  for t in range(T):
    out = ob.step(now_ts=300.0 * t, t_amb_now=float(g["t_amb"]), h_conv=float(g["h_conv"]),
                  t_amb_next=float(g["t_amb"]), comfort_now=bool(comfort[t]),
                  comfort_prev=bool(comfort[t - 1]) if t else False, comfort_next=bool(comfort[t + 1]),
                  occupancy=1.0, e_price=1e-8, e_carbon=1e-8, g_price=1e-8, g_carbon=1e-8,
                  action=None, observe=False)
    assert out["n_sweeps"] == int(g["n_sweeps"][t]), (t, out["n_sweeps"])
    assert np.array_equal(out["zone_temp_post"], g["zone_temp_post"][t]), t
    assert np.array_equal(out["zone_temp_pre"], g["zone_temp_pre"][t]), t
    assert out["blr_return_temp"] == g["blr_return_temp"][t], t
    assert out["ahu_flow"] == g["ahu_flow"][t] and out["blr_flow"] == g["blr_flow"][t]
    assert np.array_equal(out["mode"], g["mode"][t])
    assert np.array_equal(out["damper"], g["damper"][t]) and np.array_equal(out["valve"], g["valve"][t])
    if t == 0:
      assert np.array_equal(ob.grid(), g["grid_1"])
  assert np.array_equal(ob.grid(), g["final_grid"])
  assert np.array_equal(ob.input_q.reshape(plan.H, plan.W), g["final_input_q"])
  return g


def test_reference_kat_return_water_temperature():
  """simulator_flexible_floor_plan_test.py:1275-1312 expects 301.895482 +- 1e-5."""
  g = _h1("h1_r9_test_cold200.npz", "plan_r9_test.npz")
  assert abs(float(g["blr_return_temp"][0]) - 301.895482) < 1e-5
  assert list(g["n_sweeps"]) == [2, 100, 28, 7, 3, 2]


def test_h1_mild():
  _h1("h1_r9_test_mild292.npz", "plan_r9_test.npz")


def test_h1_small_and_weird_plans():
  _h1("h1_small_test.npz", "plan_small_test.npz")
  _h1("h1_weird_test.npz", "plan_weird_test.npz")


def test_rectangular_building_matches_floor_plan_building():
  """simulator_test.py:955-987: the deprecated rectangular Building gives the same KAT;
  its neighbour lists keep every in-bounds cell (building.py:529-545)."""
  g = load("rect_building_cold200.npz")
  H, W = g["conductivity"].shape
  rs, bs = g["room_shape"], g["building_shape"]
  zones = []
  for zx in range(bs[0]):
    for zy in range(bs[1]):
      x0, y0 = zx * (rs[0] + 1) + 2, zy * (rs[1] + 1) + 2  # building.py:167-180
      xs, ys = np.meshgrid(np.arange(x0, x0 + rs[0]), np.arange(y0, y0 + rs[1]), indexing="ij")
      zones.append((xs * W + ys).reshape(-1).astype(np.int32))
  plan = orc.OraclePlan(g["conductivity"], g["density"], g["heat_capacity"],
                        np.zeros((H, W), bool), zones, g["diffusers"], float(g["cv_size_cm"]),
                        float(g["floor_height_cm"]), skip_exterior=False)
  h = load("h1_r9_test_cold200.npz")
  ob = orc.OracleBuilding(plan, oracle_params(h["params_json"]), 200.0)
  out = ob.step(now_ts=0.0, t_amb_now=296.0, h_conv=12.0, t_amb_next=296.0, comfort_now=False,
                comfort_prev=False, comfort_next=False, occupancy=1.0, e_price=1e-8,
                e_carbon=1e-8, g_price=1e-8, g_carbon=1e-8, action=None, observe=False)
  assert out["blr_return_temp"] == float(g["blr_return_temp"])
  assert np.array_equal(ob.grid(), g["final_grid"])


def _ulp32(a, b):
  a = np.asarray(a, np.float32).view(np.int32).astype(np.int64)
  b = np.asarray(b, np.float32).view(np.int32).astype(np.int64)
  return np.abs(a - b)


@pytest.mark.parametrize("tag", ["const", "random"])
def test_h2_one_day_rollout(tag):
  """Config 1 (BASELINE.json configs[0]) and its random-action twin: SB1 physics on R9,
  Environment._step ordering, 288 steps."""
  g = load(f"h2_sb1_r9_{tag}.npz")
  plan = oracle_plan(load("plan_r9_sb1.npz"))
  prm = oracle_params(g["params_json"])
  ob = orc.OracleBuilding(plan, prm, float(g["initial_temp"]))
  T = len(g["n_sweeps"])
  cum = np.zeros(4)
  for t in range(T):
    out = ob.step(
        now_ts=float(g["ts_seconds"][t]), t_amb_now=float(g["t_amb_now"][t]),
        h_conv=float(g["h_conv"]), t_amb_next=float(g["t_amb_next"][t]),
        comfort_now=bool(g["comfort_now"][t]), comfort_prev=g["comfort_prev"][t] == 1,
        comfort_next=bool(g["comfort_next"][t]), occupancy=float(g["occupancy"][t]),
        e_price=float(g["e_price"][t]), e_carbon=float(g["e_carbon"][t]),
        g_price=float(g["g_price"][t]), g_carbon=float(g["g_carbon"][t]),
        action=g["action_native"][t], observe=True)
    assert out["n_sweeps"] == int(g["n_sweeps"][t]), t
    assert np.array_equal(out["zone_temp_pre"], g["zone_temp_pre"][t]), t
    assert np.array_equal(out["zone_temp_post"], g["zone_temp_post"][t]), t
    assert out["recirc_pre"] == g["recirc_pre"][t] and out["t_supply_air"] == g["t_supply_air"][t]
    assert out["ahu_flow"] == g["ahu_flow"][t] and out["ahu_count"] == g["ahu_count"][t]
    assert out["blr_flow"] == g["blr_flow"][t] and out["blr_count"] == g["blr_count"][t]
    assert out["blr_return_temp"] == g["blr_return_temp"][t], t
    assert out["blr_tank_temp"] == g["blr_tank_temp"][t], t
    assert np.array_equal(out["mode"], g["mode"][t]), t
    rates = np.array([out["blower_rate"], out["ac_rate"], out["gas_rate"], out["pump_rate"]], np.float32)
    # gas rate contains np.log (boiler.py:316): allow 1 ulp of fp32
    assert _ulp32(rates, g["rates"][t]).max() <= 1, (t, rates, g["rates"][t])
    assert _ulp32(out["reward"], g["reward"][t]) <= 2, (t, out["reward"], g["reward"][t])
    assert abs(out["norm_prod_regret"] - g["rr_norm_prod_regret"][t]) < 2e-7
    assert abs(out["norm_energy_cost"] - g["rr_norm_energy_cost"][t]) < 2e-7
    assert abs(out["norm_carbon"] - g["rr_norm_carbon"][t]) < 2e-7
    qsum = np.array([ob.input_q[z].sum() for z in plan.zones])
    assert np.allclose(qsum, g["zone_q_sum"][t], rtol=1e-12, atol=1e-9)
    cum += rates.astype(np.float64) * prm.dt
    if t + 1 in (1, 144):
      assert np.array_equal(ob.grid(), g[f"grid_{t + 1}"]), t
  assert np.array_equal(ob.grid(), g["final_grid"])
  ref_cum = g["rates"].astype(np.float64).sum(axis=0) * prm.dt
  assert np.allclose(cum, ref_cum, rtol=1e-6)


def test_reward_kat_rows():
  """The reference's own reward vectors: setpoint_energy_carbon_regret_test.py:30-167
  (TestEnergyCost(0.05 USD/kWh, 0.01 kg/kWh) for both fuels, two identical zones)."""
  import json, os
  from tests.golden_util import GOLDEN
  with open(os.path.join(GOLDEN, "reward_kat.json")) as fh:
    kat = json.load(fh)
  c = kat["config"]
  price, carbon = c["usd_per_kwh"] / 3600.0 / 1000.0, c["kg_per_kwh"] / 3600.0 / 1000.0
  for row in kat["rows"]:
    prm = orc.OracleParams(
        dt=300.0, conv_threshold=0.1, iter_limit=1, vav_max_air_flow=1, vav_max_water_flow=1,
        ahu_recirc=0, ahu_heat_sp=0, ahu_cool_sp=1, ahu_dp=1, ahu_eff=1, blr_setpoint=1, blr_head=1,
        blr_pump_eff=1, comfort_lo=0, comfort_hi=1, eco_lo=0, eco_hi=1,
        max_prod=c["max_productivity_personhour_usd"], min_prod=c["min_productivity_personhour_usd"],
        max_elec=c["max_electricity_rate"], max_gas=c["max_natural_gas_rate"],
        prod_delta=c["productivity_midpoint_delta"], prod_stiff=c["productivity_decay_stiffness"],
        w_prod=row["productivity_weight"], w_cost=row["energy_cost_weight"],
        w_carbon=row["carbon_emission_weight"])
    r, diag = orc.reward(
        prm, [row["zone_air_temperature"]] * 2, c["heating_setpoint"], c["cooling_setpoint"],
        row["average_occupancy"], row["blower"], row["air_conditioning"], row["natural_gas"],
        row["pump"], 300.0, price, carbon, price, carbon)
    assert abs(r - row["expected_reward"]) < 5e-5, row["name"]
    assert abs(diag[0] - row["expected_productivity"]) < 5e-5, row["name"]
