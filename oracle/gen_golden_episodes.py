"""Golden vectors for state that must (or must not) survive an episode reset (TEST
INFRASTRUCTURE; runs the reference from /root/reference in the build container only).

Two H2 rollouts (SimulatorBuilding.request_action -> wait_time -> request_observations ->
reward_info -> compute_reward, i.e. Environment._step without tf-agents) on ONE simulator, with
SimulatorBuilding.reset() in between, of the SB1-physics R9 building with the calibrated boiler tank
rates (sim_config.gin:121-122).  What the pair pins:
  * Vav.reset() (vav.py:93-99) zeroes the reheat valve and sets the damper to 0.1, while the
    Thermostat object -- its mode and _previous_timestamp -- survives (thermostat.py:66-69).  Episode 1
    ends with zones in HEAT (valve 1); episode 2 rejects its first requests
    (rejection_simulator_building.py:52-60), so those steps run with mode HEAT but valve 0;
  * SmartDevice._action_timestamp survives too (smart_device.py:71-72) while the clock rewinds: the
    boiler's first observations of episode 2 see a NEGATIVE duration since the last action
    (boiler.py:158-217), which its tank lag and the gas rate's tank term then use;
  * a rejected first step of an episode (no action time stamp is touched).

  python -m oracle.gen_golden_episodes        -> tests/golden/h2_sb1_r9_episodes.npz
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import gen_golden as gg  # noqa: E402


def main() -> None:
  m = gg._mods()
  r9_plan = np.load(os.path.join(gg.OUT, "plan_r9_test.npz"))["floor_plan"]
  start = pd.Timestamp("2023-07-06 07:00:00")   # comfort hours, 273..283 K outside: the zones go to HEAT
  n1, n2 = 14, 16
  sim, building, hvac, weather, schedule = gg.build_sb1(m, r9_plan, start)
  rs = np.random.RandomState(97531)
  acts1 = rs.uniform(-1.0, 1.0, size=(n1, 2)).astype(np.float32)
  acts2 = rs.uniform(-1.0, 1.0, size=(n2, 2)).astype(np.float32)
  e1 = gg.rollout_h2(m, sim, building, hvac, weather, schedule, n1, acts1, (), "sb1_r9 episode 1",
                     rejected_steps=(3,))
  assert e1["valve"][-1].max() == 1.0, "episode 1 must end with a zone in HEAT"
  last_ts = start + pd.Timedelta(300 * (n1 - 1), unit="s")
  rejected2 = (0, 1, 6)
  e2 = gg.rollout_h2(m, sim, building, hvac, weather, schedule, n2, acts2, (), "sb1_r9 episode 2",
                     rejected_steps=rejected2, prev_ts=last_ts)
  assert (e2["mode"][0] == 1).any() and e2["valve"][0].max() == 0.0, "HEAT mode with a closed valve"
  out = {}
  for tag, e in (("e1_", e1), ("e2_", e2)):
    for k, v in e.items():
      out[tag + k] = v
  prm = gg.hvac_params_of(hvac, sim, schedule)
  R = gg.SB1["reward"]
  prm.update({k: float(v) for k, v in dict(
      max_prod=R["max_productivity_personhour_usd"], min_prod=R["min_productivity_personhour_usd"],
      max_elec=R["max_electricity_rate"], max_gas=R["max_natural_gas_rate"],
      prod_delta=R["productivity_midpoint_delta"], prod_stiff=R["productivity_decay_stiffness"],
      w_prod=R["productivity_weight"], w_cost=R["energy_cost_weight"], w_carbon=R["carbon_emission_weight"]).items()})
  out["params_json"] = np.array(json.dumps(prm))
  out["h_conv"] = gg.SB1["weather"]["convection_coefficient"]
  out["initial_temp"] = gg.SB1["initial_temp"]
  out["start_timestamp"] = np.array(str(start))
  path = os.path.join(gg.OUT, "h2_sb1_r9_episodes.npz")
  np.savez_compressed(path, **out)
  print("wrote", path, "tank", e1["blr_tank_temp"][-3:], e2["blr_tank_temp"][:4], "valve e2[0]", e2["valve"][0])


if __name__ == "__main__":
  main()
