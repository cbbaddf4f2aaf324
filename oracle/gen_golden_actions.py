"""Golden vectors for the configurable action set and the action-rejection path (TEST
INFRASTRUCTURE; runs the reference from /root/reference in the build container only).

One 60-step H2 rollout (SimulatorBuilding.request_action -> wait_time -> request_observations ->
reward_info -> compute_reward, i.e. Environment._step without tf-agents) of the SB1-physics R9
building with a four-column action vector -- boiler supply_water_setpoint, air handler
supply_air_heating_temperature_setpoint, air handler supply_air_cooling_temperature_setpoint, one
VAV's supply_air_damper_percentage_command (air_handler.py:97-104, vav.py:65-69) -- in which
  * the building rejects the whole request at some steps (rejection_simulator_building.py:52-60:
    request_action raises, Environment._step goes on and returns reward -inf, environment.py:1266-1309),
  * the damper command leaves [0, 1] at others (vav.py:125-129 raises ValueError: that one action
    comes back REJECTED_NOT_ENABLED_OR_AVAILABLE, the rest apply, the reward is -inf again).

  python -m oracle.gen_golden_actions        -> tests/golden/h2_sb1_r9_actions.npz
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
  start = pd.Timestamp("2023-07-06 17:30:00")   # crosses the 19:00 comfort -> eco switch (US/Pacific = 02:00 UTC? no: naive = UTC)
  n = 60
  sim, building, hvac, weather, schedule = gg.build_sb1(m, r9_plan, start)
  zones = list(hvac.vavs.keys())
  vav = hvac.vavs[zones[4]]
  extra = (("air_handler_id", "supply_air_cooling_temperature_setpoint", (296.0, 302.0)),
           (vav.device_id(), "supply_air_damper_percentage_command", (0.0, 1.0)))
  rs = np.random.RandomState(4321)
  acts = rs.uniform(-1.0, 1.0, size=(n, 4)).astype(np.float32)
  acts[[11, 33], 3] = np.float32(1.5)     # damper command 1.25: the VAV's setter raises
  acts[40, 3] = np.float32(-1.2)          # ... and -0.1
  rejected = (5, 6, 20, 21, 22, 47)
  out = gg.rollout_h2(m, sim, building, hvac, weather, schedule, n, acts, (1, 30), "sb1_r9 actions+rejections",
                      extra_actions=extra, rejected_steps=rejected)
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
  out["action_names"] = np.array(["supply_water_setpoint", "supply_air_heating_temperature_setpoint",
                                  "supply_air_cooling_temperature_setpoint", "supply_air_damper_percentage_command"])
  out["action_zone"] = np.array([0, 0, 0, 4])
  out["action_ranges"] = np.array([gg.SB1["action_ranges"]["supply_water_setpoint"],
                                   gg.SB1["action_ranges"]["supply_air_heating_temperature_setpoint"],
                                   extra[0][2], extra[1][2]], dtype=np.float64)
  path = os.path.join(gg.OUT, "h2_sb1_r9_actions.npz")
  np.savez_compressed(path, **out)
  print("wrote", path, "rejected", out["rejected"].sum(), "not accepted", (1 - out["action_accepted"]).sum())


if __name__ == "__main__":
  main()
