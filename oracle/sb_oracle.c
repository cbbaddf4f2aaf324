/*
 * sb_oracle.c -- CPU restatement of the sbsim building-thermal step (TEST INFRASTRUCTURE).
 * See sb_oracle.h for scope and for the reference files this follows.
 *
 * Build with FP contraction OFF (oracle/Makefile): every expression below is written
 * in the reference's association order and must not be fused or reassociated.
 */
#include "sb_oracle.h"

#include <math.h>
#include <string.h>

#define C_AIR 1006.0   /* utils/constants.py:21 AIR_HEAT_CAPACITY */
#define C_WATER 4180.0 /* utils/constants.py:22 WATER_HEAT_CAPACITY */
#define RHO_WATER 1000.0 /* utils/constants.py:46 */
#define GRAVITY 9.8      /* utils/constants.py:47 */

/* ---- numpy add.reduce order --------------------------------------------------------
 * numpy/core/src/umath/loops_utils.h.src  @TYPE@_pairwise_sum (PW_BLOCKSIZE 128), as
 * used by np.mean (building.py:861 np.mean(temps); simulator.py:408 temp.mean()).
 * Third-party arithmetic (numpy 2.2): restated from its published algorithm and pinned
 * by tests/test_oracle_golden.py against numpy itself. */
double sbo_pairwise_sum(const double *a, int64_t n) {
  if (n < 8) {
    double res = 0.;
    for (int64_t i = 0; i < n; i++) res += a[i];
    return res;
  } else if (n <= 128) {
    double r[8], res;
    int64_t i;
    for (i = 0; i < 8; i++) r[i] = a[i];
    for (i = 8; i < n - (n % 8); i += 8) {
      r[0] += a[i + 0]; r[1] += a[i + 1]; r[2] += a[i + 2]; r[3] += a[i + 3];
      r[4] += a[i + 4]; r[5] += a[i + 5]; r[6] += a[i + 6]; r[7] += a[i + 7];
    }
    res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
  } else {
    int64_t n2 = n / 2;
    n2 -= n2 % 8;
    return sbo_pairwise_sum(a, n2) + sbo_pairwise_sum(a + n2, n - n2);
  }
}

/* np.add.reduce over a contiguous 1-D double array (numpy 2.2): the iterator hands the
 * inner loop at most `bufsize` = 8192 elements at a time; each chunk is pairwise-summed
 * and the chunk sums are accumulated left to right (the accumulator starts from the
 * additive identity, which is exact).  np.mean then divides by n
 * (numpy/_core/_methods.py _mean).  Checked against numpy for n in 1..40000 by
 * tests/test_oracle_golden.py::test_pairwise_sum_matches_numpy. */
double sbo_np_mean(const double *a, int64_t n) {
  double s = 0.0;
  for (int64_t i = 0; i < n; i += 8192) {
    int64_t m = n - i < 8192 ? n - i : 8192;
    s = s + sbo_pairwise_sum(a + i, m);
  }
  return s / (double)n;
}

/* building.py:845-871 get_zone_temp_stats / get_zone_average_temps: np.mean of the
 * list [temp[c] for c in room_dict[zone]] (raster order). scratch-free: gather on stack
 * in chunks is not order-preserving for pairwise, so gather fully. */
static double zone_mean(const sbo_plan *p, const double *temp, int z, double *buf) {
  int32_t b = p->zone_off[z], e = p->zone_off[z + 1];
  for (int32_t i = b; i < e; i++) buf[i - b] = temp[p->zone_cells[i]];
  return sbo_np_mean(buf, e - b);
}

void sbo_zone_means(const sbo_plan *p, const double *temp, double *out) {
  double buf[p->H * p->W];
  for (int z = 0; z < p->Z; z++) out[z] = zone_mean(p, temp, z, buf);
}

/* ---- precision experiment (oracle/experiment_fp32_offset.py; SURVEY.md section 7, hard part 1,
 *      option (b)): with mode 1 every control-volume temperature is STORED as a float32 offset
 *      from t_ref (the arithmetic stays float64).  Mode 0, the default, is the reference's own
 *      float64 state and the only mode the parity tests use. */
static int32_t g_state_mode = 0;
static double g_state_tref = 0.0;
void sbo_set_state_mode(int32_t mode, double t_ref) {
  g_state_mode = mode;
  g_state_tref = t_ref;
}

/* ---- one in-place Gauss-Seidel sweep: simulator.py:278-316 ------------------------- */
double sbo_sweep(const sbo_plan *p, const double *prev, double *est, const double *q,
                 double t_amb, double h, double dt) {
  const int N = p->H * p->W;
  const double dx = p->dx;
  const double dx2 = pow(dx, 2.0); /* Python: delta_x**2 */
  double max_delta = 0.0;
  for (int i = 0; i < N; i++) { /* for x in rows: for y in cols (row-major) */
    const int n = p->nbr_cnt[i];
    const int32_t *nb = p->nbr_idx + 4 * i;
    const double k = p->k[i], rho = p->rho[i], c = p->c[i];
    const double last = prev[i];
    double val;
    if (n <= 1) {
      val = t_amb; /* simulator.py:256-258 */
    } else if (n == 2) {
      /* corner: simulator.py:130-142 */
      double t0 = rho * dx2 * c / dt / 2.0;
      double retained = t0 * last;
      double s = 0.0; /* Python sum() starts from int 0 */
      for (int j = 0; j < 2; j++) s = s + est[nb[j]];
      double nt = k * s;
      double conv = 2.0 * h * dx * t_amb;
      double den = 2.0 * k + 2.0 * h * dx + t0;
      val = (nt + conv + retained) / den;
    } else if (n == 3) {
      /* edge: simulator.py:176-195 */
      double t0 = rho * dx2 / 2 * c / dt;
      double retained = t0 * last;
      double s = 0.0;
      for (int j = 0; j < 3; j++) {
        double f = p->nbr_cnt[nb[j]] < 4 ? 0.5 : 1.0;
        s = s + f * est[nb[j]];
      }
      double nt = k * s;
      double conv = h * dx * t_amb;
      double den = 2.0 * k + h * dx + t0;
      val = (nt + conv + retained) / den;
    } else {
      /* interior: simulator.py:225-237 */
      double alpha = k / rho / c;
      double t0 = dx2 / dt / alpha;
      double den = 4.0 + t0;
      double s = 0.0;
      for (int j = 0; j < 4; j++) s = s + est[nb[j]];
      double retained = t0 * last;
      double src = q[i] / k / p->zh;
      val = (s + src + retained) / den;
    }
    if (g_state_mode == 1) val = g_state_tref + (double)(float)(val - g_state_tref);
    double d = fabs(val - est[i]);
    if (d > max_delta) max_delta = d; /* max(delta, max_delta) */
    est[i] = val;
  }
  return max_delta;
}

/* ---- simulator.py:318-371 finite_differences_timestep ------------------------------ */
int32_t sbo_fd_timestep(const sbo_plan *p, double *temp, double *scratch, const double *q,
                        double t_amb, double h, double dt, double thr, int32_t iter_limit,
                        int32_t *n_sweeps) {
  const int N = p->H * p->W;
  memcpy(scratch, temp, sizeof(double) * (size_t)N); /* temp_estimate = temp.copy() */
  int32_t converged = 0, it = 0;
  for (it = 0; it < iter_limit; it++) {
    double md = sbo_sweep(p, temp, scratch, q, t_amb, h, dt);
    if (md <= thr) {
      converged = 1;
      it++;
      break;
    }
  }
  if (n_sweeps) *n_sweeps = it;
  memcpy(temp, scratch, sizeof(double) * (size_t)N); /* building.temp = temp_estimate */
  return converged;
}

/* ---- reset: simulator.py:80-84, building.py:784-791, vav.py:93-99, -------------------
 *      air_handler.py:131-139, boiler.py:110-121.  Thermostat mode / previous timestamp
 *      and SmartDevice._action_timestamp are NOT restored by the reference. The caller
 *      zeroes them once at construction (thermostat.py:66-69, smart_device.py:71). */
void sbo_reset(const sbo_plan *p, const sbo_params *prm, sbo_state *s, double initial_temp,
               const double *reset_temps) {
  const int N = p->H * p->W;
  for (int i = 0; i < N; i++) s->temp[i] = reset_temps ? reset_temps[i] : initial_temp;
  for (int i = 0; i < N; i++) s->input_q[i] = 0.0;
  for (int z = 0; z < p->Z; z++) {
    s->damper[z] = 0.1;
    s->valve[z] = 0.0;
    s->zone_air_temp[z] = 0.0;
  }
  s->ahu_heat_sp = prm->ahu_heat_sp;
  s->ahu_cool_sp = prm->ahu_cool_sp;
  s->ahu_flow = 0.0;
  s->ahu_count = 0;
  s->blr_setpoint = prm->blr_setpoint;
  s->blr_flow = 0.0;
  s->blr_count = 0;
  s->blr_return_temp = 0.0;
  s->blr_tank_temp = prm->blr_setpoint;
  s->blr_tank_change = 0.0;
  s->blr_last_duration = 0.0;
}

/* thermostat.py:76-112 _default_control */
static int32_t default_control(int32_t mode, double tz, double hsp, double csp) {
  double mid = 0.5 * (csp - hsp) + hsp;
  if (tz < hsp) return 1;
  if (tz > csp) return 2;
  if (tz < mid && mode == 1) return 1;
  if (tz > mid && mode == 2) return 2;
  return 0;
}

/* thermostat.py:114-148 update() + vav.py:229-243 (mode -> damper, reheat valve).  comfort_prev:
 * is_comfort_mode(previous timestamp), -1 = no previous timestamp.  Modes: 0 OFF, 1 HEAT, 2 COOL,
 * 3 PASSIVE_COOL. */
int32_t sbo_dev_thermostat(int32_t mode, double tz, double hsp, double csp, int32_t comfort_now,
                           int32_t comfort_prev, double *damper, double *valve) {
  if (comfort_now) {
    mode = default_control(mode, tz, hsp, csp);
  } else if (comfort_prev > 0) {
    mode = 3;
  } else {
    if (mode == 3 && tz > hsp) mode = 3;
    else mode = default_control(mode, tz, hsp, csp);
  }
  if (mode == 1) { *damper = 1.0; *valve = 1.0; }
  else if (mode == 2) { *damper = 1.0; *valve = 0.0; }
  else { *damper = 0.1; *valve = 0.0; }
  return mode;
}

/* simulator.py:383-396 setup_step_sim -> vav.py:219-243 -> thermostat.py:114-148 */
void sbo_setup_step(const sbo_plan *p, const sbo_params *prm, sbo_state *s,
                    int32_t comfort_now, int32_t comfort_prev) {
  double buf[p->H * p->W];
  const double hsp = comfort_now ? prm->comfort_lo : prm->eco_lo;
  const double csp = comfort_now ? prm->comfort_hi : prm->eco_hi;
  for (int z = 0; z < p->Z; z++) {
    double tz = zone_mean(p, s->temp, z, buf);
    s->zone_air_temp[z] = tz;
    s->mode[z] = sbo_dev_thermostat(s->mode[z], tz, hsp, csp, comfort_now,
                                    s->thermostat_has_prev ? comfort_prev : -1, &s->damper[z], &s->valve[z]);
  }
  s->thermostat_has_prev = 1;
}

/* air_handler.py:204-233 */
double sbo_dev_ahu_mixed(double r, double recirc, double amb) {
  return r * recirc + (1 - r) * amb;
}
double sbo_dev_ahu_supply(double heat_sp, double cool_sp, double mixed) {
  if (mixed > cool_sp) return cool_sp;
  if (mixed < heat_sp) return heat_sp;
  return mixed;
}
static double ahu_mixed(double r, double recirc, double amb) { return sbo_dev_ahu_mixed(r, recirc, amb); }
static double ahu_supply(const sbo_state *s, double mixed) {
  return sbo_dev_ahu_supply(s->ahu_heat_sp, s->ahu_cool_sp, mixed);
}
/* air_handler.py:287-320 intake + exhaust fan power; :270-285 thermal energy rate */
double sbo_dev_ahu_blower_power(const sbo_params *prm, double air_flow) {
  double intake = air_flow * prm->ahu_dp / prm->ahu_eff;
  double exhaust = (air_flow * (1.0 - prm->ahu_recirc)) * prm->ahu_dp / prm->ahu_eff;
  return intake + exhaust;
}
double sbo_dev_ahu_thermal_rate(double air_flow, double supply, double mixed) {
  return air_flow * C_AIR * (supply - mixed);
}
/* vav.py:168-195 compute_zone_supply_temp, :197-217 compute_energy_applied_to_zone */
double sbo_dev_vav_supply_temp(double t_sa, double tw, double air_flow, double reheat_flow) {
  double heat_diff = C_AIR * air_flow - C_WATER * reheat_flow;
  double water_heat = tw * C_WATER * reheat_flow;
  return (t_sa * heat_diff + water_heat) / air_flow / C_AIR;
}
double sbo_dev_vav_energy(double air_flow, double t_zs, double tz) { return air_flow * C_AIR * (t_zs - tz); }

/* boiler.py:275-320 compute_thermal_dissipation_rate */
double sbo_boiler_dissipation(const sbo_params *prm, double water_temp, double outside_temp) {
  double delta = water_temp - outside_temp;
  double num = prm->blr_len * 2.0 * M_PI * delta;
  double r1 = prm->blr_radius;
  double r2 = r1 + prm->blr_ins_thick;
  double cond = log(r2 / r1) / prm->blr_ins_k;
  double conv = 1.0 / prm->blr_conv / r2;
  return num / (cond + conv);
}

/* boiler.py:233-273 compute_thermal_energy_rate (+ the tank term of :158-217), :322-333 pump */
double sbo_dev_boiler_gas_rate(const sbo_params *prm, double setpoint, double total_flow, double return_temp,
                               double outside_temp, double tank_change, double last_duration) {
  double supply_w = setpoint > return_temp ? setpoint : return_temp;
  double flow_heat = C_WATER * total_flow * (supply_w - return_temp);
  double diss = sbo_boiler_dissipation(prm, supply_w, outside_temp);
  double tank = 0;
  if (last_duration > 0) tank = C_WATER * prm->blr_capacity * tank_change / last_duration;
  return flow_heat + diss + tank;
}
double sbo_dev_boiler_pump_power(const sbo_params *prm, double total_flow) {
  return total_flow * RHO_WATER * GRAVITY * prm->blr_head / prm->blr_pump_eff;
}

/* boiler.py:_adjust_temperature: the tank moves towards the setpoint at heating_rate /
 * cooling_rate kelvin per minute and stops there. */
double sbo_dev_boiler_adjust(double setpoint, double actual, double secs, double heating_rate, double cooling_rate) {
  double cur;
  if (setpoint > actual) {
    cur = actual + heating_rate * secs / 60.0;
    if (setpoint < cur) cur = setpoint; /* min(x, setpoint) */
  } else if (setpoint < actual) {
    cur = actual - cooling_rate * secs / 60.0;
    if (setpoint > cur) cur = setpoint; /* max(x, setpoint) */
  } else {
    cur = setpoint;
  }
  return cur;
}

/* boiler.py:146-217: reading supply_water_temperature_sensor at obs_ts advances the tank
 * lag (smart_device.py:146-168 stamps _observation_timestamp first). */
void sbo_observe_boiler(const sbo_params *prm, sbo_state *s, double obs_ts) {
  if (s->blr_has_action_ts) {
    s->blr_last_duration = obs_ts - s->blr_action_ts;
  } else {
    s->blr_action_ts = obs_ts;
    s->blr_has_action_ts = 1;
  }
  if (prm->blr_cooling_rate > 0.0 && prm->blr_heating_rate > 0.0) {
    double begin = s->blr_tank_temp;
    double cur = sbo_dev_boiler_adjust(s->blr_setpoint, begin, s->blr_last_duration, prm->blr_heating_rate,
                                       prm->blr_cooling_rate);
    s->blr_tank_temp = cur;
    s->blr_tank_change = cur - begin;
  } else {
    s->blr_tank_temp = s->blr_setpoint;
  }
}

/* reward/setpoint_energy_carbon_regret.py:142-291 on fp32 RewardInfo fields. */
double sbo_reward(const sbo_params *prm, int32_t Z, const float *zone_temp, const float *heat_sp,
                  const float *cool_sp, const float *occ, float blower, float ac, float gas,
                  float pump, double dt, double e_price, double e_carbon, double g_price,
                  double g_carbon, double *diag) {
  double cumulative = 0.0, total_occ = 0.0;
  for (int z = 0; z < Z; z++) {
    /* base_setpoint_energy_carbon_reward.py:52-123 */
    double occupancy = (double)occ[z];
    total_occ += occupancy;
    double hsp = (double)heat_sp[z], csp = (double)cool_sp[z], t = (double)zone_temp[z];
    double x0low = hsp - prm->prod_delta;
    double x0high = csp + prm->prod_delta;
    double prod;
    if (t < hsp) {
      prod = prm->max_prod / (1.0 + exp(-prm->prod_stiff * (t - x0low)));
    } else if (t > csp) {
      prod = prm->max_prod * (1.0 - 1.0 / (1.0 + exp(-prm->prod_stiff * (t - x0high))));
    } else {
      prod = prm->max_prod;
    }
    cumulative += prod * occupancy * dt / 3600.0;
  }
  double max_p = prm->max_prod * total_occ * dt / 3600.0;
  double min_p = prm->min_prod * total_occ * dt / 3600.0;
  double actual = cumulative > min_p ? cumulative : min_p; /* max(actual, min) */
  double npr = total_occ > 0.0 ? (actual - min_p) / (max_p - min_p) - 1.0 : 0.0;

  /* base_setpoint_energy_carbon_reward.py:137-172 */
  double elec = 0.0;
  elec += (double)blower + fabs((double)ac);
  elec += (double)pump;
  double gas_rate = 0.0;
  gas_rate += (double)gas;

  double cap_e = elec < prm->max_elec ? elec : prm->max_elec;
  /* electricity_energy_cost.py:166-224: price * |rate| * dt */
  double ce = e_price * fabs(cap_e) * dt;
  double ce_max = e_price * fabs(prm->max_elec) * dt;
  double ke = e_carbon * fabs(cap_e) * dt;
  double ke_max = e_carbon * fabs(prm->max_elec) * dt;
  double cap_g = gas_rate < prm->max_gas ? gas_rate : prm->max_gas;
  /* natural_gas_energy_cost.py:75-138: negative -> 0; price * (rate * dt) */
  double g_eff = cap_g < 0.0 ? 0.0 : cap_g;
  double cg = g_price * (g_eff * dt);
  double cg_max = g_price * (prm->max_gas * dt);
  double kg = g_carbon * (g_eff * dt);
  double kg_max = g_carbon * (prm->max_gas * dt);

  double nec = (ce + cg) / (ce_max + cg_max);
  double nce = (ke + kg) / (ke_max + kg_max);
  double raw = npr * prm->w_prod - nec * prm->w_cost - nce * prm->w_carbon;
  double reward = raw / (prm->w_prod + prm->w_cost + prm->w_carbon);
  if (diag) {
    diag[0] = actual;
    diag[1] = npr;
    diag[2] = nec;
    diag[3] = nce;
  }
  return reward;
}

/* One Environment._step worth of building work, in the reference's order
 * (environment.py:1228-1309; simulator_building.py:204-268):
 *   request_action: setup_step_sim, then set_action on boiler / AHU
 *   wait_time:      execute_step_sim (simulator_flexible_floor_plan.py:124-190)
 *   _get_observation: one read of boiler.supply_water_temperature_sensor
 *   _get_reward:    reward_info (simulator.py:548-576) + compute_reward */
void sbo_step(const sbo_plan *p, const sbo_params *prm, sbo_state *s, const sbo_step_in *in,
              sbo_step_out *out) {
  const int Z = p->Z, N = p->H * p->W;
  double buf[N];
  double tz_pre[Z], tzs[Z];

  out->action_accepted = in->reject ? 0 : 1;
  if (!in->reject) { /* simulator_building.py:204-263 request_action */
    /* Thermostat._previous_timestamp is the last time update() ran (thermostat.py:88): the caller's
     * comfort_prev is is_comfort_mode of the previous STEP, which is the same thing unless a request
     * was rejected in between; then the value remembered here applies. */
    const int32_t cprev = s->thermostat_skipped ? s->thermostat_prev_comfort : in->comfort_prev;
    sbo_setup_step(p, prm, s, in->comfort_now, cprev);
    s->thermostat_prev_comfort = in->comfort_now;
    s->thermostat_skipped = 0;
    if (in->has_action) { /* smart_device.py:170-201 set_action */
      s->blr_setpoint = in->blr_setpoint;
      s->blr_action_ts = in->now_ts;
      s->blr_has_action_ts = 1;
      s->ahu_heat_sp = in->ahu_heat_sp;
      if (in->has_cool_sp) s->ahu_cool_sp = in->ahu_cool_sp;
      if (in->damper_cmd)
        for (int z = 0; z < Z; z++) {
          const double c = in->damper_cmd[z];
          if (c != c) continue;
          if (c < 0 || c > 1) out->action_accepted = 0; /* vav.py:125-129 raises: this action is rejected */
          else s->damper[z] = c;
        }
    }
  } else {
    s->thermostat_skipped = 1;
  }

  /* ---- execute_step_sim ---- */
  for (int z = 0; z < Z; z++) tz_pre[z] = zone_mean(p, s->temp, z, buf);
  double recirc = sbo_np_mean(s->temp, N); /* building.temp.mean() */
  double mixed = ahu_mixed(prm->ahu_recirc, recirc, in->t_amb_now);
  double t_sa = ahu_supply(s, mixed);

  int32_t n_sweeps = 0;
  int32_t conv = sbo_fd_timestep(p, s->temp, s->scratch, s->input_q, in->t_amb_now, in->h_conv,
                                 prm->dt, prm->conv_threshold, prm->iter_limit, &n_sweeps);

  s->ahu_flow = 0.0; s->ahu_count = 0; /* air_handler.py:250-252 */
  s->blr_flow = 0.0; s->blr_count = 0; /* boiler.py:155-157 */
  for (int z = 0; z < Z; z++) {
    /* vav.py:245-264 output() */
    s->zone_air_temp[z] = tz_pre[z];
    double damper = s->damper[z], valve = s->valve[z];
    double tw = s->blr_setpoint;
    double reheat_flow = valve * prm->vav_max_water_flow;
    double air_flow = damper * prm->vav_max_air_flow;
    double t_zs = sbo_dev_vav_supply_temp(t_sa, tw, air_flow, reheat_flow);
    double qz;
    if (damper == 0 || prm->vav_max_air_flow == 0) qz = 0;
    else qz = sbo_dev_vav_energy(air_flow, t_zs, tz_pre[z]);
    tzs[z] = t_zs;
    if (air_flow > 0) { /* air_handler.py:254-268 */
      s->ahu_flow += air_flow;
      if (s->ahu_flow > prm->ahu_max_flow) s->ahu_flow = prm->ahu_max_flow;
      s->ahu_count += 1;
    }
    if (reheat_flow > 0) { /* boiler.py:219-231 */
      s->blr_flow += reheat_flow;
      s->blr_count += 1;
    }
    /* building.py:873-889 apply_thermal_power_zone */
    for (int32_t i = p->zone_off[z]; i < p->zone_off[z + 1]; i++) {
      int32_t cidx = p->zone_cells[i];
      if (p->diffuser[cidx] > 0.0) s->input_q[cidx] = qz * p->diffuser[cidx];
    }
    if (out->q_zone) out->q_zone[z] = qz;
    if (out->zone_temp_pre) out->zone_temp_pre[z] = tz_pre[z];
  }
  { /* simulator.py:373-381 */
    double num = 0.0, den = 0.0;
    for (int z = 0; z < Z; z++) {
      num += s->valve[z] * tzs[z];
      den += s->valve[z];
    }
    s->blr_return_temp = num / (den + 1e-6);
  }
  const double new_ts = in->now_ts + prm->dt;

  if (in->observe) sbo_observe_boiler(prm, s, new_ts);

  /* ---- reward_info at the new timestamp: simulator.py:457-576 ---- */
  float zt32[Z], hs32[Z], cs32[Z], oc32[Z];
  for (int z = 0; z < Z; z++) {
    double tzp = zone_mean(p, s->temp, z, buf);
    if (out->zone_temp_post) out->zone_temp_post[z] = tzp;
    zt32[z] = (float)tzp;
    hs32[z] = (float)(in->comfort_next ? prm->comfort_lo : prm->eco_lo);
    cs32[z] = (float)(in->comfort_next ? prm->comfort_hi : prm->eco_hi);
    oc32[z] = (float)in->occupancy[z];
  }
  /* air_handler.py:287-320 fan powers */
  double blower = sbo_dev_ahu_blower_power(prm, s->ahu_flow);
  double recirc2 = sbo_np_mean(s->temp, N);
  double mixed2 = ahu_mixed(prm->ahu_recirc, recirc2, in->t_amb_next);
  double supply2 = ahu_supply(s, mixed2);
  double ac = sbo_dev_ahu_thermal_rate(s->ahu_flow, supply2, mixed2);
  double gas = sbo_dev_boiler_gas_rate(prm, s->blr_setpoint, s->blr_flow, s->blr_return_temp, in->t_amb_next,
                                       s->blr_tank_change, s->blr_last_duration);
  double pump = sbo_dev_boiler_pump_power(prm, s->blr_flow);

  out->n_sweeps = n_sweeps;
  out->converged = conv;
  out->t_supply_air = t_sa;
  out->recirc_pre = recirc;
  out->blower_rate = (float)blower;
  out->ac_rate = (float)ac;
  out->gas_rate = (float)gas;
  out->pump_rate = (float)pump;
  double diag[4];
  out->reward_f64 = sbo_reward(prm, Z, zt32, hs32, cs32, oc32, out->blower_rate, out->ac_rate,
                               out->gas_rate, out->pump_rate, prm->dt, in->e_price, in->e_carbon,
                               in->g_price, in->g_carbon, diag);
  out->reward = (float)out->reward_f64;
  out->productivity = diag[0];
  out->norm_prod_regret = diag[1];
  out->norm_energy_cost = diag[2];
  out->norm_carbon = diag[3];
}
