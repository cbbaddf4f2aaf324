"""bench.py -- env-steps/s (zone-updates/s) of the batched building-thermal step on MI355X.

Workload (BASELINE.json configs[1], SURVEY.md 8d "Config 2"): 65 536 replicas per GPU of the
SB1-physics building on the reference's 3x3-room floor plan R9 (68x98 control volumes, 9
zones), per-building initial temperatures 294 + N(0,1) K (seed 7, clipped to [285,305]),
random setpoint actions U[-1,1]^2 per building per step (torch.Generator seed 1234+rank),
shared sinusoid weather 273-283 K, step-function occupancy, start 2023-07-06 07:00.
A "step" is one BatchedEnvironment.step() over the whole batch = sb_step = three launches
(per-building device algebra, Gauss-Seidel sweep kernel, reward / observation), inputs
already resident in HBM.  Synthetic data, float64 arithmetic.  The roofline object prices the
sweep kernel (> 95 % of a step), bracketed by HIP events through sb_step_phases.

Contract: `python bench.py --gpus N --steps K --warmup W`; under torchrun one rank per GPU
(weak scaling: 65 536 buildings per GPU, no data-path collective; one RCCL all_gather of
the per-building returns after the rollout, timed separately).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import datetime as dt
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

from sbsim_amd import _ffi, distributed as sd  # noqa: E402
from sbsim_amd.environment import BatchedEnvironment, MixedBatchedEnvironment, SimConfig  # noqa: E402
from sbsim_amd.floorplan import FloorPlan, Materials, rectangular_floor_plan  # noqa: E402

HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)


# what decides the sweep kernel's memory behaviour: its sources, the planner (LDS layout, tables) and the build flags
KERNEL_SOURCES = ("sbsim_amd/csrc/step_roll.hip", "sbsim_amd/csrc/sweep_common.h", "sbsim_amd/csrc/sb_device.h",
                  "sbsim_amd/csrc/sbsim_hip.hip", "sbsim_amd/build.py")


def kernel_source_sha256() -> str:
  """Hash of the files the bench kernel (k_sweep_roll) is compiled from: what ties a committed PMC traffic profile
  (profiles/traffic_latest.json, stamped by tools/adopt_profiles.py) to the kernel of this run."""
  import hashlib
  h = hashlib.sha256()
  for rel in KERNEL_SOURCES:
    with open(os.path.join(ROOT, rel), "rb") as fh:
      h.update(fh.read())
  return h.hexdigest()


def r9_plan() -> FloorPlan:
  return FloorPlan.from_file_input(rectangular_floor_plan((3, 3), (20, 30)), Materials.sb1(), 10.0, 300.0)


def cpu_baseline(env: BatchedEnvironment, plan: FloorPlan, init: np.ndarray, acts: np.ndarray,
                 warmup: int, target_cpu_seconds: float = 30.0):
  """Times the CPU oracle ("port": oracle/sb_oracle.c, float64, reference operation order,
  OpenMP over buildings) on a bounded sample of the same workload, and uses the same sample
  as an in-run parity check of the GPU path.  The oracle is the checker and the baseline,
  never the product.  Like the GPU leg it first runs the `warmup` steps untimed (the start
  state is not an equilibrium: the first steps need 10-100 sweeps), so both legs are timed on
  the same steps of the same rollout."""
  from oracle import oracle as orc
  c = env.config
  oplan = orc.OraclePlan(plan.conductivity, plan.density, plan.heat_capacity, plan.exterior_space,
                         plan.zone_cell_lists(), plan.diffusers, plan.cv_size_cm, plan.floor_height_cm)
  oprm = orc.OracleParams(
      dt=c.time_step_sec, conv_threshold=c.convergence_threshold, iter_limit=c.iteration_limit,
      vav_max_air_flow=c.vav_max_air_flow_rate, vav_max_water_flow=c.vav_reheat_max_water_flow_rate,
      ahu_recirc=c.ahu_recirculation, ahu_heat_sp=c.ahu_heating_air_temp_setpoint,
      ahu_cool_sp=c.ahu_cooling_air_temp_setpoint, ahu_dp=c.ahu_fan_differential_pressure,
      ahu_eff=c.ahu_fan_efficiency, blr_setpoint=c.boiler_reheat_water_setpoint,
      blr_head=c.boiler_water_pump_differential_head, blr_pump_eff=c.boiler_water_pump_efficiency,
      comfort_lo=c.comfort_temp_window[0], comfort_hi=c.comfort_temp_window[1],
      eco_lo=c.eco_temp_window[0], eco_hi=c.eco_temp_window[1],
      blr_heating_rate=c.boiler_heating_rate, blr_cooling_rate=c.boiler_cooling_rate, ahu_has_weather=1)
  threads = max(1, min(orc.lib().sbo_max_threads(), os.cpu_count() or 1))
  quota = None   # a container's CPU quota (cgroup v2 cpu.max / v1 cfs quota): more threads than that only get throttled
  try:
    q, per = open("/sys/fs/cgroup/cpu.max").read().split()
    quota = None if q == "max" else float(q) / float(per)
  except (OSError, ValueError):
    try:
      q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
      quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()) if q > 0 else None
    except (OSError, ValueError):
      pass
  if quota is not None:
    threads = max(1, min(threads, int(quota + 0.5)))
  nb = min(init.shape[0], max(8, 32 * threads))   # >= 32 buildings per thread and step: tens of ms of work between two barriers
  batch = orc.OracleBatch(oplan, oprm, init[:nb])
  lo, hi = c.action_ranges
  ts = env._start_timestamp
  step = dt.timedelta(seconds=c.time_step_sec)
  prev = None
  n_steps, t_cpu, sweeps, step_times = 0, 0.0, 0, []
  cpu_s, load0 = 0.0, os.getloadavg()[0]   # process CPU time inside batch.step (all its OpenMP threads) and the host's load before the sample
  single_times, n_now = [], threads   # after the all-cores sample: two steps of the same rollout on ONE thread
  max_steps = acts.shape[0]
  while n_steps < max_steps:
    env._prev_thermostat_ts = prev
    si = env.make_step_in(ts)
    ins = []
    for b in range(nb):
      a = acts[n_steps, b]
      native = [np.float32((float(a[0]) + 1.0) / 2.0 * (lo[1] - lo[0]) + lo[0]),
                np.float32((float(a[1]) + 1.0) / 2.0 * (hi[1] - hi[0]) + hi[0])]
      ins.append(orc.make_step_in(
          now_ts=300.0 * n_steps, t_amb_now=si.t_amb_now, h_conv=env.weather.convection_coefficient,
          t_amb_next=si.t_amb_next, comfort_now=si.comfort_now, comfort_prev=max(si.comfort_prev, 0),
          comfort_next=si.comfort_next, occupancy=np.full(oplan.Z, si.occupancy), observe=1,
          e_price=si.e_price, e_carbon=si.e_carbon, g_price=si.g_price, g_carbon=si.g_carbon,
          action=native))
    t0, c0 = time.perf_counter(), time.process_time()
    outs = batch.step(ins, n_threads=n_now)
    c1 = time.process_time()
    n_steps += 1
    prev, ts = ts, ts + step
    if n_steps <= warmup:
      continue
    if n_now == 1:
      single_times.append(time.perf_counter() - t0)
      if len(single_times) >= 2:
        break
      continue
    step_times.append(time.perf_counter() - t0)
    t_cpu += step_times[-1]
    cpu_s += c1 - c0
    sweeps += sum(outs[b].n_sweeps for b in range(nb))
    # bounded sample: at least ~2.5 s of wall time (a sub-second sample on a shared host is noise),
    # at most ~25 s
    if (t_cpu >= 2.5 and t_cpu * threads >= target_cpu_seconds and len(step_times) >= 12) or t_cpu > 25.0:
      if threads == 1 or n_steps + 2 > max_steps:
        break
      n_now = 1
  env._prev_thermostat_ts = None
  zones = oplan.Z
  n_timed = len(step_times)
  # the rate is the MEAN step's (what a user of the CPU path gets); the median step's rate next to it says how much of
  # the difference is the host (other tenants' threads stretch single steps)
  mean_s, median_s = float(np.mean(step_times)), float(np.median(step_times))
  value = nb * zones / mean_s
  single = nb * zones / float(min(single_times)) if single_times else None
  grids = np.stack([b.grid() for b in batch.buildings])
  try:
    affinity = len(os.sched_getaffinity(0))
  except (AttributeError, OSError):
    affinity = None
  try:
    cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
  except (OSError, StopIteration):
    cpu_model = None
  load1 = os.getloadavg()[0]
  # a host with other work on it: its load before the sample was already more than a few CPUs' worth
  quiet = load0 < 0.05 * (os.cpu_count() or 1) + 1.0
  return dict(value=value, unit="zone-updates/s", cores=threads, kind="port", cpu_model=cpu_model,
              value_from_median_step=nb * zones / median_s, step_ms_mean=mean_s * 1e3, step_ms_median=median_s * 1e3,
              # value / (one thread's rate x threads): what OpenMP over buildings makes of the host's cores
              parallel_efficiency=(value / (single * threads) if single else None), host_quiet=quiet,
              # what really ran: CPU seconds the process burnt inside the timed steps / their wall time (= threads that were
              # actually on a core: a shared host gives fewer than it advertises), the CPUs this process may use, the host's load
              threads_effective=cpu_s / t_cpu if t_cpu > 0 else None, cpus_allowed=affinity, host_cpus=os.cpu_count(),
              cgroup_cpu_quota=quota,   # CPUs' worth of time the container may use: the thread count above is capped to it
              host_loadavg_1min_before_after=[load0, load1],
              # (OMP_PROC_BIND=close / OMP_PLACES=threads was tried on the round's shared 256-CPU hosts: 5 x SLOWER -- the pinned
              # threads land on CPUs other tenants keep busy; unbound, the kernel moves them)
              omp={k: os.environ.get(k) for k in ("OMP_PROC_BIND", "OMP_PLACES", "OMP_WAIT_POLICY")},
              sample=f"{nb} buildings x {n_timed} steps of the bench workload after {warmup} untimed "
                     f"warm-up steps ({sweeps / (nb * n_timed):.2f} sweeps/step), oracle/sb_oracle.c "
                     f"with OpenMP over buildings (schedule(dynamic, 1)), {t_cpu:.2f} s wall x {threads} threads; rate from the MEAN step"
                     + ("" if quiet else "; the host was NOT quiet (load average before the sample above)"),
              env_steps_per_s=nb / mean_s,
              single_thread_value=single), grids, n_steps, nb


def launch_ranks_if_needed(args) -> None:
  """`python bench.py --gpus N` outside torchrun: re-executes itself as N ranks under
  torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1) and exits with the
  launcher's status.  Under torchrun (WORLD_SIZE set) the world size must equal --gpus."""
  world_env = os.environ.get("WORLD_SIZE")
  if world_env is not None:
    if int(world_env) != args.gpus:
      raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world_env} ranks")
    return
  if args.gpus <= 1:
    return
  with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
         "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
  env = dict(os.environ)
  env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
  raise SystemExit(subprocess.call(cmd, env=env))


def stub_rank(args) -> None:
  """The launcher / barrier / max-over-ranks / return-gather plumbing with a stub step and the
  gloo backend: what tests/test_bench_launcher.py runs on CPU (no GPU, no HIP library)."""
  import torch.distributed as dist
  rank, _, world = sd.env_rank_world()
  distributed = sd.init_process_group("gloo")
  if distributed:
    assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
  dev = torch.device("cpu")
  pf = sd.preflight(dev, args.gpus)       # before any work: world size, a 256 KiB all_gather, all_reduce(MAX)
  B, K, W = args.buildings, args.steps, args.warmup
  mixed = args.config == "mixed"
  if mixed:   # three classes, each block-partitioned over the ranks on its own: a building's "global index" is class-major
    per = B // len(MIXED_CLASSES)
    totals = [world * per] * len(MIXED_CLASSES)
    spans = sd.class_shard_ranges(totals, rank, world)
    ids = torch.cat([torch.arange(lo, hi, dtype=torch.float32) + sum(totals[:k]) for k, (lo, hi) in enumerate(spans)])
    n_total = sum(totals)
  else:
    lo, hi = sd.shard_range(world * B, rank, world)
    ids = torch.arange(lo, hi, dtype=torch.float32)
    n_total = world * B
  returns = torch.zeros((ids.numel(),), dtype=torch.float32)
  for _ in range(W):
    time.sleep(1e-3)
  sd.barrier(dev)
  t0 = time.perf_counter()
  for t in range(K):
    time.sleep(1e-3)
    returns += ids
  sd.barrier(dev)
  mine = time.perf_counter() - t0
  per_rank = sd.all_ranks(mine / K * 1e3, dev)
  elapsed = sd.max_over_ranks(mine, dev)
  g0 = time.perf_counter()
  all_returns = sd.gather_returns_by_class(returns, totals) if mixed else sd.gather_returns(returns, n_total)
  gather_ms = (time.perf_counter() - g0) * 1e3
  ok = bool(torch.equal(all_returns, torch.arange(n_total, dtype=torch.float32) * K))
  if rank == 0:
    print(json.dumps({"metric": "stub", "value": n_total * K / elapsed, "unit": "env-steps/s", "n_gpus": world,
                      "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
                      "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "stub",
                      "return_gather_ms": gather_ms, "gathered_returns": int(all_returns.numel()),
                      "gather_in_global_order": ok, "rccl_ranks": pf["ranks"], "preflight": pf,
                      "per_rank_ms_per_step": per_rank,
                      "config": {"workload": "launcher plumbing test (stub step, gloo)" + (", three classes sharded per class" if mixed else "")}}))
  if distributed:
    dist.destroy_process_group()



class TwinCheck:
  """In-run parity for the configs that do not carry the cpu_baseline leg: the first `n` buildings of a
  BatchedEnvironment are re-stepped on CPU-oracle twins (oracle/sb_oracle.c -- the checker, never the
  product) with the very inputs the device saw; Gauss-Seidel sweep counts must be EQUAL and zone
  temperatures within 1e-8 K at every step.  `record` copies a few numbers to the host per step, so it
  is off unless asked for (--check-buildings)."""

  def __init__(self, env: BatchedEnvironment, plan: FloorPlan, init_temps: np.ndarray, n: int):
    from oracle import oracle as orc   # checker only
    self.env, self.n, self.steps = env, n, []
    oplan = orc.OraclePlan(plan.conductivity, plan.density, plan.heat_capacity, plan.exterior_space,
                           plan.zone_cell_lists(), plan.diffusers, plan.cv_size_cm, plan.floor_height_cm)
    c = env.config
    oprm = orc.OracleParams(
        dt=c.time_step_sec, conv_threshold=c.convergence_threshold, iter_limit=c.iteration_limit,
        vav_max_air_flow=c.vav_max_air_flow_rate, vav_max_water_flow=c.vav_reheat_max_water_flow_rate,
        ahu_recirc=c.ahu_recirculation, ahu_heat_sp=c.ahu_heating_air_temp_setpoint,
        ahu_cool_sp=c.ahu_cooling_air_temp_setpoint, ahu_dp=c.ahu_fan_differential_pressure,
        ahu_eff=c.ahu_fan_efficiency, blr_setpoint=c.boiler_reheat_water_setpoint,
        blr_head=c.boiler_water_pump_differential_head, blr_pump_eff=c.boiler_water_pump_efficiency,
        comfort_lo=c.comfort_temp_window[0], comfort_hi=c.comfort_temp_window[1],
        eco_lo=c.eco_temp_window[0], eco_hi=c.eco_temp_window[1],
        blr_heating_rate=c.boiler_heating_rate, blr_cooling_rate=c.boiler_cooling_rate, ahu_has_weather=1)
    H, W = plan.shape
    self.twins = [orc.OracleBuilding(oplan, oprm, 0.0, reset_temps=np.full(H * W, float(init_temps[b]))) for b in range(n)]
    for tw in self.twins:
      tw.observe_boiler(0.0)   # Environment.reset()'s observation

  def record(self, step_in, actions: torch.Tensor) -> None:
    """After the device step with `step_in` / `actions`: what the twins need and what the device found."""
    env, n = self.env, self.n   # (device-side copies: no host synchronisation inside a timed loop)
    self.steps.append((step_in, actions[:n].detach().clone(), env.info[:n, 4].clone(), env.sim.zone_temps()[:n].clone()))

  def verify(self) -> dict:
    lo, hi = self.env.config.action_ranges
    worst, mismatches = 0.0, 0
    for t, (si, acts, nsw, zt) in enumerate(self.steps):
      acts, nsw, zt = acts.cpu().numpy(), nsw.cpu().numpy(), zt.cpu().numpy()
      for b, tw in enumerate(self.twins):
        native = [np.float32((float(acts[b, 0]) + 1.0) / 2.0 * (lo[1] - lo[0]) + lo[0]),
                  np.float32((float(acts[b, 1]) + 1.0) / 2.0 * (hi[1] - hi[0]) + hi[0])]
        o = tw.step(now_ts=300.0 * t, t_amb_now=si.t_amb_now, h_conv=100.0, t_amb_next=si.t_amb_next,
                    comfort_now=bool(si.comfort_now), comfort_prev=si.comfort_prev == 1,
                    comfort_next=bool(si.comfort_next), occupancy=si.occupancy, e_price=si.e_price,
                    e_carbon=si.e_carbon, g_price=si.g_price, g_carbon=si.g_carbon, action=native)
        mismatches += int(int(nsw[b]) != o["n_sweeps"])
        worst = max(worst, float(np.abs(zt[b] - o["zone_temp_post"]).max()))
    return {"buildings": self.n, "steps": len(self.steps), "sweep_count_mismatches": mismatches, "max_abs_dT_zone_K": worst}


MIXED_CLASSES = [("R9", (3, 3), (20, 30)), ("SB2-synth", (8, 5), (12, 14)), ("SB1-synth", (14, 9), (8, 7))]


def mixed_config(args, ctx=None, emit=True):
  """BASELINE.json configs[2] (SURVEY.md 8d "Config 3"): the batch split evenly over three floor-plan
  classes -- R9 (9 zones), "SB2-synth" (8x5 rooms, 40 zones), "SB1-synth" (14x9 rooms, 126 zones, the
  real SB1's VAV count; SB2 / SB3 do not exist in the reference) -- behind ONE environment
  (`MixedBatchedEnvironment`: one library handle and one HIP stream per class inside, so that the classes'
  launches overlap), stepped through its public `step()` with random setpoint actions.  One JSON line;
  `roofline` sums the algorithmic bytes of the three sweep kernels over their summed durations, each measured
  with the chip to itself in three rounds after the timed ones.
  `--gpus N`: every class is block-partitioned over the ranks on its own (each GPU holds the same
  class mix), `--buildings` per GPU, no per-step collective, one all_gather of the per-building returns at the
  end (`sbsim_amd.distributed.gather_returns_by_class`: global class-major order)."""
  rank, dev, distributed, pf = ctx if ctx is not None else start_rank(args)
  local_rank, world = dev.index, pf["ranks"]
  torch.cuda.set_device(local_rank)
  B_each, K, W = args.buildings // len(MIXED_CLASSES), args.steps, args.warmup
  N_ALONE = 3
  plans = [FloorPlan.from_file_input(rectangular_floor_plan(rooms, shape), Materials.sb1(), 10.0, 300.0)
           for _, rooms, shape in MIXED_CLASSES]
  menv = MixedBatchedEnvironment([(p, world * B_each) for p in plans], device=local_rank, rank=rank, world=world,
                                 holiday_calendar="us", collect_info=True, num_days_in_episode=3)
  menv.reset()
  rs = np.random.RandomState(sd.shard_seed(7, rank))
  t_init = torch.tensor(np.clip(294.0 + rs.randn(B_each), 285.0, 305.0), dtype=torch.float64, device=dev)
  n_check = args.check_buildings if rank == 0 else 0
  checks = []
  for env, plan in zip(menv.envs, plans):   # every class starts from the same per-building temperatures
    H, Wd = plan.shape
    env.sim.reset(temps=t_init[:, None].expand(B_each, H * Wd).contiguous())
    checks.append(TwinCheck(env, plan, t_init[:n_check].cpu().numpy(), n_check) if n_check else None)
  gen = torch.Generator(device=dev)
  gen.manual_seed(sd.shard_seed(1234, rank))
  acts = torch.rand((W + K + N_ALONE, menv.batch_size, 2), generator=gen, device=dev, dtype=torch.float32) * 2.0 - 1.0
  returns = torch.zeros((menv.batch_size,), dtype=torch.float32, device=dev)
  torch.cuda.synchronize(dev)

  def barrier():
    sd.barrier(dev)
    torch.cuda.synchronize(dev)

  def round_(t):
    sis = [env.make_step_in(env.current_simulation_timestamp) if c else None for env, c in zip(menv.envs, checks)]
    ts = menv.step(acts[t])
    returns.add_(ts.reward)
    for (lo, hi), c, si in zip(menv.slices, checks, sis):
      if c:
        c.record(si, acts[t, lo:hi])

  alone_ms = [[] for _ in menv.envs]

  def alone(t):
    """One untimed round with the classes one after the other: a class's sweep kernel with the chip to itself."""
    for k, (env, stream, (lo, hi)) in enumerate(zip(menv.envs, menv.streams, menv.slices)):
      torch.cuda.synchronize(dev)
      with torch.cuda.stream(stream):
        si = env.make_step_in(env.current_simulation_timestamp)
        a = (acts[t, lo:hi], si, env._obs, env._reward, env._info)
        env.sim.step(*a, phases=1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        env.sim.step(*a, phases=2)
        e1.record()
        env.sim.step(*a, phases=4)
        env._prev_thermostat_ts = env._now
        env._now = env._now + env._step_interval
        env._step_count += 1
        if checks[k]:
          checks[k].record(si, acts[t, lo:hi])
      torch.cuda.synchronize(dev)
      alone_ms[k].append(e0.elapsed_time(e1))

  for t in range(W):
    round_(t)
  barrier()
  menv.profile = True
  t0 = time.perf_counter()
  for t in range(W, W + K):
    round_(t)
  barrier()
  mine = time.perf_counter() - t0
  menv.profile = False
  per_rank = sd.all_ranks(mine / K * 1e3, dev)
  elapsed = sd.max_over_ranks(mine, dev)
  gather_ms, n_gathered = 0.0, menv.batch_size
  if distributed:
    g0 = time.perf_counter()
    n_gathered = int(sd.gather_returns_by_class(returns, menv.class_totals).numel())
    torch.cuda.synchronize(dev)
    gather_ms = (time.perf_counter() - g0) * 1e3
    assert n_gathered == sum(menv.class_totals)
  for t in range(W + K, W + K + N_ALONE):   # after the timed rounds, in the same regime
    alone(t)
  per_class, alg_bytes, kern_s = {}, 0.0, 0.0
  for k, ((name, _, _), env) in enumerate(zip(MIXED_CLASSES, menv.envs)):
    li = env.sim.launch_info
    ms_alone = float(np.mean(alone_ms[k]))
    alg = li["algorithmic_bytes_per_env_step"] * env.sim.B
    alg_bytes += alg
    kern_s += ms_alone * 1e-3
    per_class[name] = {
        "grid": list(env.sim.plan.shape), "zones": env.sim.Z, "buildings": env.sim.B,
        "kernel": _ffi.SWEEP_KERNELS.get(li.get("kernel", -1), "?"),
        # in the timed rounds the classes overlap on their streams (a class also waits for CUs): the GPU time of a
        # class's whole step on its stream; `alone`: its sweep kernel with the chip to itself
        "step_ms_in_round": float(np.mean([a.elapsed_time(b) for a, b in menv.class_events[k]])),
        "sweep_kernel_ms_alone": ms_alone,
        "mean_sweeps_per_env_step": float(env.info[:, 4].mean()),
        "roofline_frac_alone": alg / (ms_alone * 1e-3) / 1e9 / HBM_PEAK_GBPS,
        "launch": li}
    if checks[k]:
      per_class[name]["parity_vs_oracle"] = checks[k].verify()
  zone_updates = world * sum(env.sim.B * env.sim.Z for env in menv.envs) * K
  env_steps = world * menv.batch_size * K
  achieved = alg_bytes / kern_s / 1e9
  result = {
      "metric": "zone-updates/sec over three floor-plan classes (mixed zone counts), %d MI355X" % world,
      "value": zone_updates / elapsed, "unit": "zone-updates/s", "n_gpus": world, "steps": K, "warmup": W,
      "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
      "dtype": "f64", "data": "synthetic", "env_steps_per_s": env_steps / elapsed,
      "return_gather_ms": gather_ms, "gathered_returns": n_gathered,
      "rccl_ranks": pf["ranks"], "preflight": pf, "per_rank_ms_per_step": per_rank,
      "timed_through": "MixedBatchedEnvironment.step()",
      "config": {"workload": "BASELINE.json configs[2]: %d buildings x 3 floor-plan classes (R9, SB2-synth 40 zones, "
                             "SB1-synth 126 zones) per GPU behind one MixedBatchedEnvironment (one handle and one HIP stream "
                             "per class inside), random setpoint actions" % B_each,
                 "parallelism": f"{world} x (the same class mix: every class block-partitioned over the ranks), no per-step collective",
                 "class_ranges_rank0": [list(r) for r in menv.class_ranges], "class_totals": menv.class_totals,
                 "observation_width": menv.observation_spec().shape[0], "classes": per_class},
      "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                   "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                   "note": "sum of the three sweep kernels' algorithmic bytes over the sum of their durations, each "
                           "measured with the chip to itself (in the timed rounds the kernels overlap: wall time per "
                           "round is ms_per_step)"}}
  menv.close()
  if emit and rank == 0:
    print(json.dumps(result))
  if emit and distributed:
    import torch.distributed as dist
    dist.destroy_process_group()
  return result


def start_rank(args):
  """One process per GPU: binds the rank to its device, joins the process group ("nccl" = RCCL on ROCm) and
  runs the preflight check (sbsim_amd.distributed.preflight: world size, one rank per device, a 256 KiB
  all_gather and an all_reduce(MAX)) BEFORE any environment is built.  -> (rank, device, distributed?, preflight)."""
  rank, local_rank, world = sd.env_rank_world()
  local_rank = sd.device_index(local_rank)      # (SBSIM_BENCH_SHARE_GPU=1: ranks share the visible devices)
  if local_rank >= torch.cuda.device_count():
    raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank} but {torch.cuda.device_count()} are visible "
                     f"(--gpus {args.gpus}: one process per GPU of this node)")
  torch.cuda.set_device(local_rank)
  dev = torch.device("cuda", local_rank)
  distributed = sd.init_process_group(sd.backend_for_gpu())
  if not distributed and args.gpus != 1:
    raise SystemExit(f"bench.py: --gpus {args.gpus} but only one rank is running")
  pf = sd.preflight(dev, args.gpus)
  return rank, dev, distributed, pf


def policy_config(args, ctx=None, emit=True):
  """BASELINE.json configs[4] (SURVEY.md 8d "Config 5"): the configs[1] batch per GPU driven through
  `BatchedEnvironment.step()` by a SAC-shaped actor (2 x 128 MLP, tanh-squashed Gaussian, the policy
  network of the reference's SAC notebook) evaluated on the environment's own GPU -- observations
  never leave HBM, data-parallel actors, no per-step collective.  env-steps/s INCLUDE policy inference
  and the host-side step inputs.  tf-agents is not installable here: the loop makes the same env API
  calls (`reset()`, `step(action)` -> TimeStep) a tf-agents driver would."""
  rank, dev, distributed, pf = ctx if ctx is not None else start_rank(args)
  local_rank, world = dev.index, pf["ranks"]
  if distributed:
    import torch.distributed as dist
  B, K, W = args.buildings, args.steps, args.warmup
  plan = r9_plan()
  env = BatchedEnvironment(plan, B, device=local_rank, holiday_calendar="us", num_days_in_episode=3,
                           collect_info=bool(args.check_buildings))
  H, Wd = plan.shape
  Z = env.sim.Z
  ts = env.reset()
  rs = np.random.RandomState(sd.shard_seed(7, rank))
  t_init = np.clip(294.0 + rs.randn(B), 285.0, 305.0)
  env.sim.reset(temps=torch.tensor(t_init, dtype=torch.float64, device=dev)[:, None].expand(B, H * Wd).contiguous())
  torch.manual_seed(sd.shard_seed(0, rank))
  O = env.observation_spec().shape[0]
  actor = torch.nn.Sequential(torch.nn.Linear(O, 128), torch.nn.ReLU(), torch.nn.Linear(128, 128), torch.nn.ReLU(),
                              torch.nn.Linear(128, 4)).to(dev)
  gen = torch.Generator(device=dev)
  gen.manual_seed(sd.shard_seed(1, rank))

  @torch.no_grad()
  def act(obs):
    out = actor(torch.nan_to_num(obs))
    mean, log_std = out[:, :2], out[:, 2:].clamp(-5, 2)
    return torch.tanh(mean + log_std.exp() * torch.randn(mean.shape, device=dev, generator=gen)).contiguous()

  def barrier():
    sd.barrier(dev)
    torch.cuda.synchronize(dev)

  obs = ts.observation
  returns = torch.zeros((B,), dtype=torch.float32, device=dev)
  check = None
  if args.check_buildings and rank == 0:
    check = TwinCheck(env, plan, t_init, args.check_buildings)

  def step_once(obs):
    a = act(obs)
    si = env.make_step_in(env.current_simulation_timestamp) if check else None
    ts = env.step(a)
    if check:
      check.record(si, a)
    return ts

  for _ in range(W):
    ts = step_once(obs)
    obs = ts.observation
  barrier()
  t0 = time.perf_counter()
  for _ in range(K):
    ts = step_once(obs)
    obs = ts.observation
    returns += ts.reward
  barrier()
  mine = time.perf_counter() - t0
  per_rank = sd.all_ranks(mine / K * 1e3, dev)
  elapsed = sd.max_over_ranks(mine, dev)
  # after the timed region: eight more steps with HIP events around the sweep kernel (BatchedSimulator.sweep_events) --
  # the kernel's own time in this regime (a smooth policy: ~2 sweeps per step) and the HBM rate it sustains there
  env.sim.sweep_events = []
  for _ in range(8):
    ts = step_once(obs)
    obs = ts.observation
  torch.cuda.synchronize(dev)
  sweep_ms = float(np.mean([a.elapsed_time(b) for a, b in env.sim.sweep_events]))
  env.sim.sweep_events = None
  gather_ms, n_gathered = 0.0, B
  if distributed:
    g0 = time.perf_counter()
    n_gathered = int(sd.gather_returns(returns, world * B).numel())
    torch.cuda.synchronize(dev)
    gather_ms = (time.perf_counter() - g0) * 1e3
  result = None
  if rank == 0:
    li = env.sim.launch_info
    env_steps_per_s = world * B * K / elapsed
    achieved = li["algorithmic_bytes_per_env_step"] * B * K / elapsed / 1e9
    result = {
        "metric": "zone-updates/sec (env-steps/sec x 9 zones) INCLUDING policy inference, batch=64k buildings per GPU",
        "value": env_steps_per_s * Z, "unit": "zone-updates/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic", "env_steps_per_s": env_steps_per_s,
        "return_gather_ms": gather_ms, "gathered_returns": n_gathered,
        "rccl_ranks": pf["ranks"], "preflight": pf, "per_rank_ms_per_step": per_rank,
        "config": {"workload": "BASELINE.json configs[4]: configs[1]'s batch per GPU driven through BatchedEnvironment.step() "
                               "by a SAC-shaped actor (2 x 128 MLP, tanh-squashed Gaussian) on the environment's GPU",
                   "buildings_per_gpu": B, "grid": [H, Wd], "zones": Z, "policy": "MLP %d-128-128-4, fp32" % O,
                   "mean_return_per_step": float(returns.mean()) / K,
                   "mean_sweeps_per_env_step_last": float(env.info[:, 4].mean()) if args.check_buildings else None,
                   **({"parity_vs_oracle": check.verify()} if check else {}),
                   "parallelism": f"{world} x (building shard + its own actor), no per-step collective", "launch": li},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                     "kernel": _ffi.SWEEP_KERNELS.get(li.get("kernel", -1), "?"),
                     "sweep_kernel_ms": sweep_ms,   # eight profiled steps after the timed region
                     "hbm_TBps_real": li["state_bytes_per_env_step"] * B / (sweep_ms * 1e-3) / 1e12,   # the state, read once and written once
                     "sweep_kernel_frac": li["algorithmic_bytes_per_env_step"] * B / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                     "note": "over the whole env step including the policy network (a smooth policy needs fewer "
                             "Gauss-Seidel sweeps per step than random actions)"}}
    if emit:
      print(json.dumps(result))
  env.close()
  if emit and distributed:
    dist.destroy_process_group()
  return result


LARGE_PLANS = [("299x401 / 126 zones", (14, 9), (20, 43), 1536, 6, "k_sweep_stream"),   # beyond one CU: the grid streams from HBM / L2
               ("205x89 / 40 zones", (10, 4), (19, 20), 4096, 5, "k_sweep_band")]          # 203 rows inside the ring: four wavefronts per building


def large_plans_leg(dev, n_twins: int = 8, steps: int = 6, warmup: int = 2) -> dict:
  """The kernels no BASELINE config runs -- k_sweep_stream (floor plans beyond one CU: the SB1 blobs' scale, which the
  reference's repository no longer ships) and k_sweep_band (131-258 rows) -- on one synthetic plan each through
  `BatchedEnvironment.step()`: rate, sweep-kernel time and `n_twins` oracle twins (sweep counts EQUAL, zone temperatures
  within 1e-8 K), so that the driver's record has a parity + rate line for them too.  Bounded: a few seconds."""
  out = {}
  for name, rooms, shape, B, kern_id, kern in LARGE_PLANS:
    plan = FloorPlan.from_file_input(rectangular_floor_plan(rooms, shape), Materials.sb1(), 10.0, 300.0)
    env = BatchedEnvironment(plan, B, device=dev.index, holiday_calendar="us", collect_info=True, num_days_in_episode=3)
    H, Wd = plan.shape
    rs = np.random.RandomState(7)
    t_init = np.clip(294.0 + rs.randn(B), 285.0, 305.0)
    env.reset()
    env.sim.reset(temps=torch.tensor(t_init, dtype=torch.float64, device=dev)[:, None].expand(B, H * Wd).contiguous())
    check = TwinCheck(env, plan, t_init[:n_twins], n_twins)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    acts = torch.rand((warmup + steps, B, 2), generator=gen, device=dev, dtype=torch.float32) * 2.0 - 1.0
    env.sim.sweep_events = []
    sweeps = []
    torch.cuda.synchronize(dev)
    for t in range(warmup + steps):
      if t == warmup:
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
      si = env.make_step_in(env.current_simulation_timestamp)
      env.step(acts[t])
      check.record(si, acts[t])
      sweeps.append(env.info[:, 4].mean())
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    ms = float(np.mean([a.elapsed_time(b) for a, b in env.sim.sweep_events[warmup:]]))
    li = env.sim.launch_info
    sw = float(torch.stack(sweeps[warmup:]).mean())
    out[name] = {"kernel": _ffi.SWEEP_KERNELS.get(li.get("kernel", -1), "?"), "expected_kernel": kern, "buildings": B, "grid": [H, Wd],
                 "zones": env.sim.Z, "steps": steps, "warmup": warmup, "ms_per_step": wall / steps * 1e3,
                 "zone_updates_per_s": B * env.sim.Z * steps / wall, "sweep_kernel_ms": ms, "mean_sweeps_per_env_step": sw,
                 "cell_sweeps_per_s": B * H * Wd * sw / (ms * 1e-3),
                 "roofline_frac": li["algorithmic_bytes_per_env_step"] * B / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                 "waves_per_building": li["waves_per_building"], "parity_vs_oracle": check.verify()}
    assert li.get("kernel") == kern_id, (name, li)
    env.close()
  return out


def also_legs(args, dev, pf) -> dict:
  """The default run's extra legs (after the headline's timed region; nothing inside it changes): a bounded run
  of BASELINE.json configs[2] (`--config mixed`) and configs[4] (`--config policy`) at their own batch sizes, so
  that the driver's record carries them too.  Each leg is the same function `--config ...` runs, with a handful
  of oracle twins per class as its in-run parity check; a leg that fails reports its error instead of taking
  the headline down."""
  import copy
  import traceback
  out = {}
  for name, fn, twins in (("mixed", mixed_config, 8), ("policy", policy_config, 32)):
    a = copy.copy(args)
    a.steps, a.warmup, a.check_buildings, a.config = 24, 12, twins, name
    t0 = time.perf_counter()
    try:
      r = fn(a, ctx=(0, dev, False, pf), emit=False)
      leg = {"value": r["value"], "unit": r["unit"], "ms_per_step": r["ms_per_step"], "steps": r["steps"],
             "warmup": r["warmup"], "workload": r["config"]["workload"], "roofline_frac": r["roofline"]["frac"]}
      if name == "mixed":
        leg["classes"] = {k: {f: v[f] for f in ("kernel", "buildings", "zones", "mean_sweeps_per_env_step", "step_ms_in_round",
                                                "sweep_kernel_ms_alone", "roofline_frac_alone", "parity_vs_oracle") if f in v}
                          for k, v in r["config"]["classes"].items()}
      else:
        leg["kernel"] = r["roofline"]["kernel"]
        leg["sweep_kernel_ms"] = r["roofline"].get("sweep_kernel_ms")
        leg["hbm_TBps_real"] = r["roofline"].get("hbm_TBps_real")
        leg["mean_sweeps_per_env_step"] = r["config"].get("mean_sweeps_per_env_step_last")
        leg["parity_vs_oracle"] = r["config"].get("parity_vs_oracle")
        leg["mean_return_per_step"] = r["config"]["mean_return_per_step"]
    except Exception as exc:   # noqa: BLE001 -- the headline line must still be printed
      leg = {"error": f"{type(exc).__name__}: {exc}", "trace": traceback.format_exc()[-600:]}
    leg["leg_wall_s"] = time.perf_counter() - t0
    out[name] = leg
  t0 = time.perf_counter()
  try:   # the kernels beyond the BASELINE configs' plans: one bounded parity + rate line each
    leg = large_plans_leg(dev)
  except Exception as exc:   # noqa: BLE001
    leg = {"error": f"{type(exc).__name__}: {exc}", "trace": traceback.format_exc()[-600:]}
  leg["leg_wall_s"] = time.perf_counter() - t0
  out["large_plans"] = leg
  return out


def main() -> None:
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=24)
  ap.add_argument("--warmup", type=int, default=12)
  ap.add_argument("--buildings", type=int, default=65536, help="buildings PER GPU")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--check-buildings", type=int, default=0,
                  help="--config mixed / policy: re-step the first N buildings (per class) on CPU-oracle twins and report the agreement")
  ap.add_argument("--iteration-limit", type=int, default=100,
                  help="Simulator.iteration_limit (100 = the reference's SB1 value; 1 is used only to "
                       "calibrate the PMC byte counters on a known traffic pattern)")
  ap.add_argument("--config", choices=("replicated", "mixed", "policy"), default="replicated",
                  help="replicated: BASELINE.json configs[1] (the metric's configuration, default); "
                       "mixed: configs[2], three floor-plan classes on one GPU; policy: configs[4], the "
                       "replicated batch driven by a SAC-shaped actor on the same GPU(s)")
  ap.add_argument("--through-env-api", action="store_true",
                  help="time BatchedEnvironment.step() itself (the public API) instead of the three phase launches; "
                       "roofline.avg_kernel_ms is then the whole step's GPU time")
  ap.add_argument("--no-also", action="store_true",
                  help="skip the default run's extra legs (bounded --config mixed and --config policy runs after the headline, "
                       "reported under \"also\")")
  ap.add_argument("--stub-step", action="store_true",
                  help="developer / CPU test: launcher, barrier and return-gather plumbing with a stub step (gloo)")
  args = ap.parse_args()
  launch_ranks_if_needed(args)
  if args.stub_step:
    return stub_rank(args)
  if args.config == "mixed":
    mixed_config(args)
    return
  if args.config == "policy":
    policy_config(args)
    return

  rank, dev, distributed, pf = start_rank(args)
  local_rank, world = dev.index, pf["ranks"]
  if distributed:
    import torch.distributed as dist

  B, K, W = args.buildings, args.steps, args.warmup
  plan = r9_plan()
  env = BatchedEnvironment(plan, B, device=local_rank, holiday_calendar="us", collect_info=True,
                           num_days_in_episode=3, config=SimConfig(iteration_limit=args.iteration_limit))
  H, Wd = plan.shape
  Z = env.sim.Z
  rs = np.random.RandomState(sd.shard_seed(7, rank))
  t_init = np.clip(294.0 + rs.randn(B), 285.0, 305.0)
  init = torch.tensor(t_init, dtype=torch.float64, device=dev)[:, None].expand(B, H * Wd).contiguous()
  env.reset()
  env.sim.reset(temps=init)
  gen = torch.Generator(device=dev)
  gen.manual_seed(sd.shard_seed(1234, rank))
  total = W + K
  actions = torch.rand((total, B, 2), generator=gen, device=dev, dtype=torch.float32) * 2.0 - 1.0
  returns = torch.zeros((B,), dtype=torch.float32, device=dev)

  def barrier():
    sd.barrier(dev)
    torch.cuda.synchronize(dev)

  def new_events(n, per_step):
    return [[torch.cuda.Event(enable_timing=True) for _ in range(per_step)] for _ in range(n)]

  def one_step(t, e):
    """sb_step == its three launches (sb_step_phases 1|2|4).  The sweep kernel, the dominant
    one, is bracketed by HIP events on the stream it is launched on; with four events also
    the two small launches around it."""
    si = env.make_step_in(env.current_simulation_timestamp)
    args = (actions[t], si, env._obs, env._reward, env._info)
    if len(e) == 4:
      e[2].record()
    env.sim.step(*args, phases=1)     # k_pre: thermostats, VAV, demand
    e[0].record()
    env.sim.step(*args, phases=2)     # sweep kernel
    e[1].record()
    env.sim.step(*args, phases=4)     # k_post: reward, observation
    if len(e) == 4:
      e[3].record()
    env._prev_thermostat_ts = env._now
    env._now = env._now + env._step_interval

  if args.through_env_api:   # the public API itself: BatchedEnvironment.step() (host inputs, sb_step, TimeStep); the events bracket the whole step
    def one_step(t, e):  # noqa: F811
      if len(e) == 4:
        e[2].record()
      e[0].record()
      env.step(actions[t])
      e[1].record()
      if len(e) == 4:
        e[3].record()

  wev, ev = new_events(W, 4), new_events(K, 2)
  for t in range(W):
    one_step(t, wev[t])
    returns += env._reward
  nsw = torch.zeros((K, B), dtype=torch.float32, device=dev)
  barrier()
  t0 = time.perf_counter()
  for t in range(K):
    one_step(W + t, ev[t])
    returns += env._reward
    nsw[t].copy_(env._info[:, 4])
  barrier()
  elapsed = time.perf_counter() - t0
  sweeps = nsw.double().sum()
  kernel_ms = [e[0].elapsed_time(e[1]) for e in ev]
  warm_ms = [e[0].elapsed_time(e[1]) for e in wev]
  # the two small launches are timed on the warm-up steps only (fewer event packets in the timed loop)
  pre_ms = float(np.mean([e[2].elapsed_time(e[0]) for e in wev])) if W else None
  post_ms = float(np.mean([e[1].elapsed_time(e[3]) for e in wev])) if W else None

  per_rank = sd.all_ranks(elapsed / K * 1e3, dev)   # every rank's own ms per step (the line's value uses the slowest)
  per_rank_kernel = sd.all_ranks(float(np.mean(kernel_ms)), dev)
  elapsed = sd.max_over_ranks(elapsed, dev)
  gather_ms, n_gathered = 0.0, B
  if distributed:
    torch.cuda.synchronize(dev)
    g0 = time.perf_counter()
    all_returns = sd.gather_returns(returns, world * B)   # end-of-rollout gather (RCCL / xGMI)
    torch.cuda.synchronize(dev)
    gather_ms = (time.perf_counter() - g0) * 1e3
    n_gathered = int(all_returns.numel())
    assert n_gathered == world * B

  steady = None
  if world == 1 and not args.no_also and not args.through_env_api and args.iteration_limit == 100:
    # the same loop in its steady state: the timed window above is a transient -- steps W .. W + K after
    # a reset to per-building-uniform temperatures, sweeps per step still falling -- so: on to 100 steps since the reset
    # (untimed), then 24 timed steps, the same events
    n_more, n_t = max(0, 100 - total), 24
    acts2 = torch.rand((n_more + n_t, B, 2), generator=gen, device=dev, dtype=torch.float32) * 2.0 - 1.0
    actions = torch.cat([actions, acts2], dim=0)
    sev = new_events(n_t, 2)
    scratch = new_events(1, 2)[0]
    for t in range(n_more):
      one_step(total + t, scratch)
    nsw2 = torch.zeros((n_t, B), dtype=torch.float32, device=dev)
    barrier()
    s0 = time.perf_counter()
    for t in range(n_t):
      one_step(total + n_more + t, sev[t])
      nsw2[t].copy_(env._info[:, 4])
    barrier()
    s_el = time.perf_counter() - s0
    s_ms = [e[0].elapsed_time(e[1]) for e in sev]
    li_ = env.sim.launch_info
    steady = {"steps_since_reset_before_timing": total + n_more, "steps": n_t, "ms_per_step": s_el / n_t * 1e3,
              "value": B * Z * n_t / s_el, "unit": "zone-updates/s",
              "mean_sweeps_per_env_step": float(nsw2.double().mean()), "sweep_kernel_ms": float(np.mean(s_ms)),
              "roofline_frac": li_["algorithmic_bytes_per_env_step"] * B / (float(np.mean(s_ms)) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
              "hbm_TBps_real": li_["state_bytes_per_env_step"] * B / (float(np.mean(s_ms)) * 1e-3) / 1e12}
    actions = actions[:total]

  if rank == 0:
    li = env.sim.launch_info
    env_steps_per_s = world * B * K / elapsed
    avg_kernel_s = float(np.mean(kernel_ms)) * 1e-3
    alg_bytes = li["algorithmic_bytes_per_env_step"] * B
    achieved = alg_bytes / avg_kernel_s / 1e9
    result = {
        "metric": "zone-updates/sec (env-steps/sec x 9 zones) at batch=64k buildings per GPU",
        "value": env_steps_per_s * Z, "unit": "zone-updates/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "env_steps_per_s": env_steps_per_s,
        "return_gather_ms": gather_ms, "gathered_returns": n_gathered,   # end-of-rollout all_gather, outside the timed region
        "rccl_ranks": pf["ranks"], "preflight": pf,                      # what sbsim_amd.distributed.preflight saw before the run
        "per_rank_ms_per_step": per_rank, "per_rank_sweep_kernel_ms": per_rank_kernel,
        "timed_through": "BatchedEnvironment.step()" if args.through_env_api else "BatchedSimulator.step(phases=PRE | SWEEP | POST) with HIP events around the sweep kernel",
        "config": {"workload": "BASELINE.json configs[1]: 64k replicated SB1-physics buildings on floor plan R9 "
                               "(68x98 CVs, 9 zones), random setpoint actions, sinusoid weather",
                   "buildings_per_gpu": B, "grid": [H, Wd], "zones": Z,
                   "mean_sweeps_per_env_step": float(sweeps.item()) / (B * K),
                   "parallelism": f"{world} x independent building shards, no data-path collective"
                                  + (" (SBSIM_BENCH_SHARE_GPU: ranks share a device, gloo -- a plumbing run, not a scaling number)" if sd.share_gpu() else ""),
                   "return_gather_ms": gather_ms, "launch": li},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                     "kernel": _ffi.SWEEP_KERNELS.get(li.get("kernel", -1), "?"),
                     "avg_kernel_ms": avg_kernel_s * 1e3,
                     # the step's two small launches around it (device algebra, reward/observation)
                     "avg_pre_kernel_ms": pre_ms, "avg_post_kernel_ms": post_ms,
                     # rocprofv3 --stats averages over EVERY sweep launch of the command (warm-up
                     # steps run more sweeps: the start state is not an equilibrium); same quantity:
                     "avg_kernel_ms_all_launches_incl_warmup": float(np.mean(warm_ms + kernel_ms)),
                     "kernel_ms_timed": [round(x, 4) for x in kernel_ms],
                     # sweeps-normalised cost (the timed window is a transient: sweeps per step fall from step to
                     # step): what one of the chip's 1,024 SIMDs spends per building and sweep, everything included
                     "simd_us_per_building_sweep": avg_kernel_s * 1e6 * 1024 / (B * (float(sweeps.item()) / (B * K))),
                     "cell_sweeps_per_s": float(sweeps.item()) * H * Wd / (avg_kernel_s * K),
                     "algorithmic_bytes_per_launch": alg_bytes,
                     "state_bytes_per_launch_fp64": li["state_bytes_per_env_step"] * B},
    }
    tr = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tr):   # PMC passes of this same command (tools/collect_profiles.sh)
      with open(tr) as fh:
        t = json.load(fh)
      src_now = kernel_source_sha256()
      # the counters were taken on a particular kernel and state layout: a profile of another one is not this run's traffic
      rest = (8 * 4 + 4 * 2) * Z + 16 * 16 + 8 + 4 * env.sim.O + 4
      state_bytes = (li["state_bytes_per_env_step"] - rest) // 2
      same_kernel = _ffi.SWEEP_KERNELS.get(li.get("kernel", -1), "?") in str(t.get("kernel", ""))
      same_source = t.get("kernel_source_sha256") == src_now
      if same_kernel and same_source and t.get("state_bytes_per_building") == state_bytes and B == 65536:
        result["roofline"]["traffic"] = t.get("hbm_bytes_per_launch")
        result["roofline"]["traffic_source"] = t.get("source")
        # replayed from the committed PMC profile of this command (a run cannot read its own TCC counters);
        # the guards above tie it to this kernel's SOURCES (a hash of the files it is compiled from), state layout and batch
        result["roofline"]["traffic_measured_in_run"] = False
        result["roofline"]["traffic_profile_commit"] = t.get("commit")
        result["roofline"]["traffic_profile_kernel_source_sha256"] = src_now
      else:
        result["roofline"]["traffic_measured_in_run"] = False
        result["roofline"]["traffic_source"] = (f"profiles/traffic_latest.json is for kernel {t.get('kernel')!r} built from sources "
                                                f"{str(t.get('kernel_source_sha256'))[:12]} (this run: {src_now[:12]}), "
                                                f"{t.get('state_bytes_per_building')} state bytes per building, 65,536 buildings: not this run")
    if world == 1 and not args.no_cpu_baseline:
      nb_s = 4096
      # the CPU leg goes on past the GPU's K timed steps (same action distribution, same generator)
      # until its sample is a few seconds of wall time
      more = torch.rand((400, nb_s, 2), generator=gen, device=dev, dtype=torch.float32) * 2.0 - 1.0
      actions = torch.cat([actions[:, :nb_s], more], dim=0)
      acts_cpu = actions.cpu().numpy()
      base, grids, n_s, nb = cpu_baseline(env, plan, np.broadcast_to(t_init[:nb_s, None], (nb_s, H * Wd)).copy(),
                                          acts_cpu, W)
      # in-run parity: replay the same sample on the GPU and compare grids
      env2 = BatchedEnvironment(plan, nb, device=local_rank, holiday_calendar="us")
      env2.reset()
      env2.sim.reset(temps=init[:nb].contiguous())
      for t in range(n_s):
        env2.step(actions[t, :nb].contiguous())
      dT = float(np.abs(env2.sim.temps().cpu().numpy() - grids).max())
      env2.close()
      base["parity_max_abs_dT_K"] = dT
      result["cpu_baseline"] = base
  env.close()
  if rank == 0:
    if world == 1 and not args.no_also and not args.through_env_api and B == 65536:
      result["also"] = also_legs(args, dev, pf)   # after the headline (and its CPU leg): configs[2] and configs[4], bounded
      result["also"]["steady"] = steady
    print(json.dumps(result))
  if distributed:
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
