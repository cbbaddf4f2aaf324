"""NumPy model of the HIP sweep's schedule (test utility, CPU only).

Replays exactly the lane/band/step mapping of ``sweep()`` in
sbsim_amd/csrc/sbsim_hip.hip -- 64 lanes in lock-step, reads of a step before its writes
-- on top of the class tables from ``FloorPlan.compile``.  Comparing it with the oracle
validates, without a GPU, (1) that the skewed schedule reproduces the reference's
row-major in-place Gauss-Seidel order and (2) that the seven-term class-table update is
the reference's corner / edge / interior arithmetic."""
import numpy as np


def geometry(H, W):
  pitch = (W + 1) & ~1
  S = max(W, 64)
  nbands = (H + 63) // 64
  rows_last = H - (nbands - 1) * 64
  nsteps = (nbands - 1) * S + rows_last + W - 1
  magic = (1 << 32) // S + 1
  return pitch, S, nsteps, magic


def model_sweep(cp, E, P, t_amb, q_zone):
  """One sweep, in place on E [H*pitch]; returns max_delta.  P: [H*W] previous temps."""
  H, W = cp.H, cp.W
  pitch, S, nsteps, magic = geometry(H, W)
  NL = H * pitch
  coef = cp.class_coef
  g = coef[:, 5] * t_amb + coef[:, 6] * np.where(cp.class_zone >= 0,
                                                q_zone[np.maximum(cp.class_zone, 0)], 0.0)
  lane = np.arange(64)
  dmax = 0.0
  for d in range(nsteps):
    v = d - lane
    band = (np.maximum(v, 0).astype(np.uint64) * magic) >> 32
    band = band.astype(np.int64)
    assert np.array_equal(band, np.maximum(v, 0) // S)
    y = v - band * S
    r = band * 64 + lane
    act = (v >= 0) & (y < W) & (r < H)
    if not act.any():
      continue
    r, y = r[act], y[act]
    li = r * pitch + y
    c = cp.cell_class[r * W + y].astype(np.int64)
    U = E[np.maximum(li - pitch, 0)]
    D = E[np.minimum(li + pitch, NL - 1)]
    L = E[np.maximum(li - 1, 0)]
    R = E[np.minimum(li + 1, NL - 1)]
    old = E[li]
    nv = coef[c, 4] * P[r * W + y] + g[c]
    nv = nv + coef[c, 1] * D
    nv = nv + coef[c, 3] * R
    nv = nv + coef[c, 2] * L
    nv = nv + coef[c, 0] * U
    dmax = max(dmax, float(np.abs(nv - old).max()))
    E[li] = nv
  return dmax


def model_fd_timestep(cp, temp, t_amb, q_zone, thr, iter_limit):
  """simulator.py:318-371 on a [H, W] grid; returns (new grid, sweeps)."""
  H, W = cp.H, cp.W
  pitch = geometry(H, W)[0]
  E = np.zeros(H * pitch)
  E.reshape(H, pitch)[:, :W] = temp
  P = np.asarray(temp, dtype=np.float64).reshape(-1)
  n = 0
  for _ in range(iter_limit):
    md = model_sweep(cp, E, P, t_amb, q_zone)
    n += 1
    if md <= thr:
      break
  return E.reshape(H, pitch)[:, :W].copy(), n
