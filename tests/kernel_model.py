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


# ----------------------------------------------------------------------------------------
# Model of sweep_fast (three-stage pipeline, DPP neighbours, rolling slots).  Mirrors the
# HIP code statement by statement; DPP wave_shr:1 / wave_shl:1 semantics as measured on
# gfx950 by tools/dpp_probe.hip (lane 0 / lane 63 keep their own value).
KCH = 8
KPAD = 8


def _shr1(x):
  out = x.copy()
  out[1:] = x[:-1]
  return out


def _shl1(x):
  out = x.copy()
  out[:-1] = x[1:]
  return out


class FastSweepModel:

  def __init__(self, cp):
    self.cp = cp
    H, W = cp.H, cp.W
    self.H, self.W = H, W
    self.pitch = (W + 1) & ~1
    self.NL = H * self.pitch
    self.nbands = (H + 63) // 64
    self.S = max(W + KCH, 64 + 2 * KCH if self.nbands > 1 else 64)
    self.magic = (1 << 32) // self.S + 1
    rows_last = H - (self.nbands - 1) * 64
    self.nsteps = (self.nbands - 1) * self.S + rows_last + W - 1
    self.fast = self.nbands == 1 or (self.S - 63 > 2 * KCH)
    self.multi = self.nbands > 1
    self.cls = np.zeros(H * W + 2 * KPAD + 8, dtype=np.int64)
    self.cls[KPAD:KPAD + H * W] = cp.cell_class
    ncls = cp.n_classes
    self.btab = np.zeros((ncls + 1, 4))
    self.btab[:ncls] = cp.class_coef[:, :4]
    self.lane = np.arange(64)

  def stage_g(self, E, Pg, ch, first):
    lane = self.lane
    v0 = ch * KCH - lane
    vl = v0 + KCH - 1
    band = (np.maximum(vl, 0) * self.magic) >> 32
    assert np.array_equal(band, np.maximum(vl, 0) // self.S)
    y0 = v0 - band * self.S
    r = band * 64 + lane
    row_ok = (vl >= 0) & (r < self.H)
    k = np.arange(KCH)[None, :]
    act = row_ok[:, None] & (y0[:, None] + k >= 0) & (y0[:, None] + k < self.W)
    ra = np.minimum(r, self.H - 1)
    y0a = np.clip(np.where(vl >= 0, y0, v0), -(KCH - 1) - 1, self.W)
    li0 = ra * self.pitch + y0a
    g0 = ra * self.W + np.clip(y0a, -(KCH - 1), self.W)
    c = self.cls[KPAD + g0[:, None] + k]
    if first:
      P = E[np.clip(li0[:, None] + k, 0, self.NL - 1)]
    else:
      P = Pg[KPAD + g0[:, None] + k]
    return dict(li0=li0, act=act, P=P, c=c)

  def load_slot(self, E, agtab, K, g, li0_after, l):
    c = np.where(g["act"][:, K], g["c"][:, K], self.cp.n_classes)
    b = self.btab[c]
    l["bU"][:, K], l["bD"][:, K], l["bL"][:, K], l["bR"][:, K] = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    l["A"][:, K] = agtab[c, 0] * g["P"][:, K] + agtab[c, 1]
    idx = li0_after if K == KCH - 1 else g["li0"] + K + 1
    l["Rn"][:, K] = E[np.clip(idx, 0, self.NL - 1)]
    ea = g["li0"] + K + np.where(self.lane == 0, -self.pitch, self.pitch)
    l["Ee"][:, K] = E[np.clip(ea, 0, self.NL - 1)]

  def update_slot(self, E, K, l, li0, act, st):
    U = _shr1(st["nv"])
    D = _shl1(l["Rn"][:, K])
    if self.multi:
      U = np.where(self.lane == 0, l["Ee"][:, K], U)
      D = np.where(self.lane == 63, l["Ee"][:, K], D)
    t = l["bD"][:, K] * D + l["A"][:, K]
    t = l["bR"][:, K] * l["Rn"][:, K] + t
    t = l["bL"][:, K] * st["nv"] + t
    nvn = l["bU"][:, K] * U + t
    a = act[:, K]
    if a.any():
      E[(li0 + K)[a]] = nvn[a]
      st["dmax"] = max(st["dmax"], float(np.abs(nvn - st["oldv"])[a].max()))
    st["oldv"] = l["Rn"][:, K].copy()
    st["nv"] = nvn

  def sweep(self, E, Pg, agtab, first):
    nch = (self.nsteps + KCH - 1) // KCH
    l = {n: np.zeros((64, KCH)) for n in ("A", "bU", "bD", "bL", "bR", "Rn", "Ee")}
    g0 = self.stage_g(E, Pg, 0, first)
    g1 = self.stage_g(E, Pg, 1, first)
    for K in range(KCH):
      self.load_slot(E, agtab, K, g0, g1["li0"], l)
    st = dict(nv=np.zeros(64), dmax=0.0, oldv=E[np.clip(g0["li0"], 0, self.NL - 1)].copy())
    li0, act = g0["li0"], g0["act"]
    ch = 0
    while ch < nch:
      g0 = self.stage_g(E, Pg, ch + 2, first)
      for K in range(KCH):
        self.update_slot(E, K, l, li0, act, st)
        self.load_slot(E, agtab, K, g1, g0["li0"], l)
      li0, act = g1["li0"], g1["act"]
      g1 = self.stage_g(E, Pg, ch + 3, first)
      for K in range(KCH):
        self.update_slot(E, K, l, li0, act, st)
        self.load_slot(E, agtab, K, g0, g1["li0"], l)
      li0, act = g0["li0"], g0["act"]
      ch += 2
    return st["dmax"]

  def fd_timestep(self, temp, t_amb, q_zone, thr, iter_limit):
    cp = self.cp
    H, W = self.H, self.W
    E = np.zeros(self.NL)
    E.reshape(H, self.pitch)[:, :W] = temp
    Pg = np.zeros(H * W + 2 * KPAD)
    Pg[KPAD:KPAD + H * W] = np.asarray(temp).reshape(-1)
    coef = cp.class_coef
    agtab = np.zeros((cp.n_classes + 1, 2))
    agtab[:cp.n_classes, 0] = coef[:, 4]
    agtab[:cp.n_classes, 1] = coef[:, 5] * t_amb + coef[:, 6] * np.where(
        cp.class_zone >= 0, q_zone[np.maximum(cp.class_zone, 0)], 0.0)
    n = 0
    for it in range(iter_limit):
      md = self.sweep(E, Pg, agtab, it == 0)
      n += 1
      if md <= thr:
        break
    return E.reshape(H, self.pitch)[:, :W].copy(), n
