"""NumPy model of the HIP sweep's schedule (test utility, CPU only).

Replays exactly the lane/band/step mapping of ``sweep()`` in
sbsim_amd/csrc/step_lds.hip -- 64 lanes in lock-step, reads of a step before its writes
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


# ----------------------------------------------------------------------------------------
# Model of the register-resident sweep (sbsim_amd/csrc/step_reg.hip) and of its launch
# planning (plan_reg in sbsim_hip.hip): trim box, cyclic-skewed slots, DPP neighbours, lane
# predicates, the tail-row recurrence (mode 3) and the two-wavefront seam protocol with its
# chunk lag and 2-step-early seam reads (mode 2).  The two-wavefront model can run wave 1 as
# early as the progress counter allows ("tight") or only after wave 0 finished ("late"): every
# admissible interleaving must give the same grid.
K_LOOK = 2
REG_SLOTS = (32, 66, 96)


def seam_lag(lw0):
  return (lw0 + 8) // 8 + 1


class RegSweepModel:

  def __init__(self, cp, mode=None):
    self.cp = cp
    H, W = cp.H, cp.W
    coef = cp.class_coef
    amb = np.array([(coef[c, :5] == 0).all() and coef[c, 5] == 1.0 and coef[c, 6] == 0 for c in range(cp.n_classes)])
    cls = cp.cell_class.reshape(H, W).astype(np.int64)
    inside = ~amb[cls]
    xs, ys = np.nonzero(inside.any(axis=1))[0], np.nonzero(inside.any(axis=0))[0]
    self.x0, self.x1, self.y0, self.y1 = xs[0], xs[-1], ys[0], ys[-1]
    self.Hs, self.Ws = self.x1 - self.x0 + 1, self.y1 - self.y0 + 1
    self.cls = cls[self.x0:self.x1 + 1, self.y0:self.y1 + 1]
    # nothing may couple to a cell outside the trim box
    assert (coef[self.cls[0, :], 0] == 0).all() and (coef[self.cls[-1, :], 1] == 0).all()
    assert (coef[self.cls[:, 0], 2] == 0).all() and (coef[self.cls[:, -1], 3] == 0).all()
    zone_of = np.full(H * W, -1)
    for z in range(cp.Z):
      zone_of[cp.zone_cells[cp.zone_off[z]:cp.zone_off[z + 1]]] = z
    zone_of = zone_of.reshape(H, W)[self.x0:self.x1 + 1, self.y0:self.y1 + 1]
    if mode is None:
      if self.Hs <= 64:
        mode = 1
      elif self.Hs <= 66 and (zone_of[64:] < 0).all():
        mode = 3
      else:
        mode = 2
    self.mode = mode
    self.NR = min(s for s in REG_SLOTS if s >= self.Ws)
    self.pad = cp.n_classes
    self.coef = np.vstack([coef, [[0, 0, 0, 0, 1.0, 0, 0, 0]]])   # pad class: T' = Tprev
    NR, Hs = self.NR, self.Hs
    if mode == 2:
      best, best_slots = -1, 1 << 30
      for a0 in range(max(Hs - 64, 1), min(64, Hs - 1) + 1):
        n0, n1 = (NR + a0 - 1 + 7) // 8, (NR + (Hs - a0) - 1 + 7) // 8
        slots = max(n0, seam_lag(a0) + n1)
        if slots < best_slots or (slots == best_slots and abs(2 * a0 - Hs) < abs(2 * best - Hs)):
          best, best_slots = a0, slots
      self.lw, self.l0, self.rowbase = [best, Hs - best], [64 - best, 0], [0, best]
      self.lag = seam_lag(best)
    else:
      rows = min(Hs, 64)
      self.lw, self.l0, self.rowbase, self.lag = [rows], [0], [0], 0
    self.nch = [(NR + lw - 1 + 7) // 8 for lw in self.lw]
    self.T = Hs - 64 if mode == 3 else 0

  # -- helpers ---------------------------------------------------------------------------
  def _class(self, R, col):
    ok = (R >= 0) & (R < self.Hs) & (col >= 0) & (col < self.Ws)
    return np.where(ok, self.cls[np.clip(R, 0, self.Hs - 1), np.clip(col, 0, self.Ws - 1)], self.pad)

  def _load(self, grid):
    """grid [Hs, Ws] -> per wave e[64][NR] (cyclic skew), tail rows [T][NR]."""
    NR = self.NR
    es = []
    for w in range(len(self.lw)):
      e = np.zeros((64, NR))
      for lane in range(64):
        lp = lane - self.l0[w]
        if 0 <= lp < self.lw[w]:
          R = self.rowbase[w] + lp
          cols = np.arange(self.Ws)
          e[lane, (cols + lp) % NR] = grid[R, :]
      es.append(e)
    tail = np.zeros((self.T, NR))
    for t in range(self.T):
      tail[t, :self.Ws] = grid[64 + t, :]
    return es, tail

  def _store(self, es, tail):
    NR = self.NR
    grid = np.zeros((self.Hs, self.Ws))
    for w in range(len(self.lw)):
      for lane in range(64):
        lp = lane - self.l0[w]
        if 0 <= lp < self.lw[w]:
          cols = np.arange(self.Ws)
          grid[self.rowbase[w] + lp, :] = es[w][lane, (cols + lp) % NR]
    for t in range(self.T):
      grid[64 + t, :] = tail[t, :self.Ws]
    return grid

  def _step(self, w, e, A, g, D, seam_u, seam_d):
    """One wavefront step: all 64 lanes, slot D mod NR.  seam_u / seam_d: the value the DPP
    `old` operand carries into lane 0 (shr) / lane 63 (shl); returns max |delta| of the step."""
    NR = self.NR
    r, rm, rp = D % NR, (D - 1) % NR, (D + 1) % NR
    lane = np.arange(64)
    lp = lane - self.l0[w]
    valid = (lp >= 0) & (lp < self.lw[w])
    R = self.rowbase[w] + lp
    col = D - lp
    c = np.where(valid, self._class(R, col), self.pad)
    co = self.coef[c]
    U = np.empty(64); U[1:] = e[:-1, rm]; U[0] = seam_u
    Dn = np.empty(64); Dn[:-1] = e[1:, rp]; Dn[63] = seam_d
    nv = co[:, 1] * Dn + A[:, r]
    nv = co[:, 3] * e[:, rp] + nv
    nv = co[:, 2] * e[:, rm] + nv
    nv = co[:, 0] * U + nv
    act = valid & (col >= 0) & (col < NR)
    sel = np.where(act, nv, e[:, r])
    d = float(np.abs(sel - e[:, r]).max())
    e[:, r] = sel
    return d

  def _a_pass(self, e, w, g):
    NR = self.NR
    lane = np.arange(64)
    lp = lane - self.l0[w]
    valid = (lp >= 0) & (lp < self.lw[w])
    A = np.zeros((64, NR))
    for j in range(NR):
      col = (j - lp) % NR
      c = np.where(valid, self._class(self.rowbase[w] + lp, col), self.pad)
      A[:, j] = self.coef[c, 4] * e[:, j] + g[c]
    return A

  # -- one FD time step ------------------------------------------------------------------
  def fd_timestep(self, temp, t_amb, q_zone, thr, iter_limit, schedule="tight"):
    cp, NR = self.cp, self.NR
    full = np.array(temp, dtype=np.float64).reshape(cp.H, cp.W)
    ring_mask = np.ones((cp.H, cp.W), bool)
    ring_mask[self.x0:self.x1 + 1, self.y0:self.y1 + 1] = False
    ring = full[ring_mask]
    g = np.zeros(cp.n_classes + 1)
    g[:-1] = cp.class_coef[:, 5] * t_amb + cp.class_coef[:, 6] * np.where(
        cp.class_zone >= 0, q_zone[np.maximum(cp.class_zone, 0)], 0.0)
    es, tail = self._load(full[self.x0:self.x1 + 1, self.y0:self.y1 + 1])
    As = [self._a_pass(es[w], w, g) for w in range(len(es))]
    tcls = np.array([[self._class(np.array(64 + t), np.array(c)) for c in range(NR)] for t in range(self.T)], dtype=np.int64).reshape(self.T, NR)
    At = self.coef[tcls, 4] * tail + g[tcls] if self.T else None
    ring_d = max(abs(t_amb - ring.min()), abs(t_amb - ring.max())) if ring.size else 0.0
    n = 0
    if schedule == "rolling":
      assert self.mode in (1, 3) and NR > 63 + K_LOOK
      n = self._fd_rolling(es[0], tail, As[0], At, tcls, ring_d, thr, iter_limit)
      iter_limit = 0
    for it in range(iter_limit):
      if self.mode == 2:
        md = self._sweep_pair(es, As, schedule)
      else:
        md = 0.0
        row63 = np.zeros(NR)
        for D in range(self.nch[0] * 8):
          if D >= NR + 63:
            break
          sd = tail[0, D - 63] if (self.mode == 3 and 0 <= D - 63 < NR) else 0.0
          md = max(md, self._step(0, es[0], As[0], g, D, 0.0, sd))
        if self.mode == 3:
          row63 = es[0][63, (np.arange(NR) + 63) % NR].copy()
          md = max(md, self._tail_pass(tail, At, tcls, row63))
      if it == 0:
        md = max(md, ring_d)
      n += 1
      if md <= thr:
        break
    out = full.copy()
    out[ring_mask] = t_amb
    out[self.x0:self.x1 + 1, self.y0:self.y1 + 1] = self._store(es, tail)
    return out, n

  def _fd_rolling(self, e, tail, A, At, tcls, ring_d, thr, iter_limit):
    """Modes 1 and 3 with overlapped sweeps: period NR steps, lane l works on column (s - l) mod NR at
    global step s, so lanes 0..j start sweep k+1 during steps j = 0..62 of the period while
    lanes j+1..63 finish sweep k.  The stopping decision for sweep k falls after step 62 (and
    the tail pass); the speculative updates of sweep k+1 are undone from per-step backups."""
    NR = self.NR
    lane = np.arange(64)

    def step(D, first):
      r, rm, rp = D % NR, (D - 1) % NR, (D + 1) % NR
      col = (D - lane) if first else (D - lane) % NR
      c = self._class(lane, col)
      co = self.coef[c]
      U = np.empty(64); U[1:] = e[:-1, rm]; U[0] = 0.0
      Dn = np.empty(64); Dn[:-1] = e[1:, rp]
      tc = col[63]
      Dn[63] = tail[0, tc] if (self.T and 0 <= tc < NR) else 0.0
      nv = co[:, 1] * Dn + A[:, r]
      nv = co[:, 3] * e[:, rp] + nv
      nv = co[:, 2] * e[:, rm] + nv
      nv = co[:, 0] * U + nv
      act = (lane <= D) if first else np.ones(64, bool)
      sel = np.where(act, nv, e[:, r])
      d = np.abs(sel - e[:, r])
      old = e[:, r].copy()
      e[:, r] = sel
      return d, old

    valid = lane < self.lw[0]               # mode 1: lanes without a row run on a copy nobody stores
    dcur = np.zeros(64)
    for D in range(63):                       # ramp-up of sweep 0
      d, _ = step(D, True)
      dcur = np.maximum(dcur, d)
    n = 0
    while True:
      for D in range(63, NR):                 # every lane inside sweep n
        d, _ = step(D, False)
        dcur = np.maximum(dcur, d)
      dnext = np.zeros(64)
      bk = []
      for j in range(63):                     # lanes <= j: sweep n+1 (speculative); lanes > j: sweep n
        d, old = step(NR + j, False)
        bk.append(old)
        dcur = np.maximum(dcur, np.where(lane > j, d, 0.0))
        dnext = np.maximum(dnext, np.where(lane <= j, d, 0.0))
      row63 = e[63, (np.arange(NR) + 63) % NR].copy()
      md = float(dcur[valid].max())
      if self.T:
        md = max(md, self._tail_pass(tail, At, tcls, row63))
      if n == 0:
        md = max(md, ring_d)
      n += 1
      if md <= thr or n >= iter_limit:
        for j in range(63):
          e[:, j] = np.where(lane <= j, bk[j], e[:, j])
        return n
      dcur = dnext

  def _tail_pass(self, tail, At, tcls, row63):
    """Rows 64.. by the recurrence x_c = bL x_{c-1} + q_c (scan order of operations: the q's are
    complete before the recurrence starts; evaluated left to right here)."""
    NR, dmax = self.NR, 0.0
    for t in range(self.T):
      co = self.coef[tcls[t]]
      U = row63 if t == 0 else tail[t - 1]
      Dn = tail[t + 1] if t + 1 < self.T else np.zeros(NR)
      Rn = np.append(tail[t, 1:], 0.0)
      q = co[:, 0] * U + (co[:, 3] * Rn + (co[:, 1] * Dn + At[t]))
      x = np.zeros(NR)
      prev = 0.0
      for c in range(NR):
        prev = co[c, 2] * prev + q[c]
        x[c] = prev
      dmax = max(dmax, float(np.abs(x - tail[t]).max()))
      tail[t] = x
    return dmax

  def _sweep_pair(self, es, As, schedule):
    """Mode 2.  Wave 0 publishes its seam row per chunk and bumps the progress counter; wave 1
    may start chunk i once progress >= min(i + lag, nch0).  Seam values are read K_LOOK steps
    before they are used.  schedule 'tight': wave 1 runs each chunk at the earliest moment;
    'late': wave 0 finishes the whole sweep first."""
    NR = self.NR
    lw0 = self.lw[0]
    seamU = np.full(NR + 64, np.nan)   # new values of wave 0's last row, by column (NaN = not published)
    seamD = es[1][0, (np.arange(NR)) % NR].copy()  # old values of wave 1's first row (lane 0: slot c)
    seamD = np.concatenate([seamD, np.zeros(64)])
    md = 0.0
    prog0 = 0
    done1 = 0
    pre0, pre1 = {}, {}   # step -> seam value captured at prefetch time

    def fetch0(D):   # wave 0 reads old (row lw0, col D - (lw0-1)) for its lane 63
      c = D - (lw0 - 1)
      pre0[D] = seamD[c] if 0 <= c < NR else 0.0

    def fetch1(D):   # wave 1 reads new (row lw0-1, col D) for its lane 0
      pre1[D] = seamU[D] if 0 <= D < NR else 0.0

    def run_chunk0(ci):
      nonlocal md
      for D in range(8 * ci, 8 * ci + 8):
        if D >= NR + 63:
          break
        fetch0(D + K_LOOK)
        md = max(md, self._step(0, es[0], As[0], None, D, 0.0, pre0.pop(D)))
      for D in range(8 * ci, 8 * ci + 8):          # publish the edge row (lane 63)
        c = D - (lw0 - 1)
        if 0 <= c < NR and D < NR + 63:
          seamU[c] = es[0][63, D % NR]

    def run_chunk1(ci):
      nonlocal md
      for D in range(8 * ci, 8 * ci + 8):
        if D >= NR + 63:
          break
        fetch1(D + K_LOOK)
        v = pre1.pop(D)
        assert not np.isnan(v), ("wave 1 read a seam value before wave 0 published it", D)
        md = max(md, self._step(1, es[1], As[1], None, D, v, 0.0))
      for D in range(8 * ci, 8 * ci + 8):          # lane 0 publishes its new values
        if 0 <= D < NR:
          seamD[D] = es[1][0, D % NR]

    fetch0(0); fetch0(1)
    started1 = False
    nch0, nch1 = self.nch
    while prog0 < nch0 or done1 < nch1:
      need = min(done1 + self.lag, nch0)
      can1 = done1 < nch1 and prog0 >= need
      if can1 and (schedule == "tight" or prog0 >= nch0):
        if not started1:
          fetch1(0); fetch1(1)
          started1 = True
        run_chunk1(done1)
        done1 += 1
      else:
        run_chunk0(prog0)
        prog0 += 1
    return md
