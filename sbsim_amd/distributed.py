"""Multi-GPU: independent building shards, one process per GPU, one gather at rollout end.

Buildings never exchange data, so there is no per-step collective.
Each rank owns a contiguous block of the global batch; after a rollout the per-building
returns are all-gathered (RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU
tests).  Nothing here touches HIP, so it is covered by world_size-2 gloo tests."""
from __future__ import annotations

import os
import socket
import zlib
from typing import List, Tuple

import torch
import torch.distributed as dist


def env_rank_world() -> Tuple[int, int, int]:
  """(rank, local_rank, world_size) from the torchrun environment (defaults: single process)."""
  return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
          int(os.environ.get("WORLD_SIZE", "1")))


def init_process_group(backend: str = "nccl") -> bool:
  """Initialises torch.distributed from the torchrun environment; False when single-process."""
  rank, _, world = env_rank_world()
  if world <= 1:
    return False
  os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
  os.environ.setdefault("MASTER_PORT", "29500")
  if not dist.is_initialized():
    dist.init_process_group(backend, rank=rank, world_size=world)
  return True


def shard_range(n_buildings_total: int, rank: int, world_size: int) -> Tuple[int, int]:
  """Contiguous block partition [lo, hi) of the global building index space; the first
  ``n % world`` ranks take one extra building."""
  if not 0 <= rank < world_size:
    raise ValueError("rank out of range")
  base, extra = divmod(n_buildings_total, world_size)
  lo = rank * base + min(rank, extra)
  return lo, lo + base + (1 if rank < extra else 0)


def class_shard_ranges(class_totals, rank: int, world_size: int) -> List[Tuple[int, int]]:
  """Mixed floor-plan classes: every class is block-partitioned on its own, so each rank
  holds the same class mix (load balance: the classes need very different numbers of sweeps).  Returns the
  rank's [lo, hi) inside every class's own global index space."""
  return [shard_range(int(n), rank, world_size) for n in class_totals]


def shard_seed(base_seed: int, rank: int) -> int:
  """Per-rank RNG seed of the synthetic workload (1234 + rank)."""
  return base_seed + rank


def share_gpu() -> bool:
  """Developer / test mode (SBSIM_BENCH_SHARE_GPU=1): the ranks of one node share the visible GPUs
  (rank -> device local_rank mod device count) and talk over gloo with host-staged collectives -- RCCL
  refuses two ranks on one device.  Exercises the N > 1 code path of bench.py on a one-GPU box."""
  return os.environ.get("SBSIM_BENCH_SHARE_GPU") == "1"


def device_index(local_rank: int) -> int:
  if share_gpu() and torch.cuda.is_available():
    return local_rank % torch.cuda.device_count()
  return local_rank


def backend_for_gpu() -> str:
  return "gloo" if share_gpu() else "nccl"   # "nccl" is RCCL on ROCm


def _host_staged(t: torch.Tensor) -> bool:
  return t.is_cuda and dist.get_backend() == "gloo"


def gather_returns(local_returns: torch.Tensor, n_buildings_total: int) -> torch.Tensor:
  """All-gathers the per-building episode returns into global building order.

  ``local_returns``: [B_local] on the rank's device.  Returns [n_buildings_total] on every
  rank.  Ragged shards (n not divisible by world) are padded to the largest shard for the
  collective and trimmed afterwards."""
  if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
    assert local_returns.numel() == n_buildings_total
    return local_returns.clone()
  world, rank = dist.get_world_size(), dist.get_rank()
  sizes = [shard_range(n_buildings_total, r, world) for r in range(world)]
  widest = max(hi - lo for lo, hi in sizes)
  lo, hi = sizes[rank]
  assert local_returns.numel() == hi - lo, (local_returns.numel(), lo, hi)
  staged = _host_staged(local_returns)
  src = local_returns.cpu() if staged else local_returns
  padded = torch.zeros((widest,), dtype=src.dtype, device=src.device)
  padded[: hi - lo] = src
  parts: List[torch.Tensor] = [torch.empty_like(padded) for _ in range(world)]
  dist.all_gather(parts, padded)
  out = torch.cat([p[: h - l] for p, (l, h) in zip(parts, sizes)])
  return out.to(local_returns.device) if staged else out


def gather_returns_by_class(local_returns: torch.Tensor, class_totals) -> torch.Tensor:
  """The end-of-rollout gather for a batch mixed over floor-plan classes.  ``local_returns``: the rank's
  buildings in class order (its shard of class 0, then its shard of class 1, ...: the layout of
  ``MixedBatchedEnvironment``).  Returns [sum(class_totals)] on every rank in GLOBAL order: class-major, the
  buildings of a class in their global order (rank 0's shard of the class first).  One all_gather, as for
  one class."""
  class_totals = [int(n) for n in class_totals]
  total = sum(class_totals)
  if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
    assert local_returns.numel() == total
    return local_returns.clone()
  world, rank = dist.get_world_size(), dist.get_rank()
  spans = [class_shard_ranges(class_totals, r, world) for r in range(world)]   # [rank][class] -> (lo, hi)
  sizes = [sum(hi - lo for lo, hi in sp) for sp in spans]
  assert local_returns.numel() == sizes[rank], (local_returns.numel(), sizes[rank])
  staged = _host_staged(local_returns)
  src = local_returns.cpu() if staged else local_returns
  padded = torch.zeros((max(sizes),), dtype=src.dtype, device=src.device)
  padded[: sizes[rank]] = src
  parts: List[torch.Tensor] = [torch.empty_like(padded) for _ in range(world)]
  dist.all_gather(parts, padded)
  out = []
  for k in range(len(class_totals)):
    for r in range(world):
      off = sum(hi - lo for lo, hi in spans[r][:k])
      lo, hi = spans[r][k]
      out.append(parts[r][off: off + hi - lo])
  full = torch.cat(out)
  return full.to(local_returns.device) if staged else full


def barrier(device: torch.device | None = None) -> None:
  """dist.barrier() that names the rank's device under RCCL (without device_ids the first barrier of a
  process binds NCCL/RCCL to "the current device" with a warning, and to the wrong one if set_device was
  forgotten); a plain barrier on gloo; nothing when single-process."""
  if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
    return
  if dist.get_backend() == "nccl" and device is not None and device.type == "cuda":
    dist.barrier(device_ids=[device.index if device.index is not None else torch.cuda.current_device()])
  else:
    dist.barrier()


def all_ranks(value: float, device: torch.device) -> List[float]:
  """One float per rank, in rank order, on every rank (per-rank step times in the bench line)."""
  if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
    return [float(value)]
  on_host = dist.get_backend() == "gloo"
  t = torch.tensor([value], dtype=torch.float64, device="cpu" if on_host else device)
  parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
  dist.all_gather(parts, t)
  return [float(p.item()) for p in parts]


def preflight(device: torch.device, n_expected: int, payload_bytes: int = 256 * 1024) -> dict:
  """Checks the multi-process set-up before any environment is built, so that the first RCCL run fails
  early and legibly instead of hanging in a collective: the world size is what the caller expects, every
  rank of a NODE sits on its own device (unless SBSIM_BENCH_SHARE_GPU=1; ranks on different nodes may well
  both use cuda:0 -- devices are compared per host), a 256 KiB all_gather returns every rank's
  pattern in rank order and an all_reduce(MAX) returns the largest rank.  Returns what it saw (rank 0 prints
  it in the bench line); raises SystemExit with the reason otherwise.  Backend-agnostic: "nccl" (= RCCL) on
  GPUs, "gloo" in the CPU tests."""
  if not (dist.is_available() and dist.is_initialized()):
    if n_expected != 1:
      raise SystemExit(f"preflight: {n_expected} ranks expected but torch.distributed is not initialised "
                       "(start under torch.distributed.run, or let bench.py --gpus N launch the ranks)")
    return {"ranks": 1, "backend": None, "devices": [str(device)], "ok": True}
  world, rank, backend = dist.get_world_size(), dist.get_rank(), dist.get_backend()
  if world != n_expected:
    raise SystemExit(f"preflight: {n_expected} ranks expected, the process group has {world}")
  on_host = backend == "gloo" or device.type != "cuda"
  cdev = torch.device("cpu") if on_host else device
  local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))   # ranks on THIS node (torchrun sets it)
  if device.type == "cuda":
    n_dev = torch.cuda.device_count()
    if not share_gpu() and n_dev < local_world:
      raise SystemExit(f"preflight: {local_world} ranks on this node but only {n_dev} visible GPU(s) "
                       "(one process per GPU; SBSIM_BENCH_SHARE_GPU=1 shares devices over gloo for tests)")
  # every rank's (host, device), in rank order
  dev_index = (device.index if device.index is not None else torch.cuda.current_device()) if device.type == "cuda" else -1
  host = zlib.crc32(socket.gethostname().encode()) & 0x7FFFFFFF
  mine = torch.tensor([host, dev_index], dtype=torch.int64, device=cdev)
  seen = [torch.empty_like(mine) for _ in range(world)]
  dist.all_gather(seen, mine)
  hosts, devices = [int(t[0].item()) for t in seen], [int(t[1].item()) for t in seen]
  if device.type == "cuda" and not share_gpu() and len(set(zip(hosts, devices))) != world:
    raise SystemExit(f"preflight: two ranks of one node share a device (rank -> device: {devices}); "
                     "each rank must call torch.cuda.set_device(LOCAL_RANK)")
  # a payload-sized all_gather: rank r sends r + 1 everywhere
  n = max(1, payload_bytes // 4)
  src = torch.full((n,), float(rank + 1), dtype=torch.float32, device=cdev)
  parts = [torch.empty_like(src) for _ in range(world)]
  dist.all_gather(parts, src)
  for r, p in enumerate(parts):
    if not bool((p == float(r + 1)).all()):
      raise SystemExit(f"preflight: all_gather returned wrong data for rank {r} on rank {rank} ({backend})")
  top = torch.tensor([float(rank)], dtype=torch.float64, device=cdev)
  dist.all_reduce(top, op=dist.ReduceOp.MAX)
  if int(top.item()) != world - 1:
    raise SystemExit(f"preflight: all_reduce(MAX) over ranks gave {top.item()}, expected {world - 1} ({backend})")
  if device.type == "cuda":
    torch.cuda.synchronize(device)
  names = devices if device.type != "cuda" else [f"cuda:{d}" for d in devices]
  return {"ranks": world, "nodes": len(set(hosts)), "backend": "rccl (torch 'nccl')" if backend == "nccl" else backend, "devices": names,
          "all_gather_bytes_per_rank": n * 4, "ok": True}


def max_over_ranks(seconds: float, device: torch.device) -> float:
  """Wall time of the slowest rank (bench.py contract)."""
  if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
    return seconds
  t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  return float(t.item())
