"""Multi-GPU: independent building shards, one process per GPU, one gather at rollout end.

Buildings never exchange data, so there is no per-step collective.
Each rank owns a contiguous block of the global batch; after a rollout the per-building
returns are all-gathered (RCCL over xGMI when the backend is "nccl"; "gloo" in the CPU
tests).  Nothing here touches HIP, so it is covered by world_size-2 gloo tests."""
from __future__ import annotations

import os
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


def max_over_ranks(seconds: float, device: torch.device) -> float:
  """Wall time of the slowest rank (bench.py contract)."""
  if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
    return seconds
  t = torch.tensor([seconds], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  return float(t.item())
