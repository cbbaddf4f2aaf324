"""world_size-2 gloo tests of the multi-GPU path (no GPU needed: shards never exchange data,
the only collective is the end-of-rollout return gather)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sbsim_amd import distributed as sd


def test_shard_ranges_cover_the_batch():
  for n, w in [(65536, 1), (65536, 8), (524288, 8), (10, 3), (7, 8), (1, 2)]:
    spans = [sd.shard_range(n, r, w) for r in range(w)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    for (a, b), (c, d) in zip(spans, spans[1:]):
      assert b == c and b >= a
    sizes = [b - a for a, b in spans]
    assert max(sizes) - min(sizes) <= 1
  with pytest.raises(ValueError):
    sd.shard_range(10, 3, 3)
  assert sd.shard_seed(1234, 5) == 1239


def _free_port() -> int:
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _worker(rank: int, world: int, port: int, n_total: int):
  os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                    MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  assert sd.init_process_group("gloo")
  lo, hi = sd.shard_range(n_total, rank, world)
  # a "rollout": every building's return is a function of its GLOBAL index only
  g = torch.Generator().manual_seed(sd.shard_seed(1234, rank))
  _ = torch.rand((4, hi - lo, 2), generator=g)          # per-rank action stream (shape only)
  local = torch.arange(lo, hi, dtype=torch.float32) * 0.5 - 3.0
  full = sd.gather_returns(local, n_total)
  expect = torch.arange(0, n_total, dtype=torch.float32) * 0.5 - 3.0
  assert torch.equal(full, expect), (rank, full, expect)
  slowest = sd.max_over_ranks(1.0 + rank, torch.device("cpu"))
  assert slowest == float(world)
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [64, 37])
def test_return_gather_world_size_2_gloo(n_total):
  port = _free_port()
  mp.spawn(_worker, args=(2, port, n_total), nprocs=2, join=True)


def test_single_process_paths():
  for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
    os.environ.pop(k, None)
  assert sd.env_rank_world() == (0, 0, 1)
  assert not sd.init_process_group("gloo")
  x = torch.arange(5, dtype=torch.float32)
  assert torch.equal(sd.gather_returns(x, 5), x)
  assert sd.max_over_ranks(2.5, torch.device("cpu")) == 2.5
