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


def test_class_shards_give_every_rank_the_same_mix():
  totals = [21845, 21845, 21846]
  for world in (1, 2, 8):
    spans = [sd.class_shard_ranges(totals, r, world) for r in range(world)]
    for k, n in enumerate(totals):
      assert spans[0][k][0] == 0 and spans[-1][k][1] == n
      sizes = [sp[k][1] - sp[k][0] for sp in spans]
      assert max(sizes) - min(sizes) <= 1 and sum(sizes) == n
      for a, b in zip(spans, spans[1:]):
        assert a[k][1] == b[k][0]


def _worker8(rank: int, world: int, port: int, n_total: int, totals):
  os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                    MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  assert sd.init_process_group("gloo")
  pf = sd.preflight(torch.device("cpu"), world)
  assert pf["ranks"] == world and pf["ok"] and pf["nodes"] == 1
  lo, hi = sd.shard_range(n_total, rank, world)
  full = sd.gather_returns(torch.arange(lo, hi, dtype=torch.float32), n_total)
  assert torch.equal(full, torch.arange(n_total, dtype=torch.float32)), (rank, full)
  # three classes, each sharded on its own; the rank's vector is in class order, the gathered one class-major
  spans = sd.class_shard_ranges(totals, rank, world)
  local = torch.cat([torch.arange(a, b, dtype=torch.float32) + 1000.0 * k for k, (a, b) in enumerate(spans)])
  got = sd.gather_returns_by_class(local, totals)
  want = torch.cat([torch.arange(0, n, dtype=torch.float32) + 1000.0 * k for k, n in enumerate(totals)])
  assert torch.equal(got, want), (rank, got, want)
  assert sd.all_ranks(float(rank), torch.device("cpu")) == [float(r) for r in range(world)]
  assert sd.max_over_ranks(1.0 + rank, torch.device("cpu")) == float(world)
  sd.barrier(torch.device("cpu"))
  dist.destroy_process_group()


def test_world_size_8_gloo_shards_gathers_and_preflight():
  """Eight ranks (the node the scaling bench runs on): preflight, shard ranges, the gather in global order for
  one class (ragged: 8 does not divide 75) and for three classes sharded per class."""
  port = _free_port()
  mp.spawn(_worker8, args=(8, port, 75, [19, 8, 30]), nprocs=8, join=True)


def test_gather_by_class_single_process():
  for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
    os.environ.pop(k, None)
  x = torch.arange(7, dtype=torch.float32)
  assert torch.equal(sd.gather_returns_by_class(x, [3, 4]), x)
