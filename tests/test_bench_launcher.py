"""bench.py --gpus N launches N ranks by itself (re-exec under torch.distributed.run) and the ranks
agree on the world size; run here with a stub step on the gloo backend (no GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
  env = dict(os.environ)
  for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
    env.pop(k, None)
  env.update(env_extra or {})
  return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                        timeout=timeout, env=env, cwd=ROOT)


def _json_line(out):
  lines = [l for l in out.splitlines() if l.startswith("{")]
  assert len(lines) == 1, out
  return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks_and_gathers_in_global_order():
  r = _run(["--gpus", "2", "--stub-step", "--steps", "3", "--warmup", "1", "--buildings", "37"])
  assert r.returncode == 0, r.stderr[-2000:]
  d = _json_line(r.stdout)
  assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
  assert d["gathered_returns"] == 74 and d["gather_in_global_order"] is True
  assert d["return_gather_ms"] >= 0.0 and d["scaling"] == "weak"
  # the preflight check ran before the steps (world size, a 256 KiB all_gather, all_reduce(MAX)) and every
  # rank's own step time is in the line, so that a scaling line can be verified
  assert d["rccl_ranks"] == 2 and d["preflight"]["ok"] is True and d["preflight"]["backend"] == "gloo"
  assert d["preflight"]["all_gather_bytes_per_rank"] == 256 * 1024
  assert len(d["per_rank_ms_per_step"]) == 2 and max(d["per_rank_ms_per_step"]) <= d["ms_per_step"] * 1.001 + 1e-9


def test_gpus_1_runs_in_process():
  r = _run(["--gpus", "1", "--stub-step", "--steps", "2", "--warmup", "0", "--buildings", "5"])
  assert r.returncode == 0, r.stderr[-2000:]
  d = _json_line(r.stdout)
  assert d["n_gpus"] == 1 and d["gathered_returns"] == 5
  assert d["rccl_ranks"] == 1 and d["per_rank_ms_per_step"] == [d["ms_per_step"]]


def test_world_size_mismatch_is_an_error():
  r = _run(["--gpus", "4", "--stub-step"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
  assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_preflight_tool_on_gloo():
  """tools/preflight_multi_gpu.py --cpu under torch.distributed.run: both ranks pass and say so."""
  import socket
  with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
  env = dict(os.environ)
  for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
    env.pop(k, None)
  r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                      "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tools", "preflight_multi_gpu.py"), "--cpu"],
                     capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
  assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
  assert r.stdout.count(": ok") == 2 and "preflight rank 0/2" in r.stdout and "preflight rank 1/2" in r.stdout


def test_preflight_refuses_a_world_size_it_was_not_told():
  from sbsim_amd import distributed as sd
  import torch
  assert sd.preflight(torch.device("cpu"), 1)["ranks"] == 1
  with pytest.raises(SystemExit, match="2 ranks expected"):
    sd.preflight(torch.device("cpu"), 2)


def test_gpus_8_rehearsal_on_gloo():
  """The driver's 8-GPU command shape on CPU: eight ranks, the preflight, barrier-bracketed timing, the gather
  of 8 x B returns in global order (the first time more than two ranks meet in this code)."""
  r = _run(["--gpus", "8", "--stub-step", "--steps", "2", "--warmup", "1", "--buildings", "11"], timeout=600)
  assert r.returncode == 0, r.stderr[-2000:]
  d = _json_line(r.stdout)
  assert d["n_gpus"] == 8 and d["gathered_returns"] == 88 and d["gather_in_global_order"] is True
  assert d["rccl_ranks"] == 8 and d["preflight"]["ok"] is True and d["preflight"]["nodes"] == 1
  assert len(d["per_rank_ms_per_step"]) == 8 and len(d["preflight"]["devices"]) == 8


def test_mixed_config_two_ranks_every_class_on_every_rank():
  """--config mixed --gpus 2: each class is block-partitioned over the ranks on its own and the
  end-of-rollout gather returns the buildings in global class-major order."""
  r = _run(["--config", "mixed", "--gpus", "2", "--stub-step", "--steps", "2", "--warmup", "0", "--buildings", "30"])
  assert r.returncode == 0, r.stderr[-2000:]
  d = _json_line(r.stdout)
  assert d["n_gpus"] == 2 and d["gathered_returns"] == 60 and d["gather_in_global_order"] is True
  assert "three classes" in d["config"]["workload"]
