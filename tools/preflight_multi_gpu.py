"""Checks a one-node multi-GPU set-up before the first RCCL run.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        tools/preflight_multi_gpu.py [--cpu]

Every rank: one process per device (LOCAL_RANK), `sbsim_amd.distributed.preflight` -- world size, one rank per
device, a 256 KiB all_gather and an all_reduce(MAX) on the "nccl" (= RCCL) backend -- and a line with what it
saw.  Exit status 0 only if every rank passed.  `--cpu`: the same checks on gloo (what the CPU tests run).
`bench.py --gpus N` runs the same check before it builds its environments."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sbsim_amd import distributed as sd  # noqa: E402


def main() -> None:
  ap = argparse.ArgumentParser()
  ap.add_argument("--cpu", action="store_true", help="gloo on CPU tensors (no GPU needed)")
  args = ap.parse_args()
  rank, local_rank, world = sd.env_rank_world()
  if args.cpu:
    dev = torch.device("cpu")
    sd.init_process_group("gloo")
  else:
    if not torch.cuda.is_available():
      raise SystemExit("preflight: no GPU visible (use --cpu for the gloo check)")
    local_rank = sd.device_index(local_rank)
    if local_rank >= torch.cuda.device_count():
      raise SystemExit(f"preflight: LOCAL_RANK {local_rank} but {torch.cuda.device_count()} visible GPU(s)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sd.init_process_group(sd.backend_for_gpu())
  seen = sd.preflight(dev, world)
  sd.barrier(dev)
  name = torch.cuda.get_device_name(dev) if dev.type == "cuda" else "cpu"
  print(f"preflight rank {rank}/{world}: device {dev} ({name}), backend {seen['backend']}, "
        f"rank -> device {seen['devices']}, all_gather {seen.get('all_gather_bytes_per_rank', 0)} B per rank: ok", flush=True)
  if torch.distributed.is_initialized():
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
  main()
