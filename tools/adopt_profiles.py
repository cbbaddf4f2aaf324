"""Developer tool: copies what tools/collect_profiles.sh wrote under gpurun_out/profiles_<tag>/ into profiles/
with the round's prefix and stamps the traffic profile with the commit it was measured on.

    python tools/adopt_profiles.py <tag>          (e.g. r05; run in the build container after the gpurun call)

bench.py replays profiles/traffic_latest.json into `roofline.traffic` (it cannot collect PMC counters inside its
own run) and says so: `traffic_measured_in_run: false`, `traffic_profile_commit` -- and only when the profile's
`kernel_source_sha256` (the files the bench kernel is compiled from) is this tree's."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main() -> None:
  tag = sys.argv[1]
  src = os.path.join(ROOT, "gpurun_out", f"profiles_{tag}")
  dst = os.path.join(ROOT, "profiles")
  commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
  dirty = bool(subprocess.run(["git", "status", "--porcelain", "sbsim_amd", "bench.py"], cwd=ROOT, capture_output=True, text=True).stdout.strip())
  for name in sorted(os.listdir(src)):
    shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{name}"))
  tr = os.path.join(src, "traffic.json")
  if os.path.exists(tr):
    t = json.load(open(tr))
    t["commit"] = commit + ("+uncommitted" if dirty else "")
    sys.path.insert(0, ROOT)
    import bench   # the hash bench.py compares: the files k_sweep_roll is compiled from
    t["kernel_source_sha256"] = bench.kernel_source_sha256()
    t["kernel_source_files"] = list(bench.KERNEL_SOURCES)
    json.dump(t, open(os.path.join(dst, f"{tag}_traffic.json"), "w"), indent=1)
    json.dump(t, open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
  print(f"profiles/{tag}_* <- {src} (commit {commit}{'+uncommitted' if dirty else ''})")


main()
