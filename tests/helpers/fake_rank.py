"""Stand-in for bench.py's ranks in tests/test_bench_launch.py: rank 0 prints one JSON line, FAKE_RANK_FAIL picks a rank that fails."""
import json
import os
import sys

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
if os.environ.get("FAKE_RANK_FAIL") == str(rank):
    sys.exit(7)
if rank == 0:
    print(json.dumps({"n_gpus": world, "argv": sys.argv[1:], "master": os.environ["MASTER_ADDR"]}), flush=True)
