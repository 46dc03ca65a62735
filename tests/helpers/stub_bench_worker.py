"""Stand-in for bench.py's GPU workers (tests/test_bench_launch.py): behaves as LZS_STUB_MODE says,
so the launcher / supervisor / fallback machinery of bench.py can be exercised without a GPU.
Modes: ok | fail_rank1 | hang_rank0 | fail_all."""
import json
import os
import sys
import time

role, rank, world = os.environ["LZS_BENCH_ROLE"], int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
mode = os.environ.get("LZS_STUB_MODE", "ok")
shared = os.environ["LZS_BENCH_DIR"]
assert os.path.isdir(shared)
if os.environ.get("LZS_STUB_PIDFILE"):
    for name in (os.environ["LZS_STUB_PIDFILE"], os.environ["LZS_STUB_PIDFILE"] + f".rank{rank}"):   # (the plain name: one-rank tests)
        with open(f"{name}.tmp{os.getpid()}", "w") as f:
            f.write(str(os.getpid()))
        os.replace(f"{name}.tmp{os.getpid()}", name)
if role == "job":
    if mode == "fail_rank1" and rank == 1:
        sys.exit(3)
    if mode == "fail_all":
        sys.exit(4)
    if mode == "hang_rank0" and rank == 0:
        time.sleep(600)
    if mode == "fail_rank1":
        time.sleep(600)                      # the other ranks would hang in a collective: the supervisor must end them
    if rank == 0:
        print("some banner on stdout")
        print(json.dumps({"metric": "m", "value": 1.0, "n_gpus": world, "from": "job", "argv": sys.argv[1:]}))
    sys.exit(0)
if role == "independent":
    if mode == "fail_all":
        sys.exit(5)
    open(os.path.join(shared, f"stub_ready.{rank}"), "w").close()
    t_end = time.time() + 60
    while sum(n.startswith("stub_ready.") for n in os.listdir(shared)) < world:
        assert time.time() < t_end
        time.sleep(0.01)
    if rank == 0:
        print(json.dumps({"metric": "m", "value": 0.5, "n_gpus": world, "from": "independent",
                          "fallback": {"reason": open(os.path.join(shared, "reason")).read()}}))
    sys.exit(0)
sys.exit(9)
