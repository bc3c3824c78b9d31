"""Can RCCL carry two ranks on ONE device (the lease has one GPU)?  Runs the rendezvous, a barrier and the all-reduces bench.py uses,
both ranks on cuda:0.  Prints what happens; exit code 0 either way (it is a probe, not a test)."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if "RANK" not in os.environ:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
    codes = []
    for p in procs:
        try:
            codes.append(p.wait(timeout=120))
        except subprocess.TimeoutExpired:
            p.kill()
            codes.append("timeout")
    print("rank exit codes", codes)
    # one rank alone: the RCCL calls themselves (communicator, barrier, all-reduce) without a second device
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port + 1), PRS_ONE_RANK="1")
    print("single-rank exit code", subprocess.call([sys.executable, os.path.abspath(__file__)], env=env))
    sys.exit(0)

import torch  # noqa: E402
from srrg2_proslam_amd import sharding  # noqa: E402

torch.cuda.set_device(0)
if os.environ.get("PRS_ONE_RANK") == "1":
    import torch.distributed as dist
    dist.init_process_group(backend="nccl", world_size=1, rank=0, device_id=torch.device("cuda", 0))
    dist.barrier()
    t = torch.tensor([3.0], dtype=torch.float64, device="cuda:0")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    print("single rank: RCCL communicator, barrier and all_reduce OK:", float(t.item()), "world", dist.get_world_size())
    dist.destroy_process_group()
    sys.exit(0)
try:
    sharding.init_distributed("nccl", 0)
    sharding.barrier()
    dev = torch.device("cuda", 0)
    fps, slow = sharding.aggregate_throughput(100 * (1 + int(os.environ["RANK"])), 1.0 + int(os.environ["RANK"]), dev)
    per = sharding.gather_over_ranks(float(os.environ["RANK"]) + 1.0, dev)
    print("rank", os.environ["RANK"], "nccl on one device OK: world", sharding.backend_world_size(), "fps", fps, "slowest", slow, "gather", per)
    sharding.shutdown()
except Exception as exc:  # noqa: BLE001
    print("rank", os.environ["RANK"], "nccl on one device FAILED:", repr(exc)[:300])
