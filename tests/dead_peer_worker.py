"""Worker of tests/test_rendezvous_cpu.py::test_a_rank_that_dies_after_the_rendezvous: one rank of a two-rank gloo world
(the CPU stand-in for the RCCL communicator, tests/gloo_comm.py).  Both ranks meet in a barrier (the rendezvous is over);
rank 1 then DIES (os._exit, no clean-up, as a crashed process would); rank 0 goes on into the failure agreement and the
record gather of mcmc_ref_hip.shard and must come out of them with an exception inside the collective's deadline -- and
leave with a non-zero exit code -- instead of waiting for ever.  argv: rank world port deadline_seconds."""
import datetime
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "mcmc-db_amd"), str(ROOT / "tests")]


def main() -> int:
    rank, world, port, deadline = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
    import torch.distributed as dist
    from gloo_comm import GlooComm
    from mcmc_ref_hip import shard
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=deadline))
    comm = GlooComm(dist)
    comm.barrier()                                    # every rank is here: the rendezvous is behind us
    if rank == 1:
        os._exit(17)                                  # dies without a word
    t0 = time.time()
    try:
        shard._agree(comm, None, "unit")              # the 1-double agreement in front of every data collective
        shard.gather_records(np.zeros((3, shard.RECORD_DOUBLES)), comm)
    except Exception as exc:  # noqa: BLE001 - whatever the transport raises is the point
        print(f"RAISED {type(exc).__name__} after {time.time() - t0:.1f} s", flush=True)
        return 3
    print("NO ERROR", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
