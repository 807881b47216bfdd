"""Worker of tests/test_rendezvous_cpu.py: one rank of a `python -m torch.distributed.run` launch that runs ONLY the
id hand-over of mcmc_ref_hip.shard (no GPU call, no library load) and writes what it saw to $RDZV_OUT/rank<r>.json."""
import hashlib
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path[:0] = [str(ROOT / "mcmc-db_amd")]

from mcmc_ref_hip import shard  # noqa: E402


def main() -> int:
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out = Path(os.environ["RDZV_OUT"])
    mode = os.environ.get("RDZV_MODE", "ok")
    base = shard._id_file(world)
    made = []

    def make_id() -> bytes:
        made.append(os.urandom(shard.COMM_ID_BYTES))
        return made[0]

    rec = {"rank": rank, "base": str(base), "key_parts": shard.comm_key(world)[1], "ppid": os.getppid()}
    if mode == "absent0" and rank == 0:                       # rank 0 never takes part: the others must time out
        rec["skipped"] = True
    else:
        try:
            uid = shard.exchange_unique_id(rank, world, make_id, timeout=float(os.environ.get("RDZV_TIMEOUT", "60")))
            rec["id_sha"] = hashlib.sha256(uid).hexdigest()
            rec["made"] = hashlib.sha256(made[0]).hexdigest() if made else None
        except TimeoutError as exc:
            rec["timeout"] = str(exc)
    (out / f"rank{rank}.json").write_text(json.dumps(rec))
    return 0


if __name__ == "__main__":
    sys.exit(main())
