"""Pin the oracle against EVERY packaged golden the reference holds: the
meta.json["diagnostics"] of all models whose draws are present under /root/reference.  Runs only
where the reference checkout exists (the build container); reads data files only (Parquet + JSON),
imports no reference code.  The GPU box never sees this path (it has the six committed fixtures)."""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import pytest

from conftest import rel_close

DATA = Path("/root/reference/packages/mcmc-ref-data/src/mcmc_ref_data/data")

pytestmark = pytest.mark.skipif(not (DATA / "draws").exists(), reason="reference data not present")


def test_all_packaged_goldens(oracle):
    import pyarrow.parquet as pq
    n_values = n_bit = n_models = 0
    worst = 0.0
    for path in sorted((DATA / "draws").glob("*.draws.parquet")):
        name = path.name[: -len(".draws.parquet")]
        meta = json.loads((DATA / "meta" / f"{name}.meta.json").read_text())
        table = pq.read_table(path)
        params = [c for c in table.column_names if c not in {"chain", "draw"}]
        C, N = meta["n_chains"], meta["n_draws_per_chain"]
        chain = table.column("chain").to_numpy()
        assert np.array_equal(chain, np.repeat(np.arange(C), N)), name      # files are chain-major ordered
        x = np.stack([table.column(p).to_numpy().reshape(C, N) for p in params])
        s = oracle.summarize(x, "pcn")
        for i, p in enumerate(params):
            for k in ("rhat", "ess_bulk", "ess_tail"):
                got, exp = float(s[k][i]), meta["diagnostics"][p][k]
                assert rel_close(got, exp, 1e-12), (name, p, k, got, exp)
                n_values += 1
                n_bit += got == exp
                if exp == exp and exp not in (float("inf"),):
                    worst = max(worst, abs(got - exp) / abs(exp))
        chk = {"ndraws_is_10k": C * N == 10000, "nchains_is_gte_4": C >= 4,
               "ess_above_400": all(v > 400 for v in s["ess_bulk"]), "rhat_below_1_01": all(v < 1.01 for v in s["rhat"])}
        assert chk == meta["checks"], name
        n_models += 1
    assert n_models >= 57 and n_values >= 1380
    assert n_bit >= 1100 and worst < 1e-13            # SURVEY: 1143 / 1380 bit-equal, max deviation 6.4e-15
