"""Pin the oracle against EVERY packaged golden the reference holds: the
meta.json["diagnostics"] of all 57 models whose draws the reference packages.  Reads the committed copies of
those data files (tests/golden/corpus, Parquet + JSON; tests/golden/make_corpus_fixture.py), imports no reference code."""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import pytest

from conftest import GOLDEN, rel_close

DATA = GOLDEN / "corpus"


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


def test_committed_corpus_files_match_the_reference_manifest():
    """The reference pins its packaged files by sha256 (provenance_manifest.json, checked by its own
    tests/integration/test_package_data_completeness.py:82-97): the committed copies are those files."""
    import hashlib
    manifest = json.loads((DATA / "provenance_manifest.json").read_text())["files"]
    n = 0
    for sub, pat in (("draws", "*.draws.parquet"), ("meta", "*.meta.json")):
        for f in sorted((DATA / sub).glob(pat)):
            assert manifest[f"{sub}/{f.name}"] == hashlib.sha256(f.read_bytes()).hexdigest(), f.name
            n += 1
    assert n == 57 + 63
