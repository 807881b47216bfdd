"""All packaged models (real draws files of the reference corpus, parquet-cpp-arrow 23.0.0) through the native Parquet
ingest and the kernels, against the diagnostics the reference itself packaged in meta.json (1 380
checkable goldens).  The data files (57 draws files, 63 meta files: DATA, not source) are committed under
tests/golden/corpus/{draws,meta} (tests/golden/make_corpus_fixture.py copies them), so BASELINE config 3
"full packaged mcmc-ref-data reference set, 1 MI355X" runs in every `-m gpu` pass."""
from __future__ import annotations

import json

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
ROOT = GOLDEN / "corpus"


def test_every_packaged_golden_through_native_ingest_and_kernels():
    import pyarrow.parquet as pq
    from mcmc_ref_hip import _ffi, parquet
    from mcmc_ref_hip.convert import _checks, table_to_tensor
    paths = sorted((ROOT / "draws").glob("*.draws.parquet"))
    assert len(paths) == 57
    ctx = _ffi.Context(0)
    res = parquet.summarize_files(ctx, paths, min_chains=4)
    n_vals, worst, n_exact = 0, 0.0, 0
    for path, got in zip(paths, res):
        name = path.name[: -len(".draws.parquet")]
        meta = json.loads((ROOT / "meta" / f"{name}.meta.json").read_text())
        diag = meta["diagnostics"]
        assert list(got) == meta["parameters"] == list(diag), name
        for p, gold in diag.items():
            for k in ("rhat", "ess_bulk", "ess_tail"):
                a, b = got[p][k], gold[k]
                rel = abs(a - b) / max(abs(b), 1e-300)
                worst = max(worst, rel)
                n_exact += a == b
                n_vals += 1
                assert rel <= 1e-6, (name, p, k, a, b)
        slim = {p: {k: got[p][k] for k in ("rhat", "ess_bulk", "ess_tail")} for p in got}
        assert _checks(meta["n_chains"], meta["n_draws_per_chain"], slim) == meta["checks"], name
        # the native ingest against pyarrow on the real file, bit for bit
        t = pq.read_table(path)
        x, counts = table_to_tensor(t, meta["parameters"])
        d = parquet.read_draws(ctx, path)
        assert np.array_equal(d.to_host().view(np.int64), x.view(np.int64)) and list(d.counts) == list(counts), name
        assert len(counts) == meta["n_chains"] and int(counts[0]) == meta["n_draws_per_chain"]
        d.free()
    print(f"\n{len(paths)} models, {n_vals} goldens, worst rel {worst:.3g}, {n_exact} bit-equal")
    assert n_vals >= 1300 and worst < 1e-9
    ctx.close()
