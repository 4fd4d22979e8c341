"""bench.py's CPU-side pieces: workload tables, the algorithmic-byte model and
the partition used by the multi-rank merge agree with each other."""
import importlib.util
from pathlib import Path

import pytest

from shardmerge_amd.distributed import alg_bytes, partition_lpt

REPO = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_module", REPO / "bench.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_workloads_match_the_survey_shapes(bench):
    shapes, desc = bench.workload_shapes("llama3-8b", 0)
    assert len(shapes) == 288 and sum(r * c for r, c in shapes) == 6_979_584_000      # 6.98 G FFT'd params
    shapes70, _ = bench.workload_shapes("llama3-70b-slice", 0)
    assert sum(r * c for r, c in shapes70) * 8 == 80 * sum(r * c for r, c in bench.LLAMA3_70B_BLOCK)
    assert bench.workload_shapes("8192sq", 0)[0] == [(8192, 8192)] * 4
    with pytest.raises(SystemExit):
        bench.workload_shapes("nope", 0)


def test_algorithmic_bytes_model(bench):
    assert [bench.alg_bytes_per_elem(k) for k in (2, 3, 4)] == [60, 122, 182]
    for k in (2, 3, 4):
        assert alg_bytes(1000, k) == 1000 * bench.alg_bytes_per_elem(k)
    # the per-kernel table adds up to the pipeline total of one raw pair minus the reduction pass that is fused away
    t = bench.KERNEL_ALG_BYTES
    assert t["f1_rows_fwd"] + t["f2_cols_fwd"] + 2 * t["select_lvl2"] + t["blend"] + t["i1_cols_inv"] + t["i2_rows_inv"] == 56


def test_lpt_partition_is_balanced_and_deterministic():
    costs = [alg_bytes(r * c, 3) for r, c in [(8192, 8192), (1024, 8192), (28672, 8192), (28672, 8192), (8192, 28672)] * 16]
    a = partition_lpt(costs, 8)
    assert a == partition_lpt(costs, 8)
    load = [sum(c for c, o in zip(costs, a) if o == r) for r in range(8)]
    assert max(load) <= 1.05 * (sum(costs) / 8)
