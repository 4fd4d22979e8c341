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
    # (the cull's 2n selection pass rides in the blend's sweep: its own launch, select_lvl2_cull, returns at once)
    assert t["f1_rows_fwd"] + t["f2_cols_fwd"] + t["select_lvl2"] + t["select_lvl2_cull"] + t["blend"] + t["i1_cols_inv"] + t["i2_rows_inv"] == 54
    # per layer: K-1 pair merges, an fp32 intermediate costs the row pass 2n more
    names = ["f1_rows_fwd", "f2_cols_fwd", "select_lvl2", "select_lvl2_cull", "blend", "i1_cols_inv", "i2_rows_inv"]
    assert sum(bench.kernel_alg_bytes_per_elem(n, 2) for n in names) == 54
    # K = 3: the intermediate stays spectral - 93n moved where the canonical model counts 122n
    names3 = names + ["f2s_cols_fwd1", "spec_norm", "spec_rescale"]
    assert sum(bench.kernel_alg_bytes_per_elem(n, 3) for n in names3) == 24 + 0 + 21 + 8 + 0 + 12 + 8 + 8 + 4 + 4    # 89n
    assert bench.kernel_alg_bytes_per_elem("publish", 3) == 0


def test_default_workload_is_the_metrics_configuration(bench, monkeypatch):
    import sys
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    import argparse
    seen = {}
    real = argparse.ArgumentParser.parse_args

    def spy(self, *a, **kw):
        ns = real(self, *a, **kw)
        seen["ns"] = ns
        raise SystemExit(0)
    monkeypatch.setattr(argparse.ArgumentParser, "parse_args", spy)
    with pytest.raises(SystemExit):
        bench.main()
    assert seen["ns"].workload == "llama3-70b-slice" and seen["ns"].k == 3 and seen["ns"].gpus == 1


def test_traffic_file_is_refused_for_another_workload(bench, tmp_path, monkeypatch):
    import json
    (tmp_path / "profiles").mkdir()
    sha = bench.csrc_stamp()
    json.dump({"_meta": {"workload": "llama3-8b", "k": 2, "csrc_sha": sha}, "f2_cols_fwd": {"hbm_bytes_per_launch": 5.0}},
              open(tmp_path / "profiles" / "traffic_latest.json", "w"))
    real_repo = bench.REPO
    monkeypatch.setattr(bench, "csrc_stamp", lambda: sha)
    monkeypatch.setattr(bench, "REPO", tmp_path)
    assert bench.load_traffic("llama3-8b", 2, "f2_cols_fwd") == 5.0
    assert bench.load_traffic("llama3-70b-slice", 3, "f2_cols_fwd") is None
    assert bench.load_traffic("llama3-8b", 3, "f2_cols_fwd") is None
    # ... and for other kernel sources than the ones it was measured on
    monkeypatch.setattr(bench, "csrc_stamp", lambda: "0" * 16)
    assert bench.load_traffic("llama3-8b", 2, "f2_cols_fwd") is None
    assert len(sha) == 16 and real_repo.exists()


def test_moved_bytes_model(bench):
    assert bench.moved_bytes_per_elem(2, "exact") == 56 and bench.moved_bytes_per_elem(3, "exact") == 85      # (89 less the base rows shared by the K row passes)
    assert bench.moved_bytes_per_elem(3, "reference_cpu") > bench.moved_bytes_per_elem(3, "exact")
    assert bench.moved_bytes_per_elem(3, "reference_cpu") < bench.alg_bytes_per_elem(3)


def test_gpus_n_outside_torchrun_starts_the_ranks_as_a_child(bench, monkeypatch):
    """`python bench.py --gpus 8` must not die with a usage message (the driver's SCALE run):
    it spawns torch.distributed.run as a CHILD before touching the GPU."""
    import subprocess
    import sys
    calls = []
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd, env = calls[0]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_lpt_partition_is_balanced_and_deterministic():
    costs = [alg_bytes(r * c, 3) for r, c in [(8192, 8192), (1024, 8192), (28672, 8192), (28672, 8192), (8192, 28672)] * 16]
    a = partition_lpt(costs, 8)
    assert a == partition_lpt(costs, 8)
    load = [sum(c for c, o in zip(costs, a) if o == r) for r in range(8)]
    assert max(load) <= 1.05 * (sum(costs) / 8)
