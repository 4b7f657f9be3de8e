"""Data parallel over two processes on the GPU: the product train step (kernels, sharded batch, the
table / dense gradient exchange in two ranges, replicated clip + Adam) and the sharded eval must
reproduce the single-process run.  Transport is gloo with both ranks on cuda:0 (see dp_worker.py)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("exchange", ["allreduce", "sharded", "single"])
def test_two_rank_training_matches_single_process(tmp_path, exchange):
    """exchange = how the replicated table is kept in step: all-reduce + replicated Adam, or reduce-scatter +
    Adam on 1/world of the rows + all-gather of the updated rows (same arithmetic, 1/world of the Adam traffic)."""
    _run_two_ranks(tmp_path, "gloo", exchange)


@pytest.mark.parametrize("exchange", ["allreduce", "rows", "auto"])
def test_two_rank_two_pass_step_matches_single_process(tmp_path, monkeypatch, exchange):
    """The N-GPU step as the 1-GPU step (_train_step_dp): union of all ranks' ids marked, early table-Adam pass, then
    the table gradient exchanged densely or as touched rows, late pass over the union -- forced on for this small table
    through HPMN_TWO_PASS_MIN_NUMEL=0 (single-process reference included: it runs its own two-pass step)."""
    monkeypatch.setenv("HPMN_TWO_PASS_MIN_NUMEL", "0")
    _run_two_ranks(tmp_path, "gloo", exchange, extra_env={"HPMN_TWO_PASS_MIN_NUMEL": "0"})


@pytest.mark.parametrize("nproc", [2, 3])
def test_rows_step_prepared_a_step_ahead_matches_single_process(tmp_path, monkeypatch, nproc):
    """r5: train_step(next_ids=) -- the next step's scatter plan and the all-gather of the ranks' distinct-row lists and counts
    are issued a step ahead, on their own stream and a second communicator; the step itself then only marks, runs the early
    pass, and consumes the gathered rows with hpmn_rows_sum_adam.  Two and three ranks (ragged shards of 50: 16 / 17 / 17; the
    one-sample last batch leaves all but one rank with an empty NEXT shard) against the single process, which is told its next
    ids too."""
    env = {"HPMN_TWO_PASS_MIN_NUMEL": "0", "HPMN_DP_NEXT_IDS": "1"}
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    _run_two_ranks(tmp_path, "gloo", "rows", extra_env=env, nproc=nproc)


@pytest.mark.parametrize("exchange", ["rows", "auto"])
def test_four_rank_two_pass_step_matches_single_process(tmp_path, monkeypatch, exchange):
    """The same step among FOUR ranks (shards of 12 / 13 sequences; the one-sample last batch leaves three ranks with an empty
    shard, each of which still has to enter every collective): the chunked rows exchange with four contributors per range."""
    monkeypatch.setenv("HPMN_TWO_PASS_MIN_NUMEL", "0")
    _run_two_ranks(tmp_path, "gloo", exchange, extra_env={"HPMN_TWO_PASS_MIN_NUMEL": "0"}, nproc=4)


@pytest.mark.parametrize("nproc", [2, 4])
def test_dataset_sharded_evaluation_matches_single_process(tmp_path, nproc):
    """r6 (SURVEY 8e "eval shards the dataset and all-gathers predictions"; code/hpmn.py:351-373): under data
    parallel Hpmn.eval gives rank r the rows [n r / N, n (r+1) / N) of the WHOLE set, full-width passes on the same kernels a
    single process uses, ONE all-gather of predictions and ONE all-reduce of the memory-loss sum.  Same weights, no training:
    AUC to 1e-6, log-loss to 2e-6 relative, the memory-loss mean to 1e-5 relative (float32 sums in another order).  6 800 rows:
    every shard (3 400 / 1 700 rows) stays on tile kernels like the single process."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    out = str(tmp_path / "dp.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", HPMN_DP_BACKEND="gloo", HPMN_DP_EVAL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(HERE, "dp_worker.py"), out]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    got = np.load(out)["__eval__"]
    sys.path.insert(0, HERE)
    import dp_worker
    m, tr, te = dp_worker.build_eval(str(tmp_path / "single"))
    assert m.world == 1
    want = dp_worker.run_eval(m, tr, te)["__eval__"]
    assert 0.0 < want[0] < 1.0 and want[1] > 0 and want[2] > 0
    np.testing.assert_allclose(got[0], want[0], rtol=0, atol=1e-6)           # AUC
    np.testing.assert_allclose(got[1], want[1], rtol=2e-6)                   # log-loss (measured 2.4e-7: the shards' 3 400 / 1 700 rows
    #                                                                          take the twelve-wave tile kernel, 6 800 the four-wave one)
    np.testing.assert_allclose(got[2], want[2], rtol=1e-5)


def test_two_rank_lazy_table_adam_matches_single_process(tmp_path, monkeypatch):
    """Row-wise (lazy) Adam under data parallel -- what a table sized to HBM needs (BASELINE configs[4]): every rank's
    touched rows and compact gradient rows are all-gathered, the union updated identically everywhere."""
    monkeypatch.setenv("HPMN_LAZY_TABLE_ADAM", "1")
    _run_two_ranks(tmp_path, "gloo", "auto", extra_env={"HPMN_LAZY_TABLE_ADAM": "1"})


def test_two_rank_training_over_rccl_matches_single_process(tmp_path):
    """The same run with one GPU per rank and backend "nccl" (= RCCL over xGMI): the asynchronous chunked table
    all-reduce, wait() as a stream dependency (not a host block as with gloo), Adam of range i under the reduce
    of range i+1, against the caching allocator and the side stream.  Needs two devices; the 1-GPU boxes skip."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (RCCL refuses two ranks on one device)")
    _run_two_ranks(tmp_path, "nccl")
    _run_two_ranks(tmp_path, "nccl", "sharded")
    _run_two_ranks(tmp_path, "nccl", "rows", extra_env={"HPMN_TWO_PASS_MIN_NUMEL": "0"})


@pytest.mark.parametrize("exchange", ["rows", "rows_ahead", "rows_wire_standin", "allreduce", "sharded", "lazy"])
def test_every_collective_of_the_step_runs_on_rccl_with_one_rank(tmp_path, monkeypatch, exchange):
    """RCCL refuses two ranks on one device and the boxes have one GPU, so until r4 no RCCL call of the data-parallel
    step had ever executed.  HPMN_DP_FORCE_COLLECTIVES=1 removes the world-size-1 short cuts: a ONE-rank "nccl" process
    group runs the data-parallel train step and eval with every collective really issued on RCCL -- process-group
    initialisation with device_id, all_gather_into_tensor of int32 ids / int64 counts / fp32 rows, the asynchronous
    counts copy into pinned memory, all_reduce / reduce_scatter_tensor on the caller's streams -- and has to land on the
    single-process result."""
    env = {"HPMN_DP_FORCE_COLLECTIVES": "1", "HPMN_TWO_PASS_MIN_NUMEL": "0"}
    if exchange == "rows_ahead":
        # (r5: the next step's plan + id exchange a step ahead, on the plan stream and the SECOND RCCL communicator)
        env["HPMN_DP_NEXT_IDS"] = "1"
        monkeypatch.setenv("HPMN_DP_NEXT_IDS", "1")
        exchange = "rows"
    if exchange == "rows_wire_standin":
        # (r5, measurement switch: sleep kernels on a stream of their own stand in for the wire -- the late launches wait for
        #  them instead of for the collective; the result must not change)
        env["HPMN_DP_WIRE_STANDIN"] = "300,8"
        exchange = "rows"
    if exchange == "lazy":
        env["HPMN_LAZY_TABLE_ADAM"] = "1"
        monkeypatch.setenv("HPMN_LAZY_TABLE_ADAM", "1")          # (the single-process reference of the comparison too)
        exchange = "auto"
    _run_two_ranks(tmp_path, "nccl", exchange, extra_env=env, nproc=1)


def test_bench_gpus_n_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the way the driver's 1-GPU command line spells it)
    re-executes itself under torch.distributed.run and rank 0 prints ONE JSON line (VERDICT r3 missing #1: it used to exit).
    gloo transport, both ranks on this box's GPU."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--config", "c1", "--steps", "4",
           "--warmup", "2", "--no-cpu-baseline", "--no-roofline", "--no-auc", "--no-parity-gate", "--no-eval"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2 * d["config"]["per_gpu_batch"] and d["value"] > 0
    assert d["scaling"] == "weak" and d["steps"] == 4


def test_bench_gpus_n_with_the_rank_0_legs(tmp_path):
    """The driver's spelling has NO --no-* flags: after the timed steps rank 0 alone runs the parity gate and the roofline
    probes -- and the probe of the dominant kernel runs TRAINING STEPS (collectives).  The other ranks have to run them
    too (bench.in_step_probe_partner), or rank 0 waits in an all-gather while they wait in the final barrier.
    Taobao shape (H = 64: the in-step probe exists), two gloo ranks on this box's GPU."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--config", "c2", "--steps", "3",
           "--warmup", "2", "--no-cpu-baseline", "--no-auc"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["roofline"]["ms_per_launch"] > 0 and d["parity_gate"]["pass"]
    assert d["replicas_identical"] is True          # (r5: bit patterns of parameters and moments, compared across the ranks)


def test_two_ranks_on_a_table_beyond_int32_rows(tmp_path):
    """Row g under data parallel: two ranks, each with a 2.2 G-row table (35 GB per buffer, param + m + v under lazy table
    Adam), int64 ids, the touched-rows exchange with int64 row ids on the wire -- against the single-process run."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    free, _ = torch.cuda.mem_get_info()
    if free < 240e9:
        pytest.skip("needs ~230 GB of free HBM for the two ranks")
    out = str(tmp_path / "dp.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", HPMN_DP_BACKEND="gloo", HPMN_TABLE_EXCHANGE="auto", HPMN_DP_BIG="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(HERE, "dp_worker.py"), out]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    got = np.load(out)
    sys.path.insert(0, HERE)
    import dp_worker
    m, tr, te = dp_worker.build_big(str(tmp_path / "single"))
    want = dp_worker.run_big(m, tr, te)
    del m
    torch.cuda.empty_cache()
    assert float(got["below"][0]) == float(want["below"][0])          # rows nobody named never moved (same init on both)
    assert np.abs(want["m_top"]).max() > 0
    for k in want:
        if k in ("__eval__", "below"):
            continue
        tol = 0.003 * 4 * 1.05 if k.endswith(("dense_3/bias", "dense_6/bias", "dense_9/bias")) else 2e-5
        np.testing.assert_allclose(got[k], want[k], rtol=0, atol=tol, err_msg=k)
    np.testing.assert_allclose(got["__eval__"], want["__eval__"], rtol=1e-4, atol=1e-4)


def _run_two_ranks(tmp_path, backend, exchange="allreduce", extra_env=None, nproc=2):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    out = str(tmp_path / "dp.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", HPMN_DP_BACKEND=backend, HPMN_TABLE_EXCHANGE=exchange)
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(HERE, "dp_worker.py"), out]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    got = np.load(out)

    sys.path.insert(0, HERE)
    import dp_worker
    m, tr, te = dp_worker.build(str(tmp_path / "single"))
    assert m.world == 1
    want = dp_worker.run(m, tr, te)
    for k in want:
        if k == "__eval__":
            continue
        if k.endswith(("dense_3/bias", "dense_6/bias", "dense_9/bias")):
            # the bias in front of a softmax shifts every score alike: its exact gradient is 0, what arrives is
            # rounding noise, and Adam normalises noise to +-lr steps -- within the trust region is all one can ask
            np.testing.assert_allclose(got[k], want[k], rtol=0, atol=0.003 * 6 * 1.05, err_msg=k)
            continue
        # same kernels, but per-rank partial sums are added in a different order than one full batch
        np.testing.assert_allclose(got[k], want[k], rtol=0, atol=2e-5, err_msg=k)
    np.testing.assert_allclose(got["__eval__"], want["__eval__"], rtol=1e-4, atol=1e-4)
