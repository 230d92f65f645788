"""GPU: the data-parallel launch path of the graph-replayed trainer on ONE device — an RCCL process group
of size 1 with the world size spoofed to 2, so the data-parallel branch of SPFNTrainer (RCCL all-reduce with
in-collective averaging + the flat optimizer captured INSIDE the step's graph) runs for real (a true multi-GPU run needs a multi-GPU node; the driver does that
with bench.py).  Runs in a subprocess so the process group does not leak into other tests."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(600)
@pytest.mark.parametrize("in_graph", ["1", "0"])
def test_graph_trainer_with_rccl_group(in_graph):
    """in_graph = 1: all-reduce + Adam are nodes of the step's graph (default); 0: the layout the trainer falls back to
    when the collective cannot be captured — graph up to the gradient packing, exchange + optimizer as eager launches."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", CPFN_EXCHANGE_IN_GRAPH=in_graph)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_graph_smoke.py")], capture_output=True,
                       text=True, env=env, timeout=540)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("graph captured")][-1]
    assert "graph captured: True" in line and "world in graph: 2" in line and "skipped 0.0" in line, line
    assert ("exchange in graph: %s" % (in_graph == "1")) in line, line + r.stderr[-1500:]
    first, last = [float(x) for x in line.split("loss")[1].split("skipped")[0].replace("->", " ").split()]
    assert last < 0.75 * first, line


@pytest.mark.timeout(600)
@pytest.mark.parametrize("in_graph", ["1", "0"])
@pytest.mark.parametrize("mode", ["own", "peer"])
def test_fault_word_on_the_replayed_path(mode, in_graph):
    """the fault word on the product's REPLAYED step (tests/test_ddp_gloo.py holds the eager CPU path).  own: the rank
    that raised runs the step that carries the word, skips, then errors.  peer: a rank that raised nothing finds the REDUCED slot
    non-zero in the pinned host word the step itself wrote behind the exchange and errors at the head of its next step — it never
    replays into a collective whose partner has stopped.  Weights and Adam moments bit-identical before / after in both."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", CPFN_EXCHANGE_IN_GRAPH=in_graph,
               CPFN_SMOKE_FAULT=mode)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_graph_smoke.py")], capture_output=True,
                       text=True, env=env, timeout=540)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("fault mode")][-1]
    assert "raised: True" in line and "weights untouched: True" in line and "moments untouched: True" in line, line
    assert "skipped + 1.0" in line, line


@pytest.mark.timeout(300)
def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` launches its own ranks; asking for more GPUs than the box has must fail loudly instead
    of reporting a 1-GPU number under another name."""
    import torch
    n = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n + 1), "--steps", "2", "--warmup", "3"],
                       capture_output=True, text=True, timeout=240)
    assert r.returncode != 0 and "GPUs requested" in r.stderr and not r.stdout.strip(), (r.returncode, r.stdout, r.stderr)


@pytest.mark.timeout(600)
def test_two_real_ranks_on_one_gpu_gloo():
    """World size 2 for real on a one-GPU box: two processes, both on GPU 0, gloo collective on device tensors (RCCL
    refuses two ranks on one device) — the bf16 graph trainer with the exchange + optimizer after the graph (gloo
    cannot be captured).  Replicas bit-identical after 12 steps from different initial weights, training progresses."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_two_ranks_one_gpu.py")], capture_output=True,
                       text=True, env=env, timeout=540)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("replicas identical")][-1]
    assert "replicas identical: True" in line and "graph: True" in line and "skipped: 0" in line, line
    v = [float(x) for x in line.replace("->", " ").split() if x.replace(".", "").isdigit() and "." in x]
    assert v[1] < 0.85 * v[0] and v[3] < 0.85 * v[2], line


@pytest.mark.timeout(900)
@pytest.mark.parametrize("collective", ["all_reduce", "rs_ag"])
def test_bench_two_ranks_share_one_gpu(collective):
    """bench.py's own N-rank plumbing executed for real on the one GPU there is — `--gpus 2` self-launches two
    ranks (children started before the parent touches the GPU), both on GPU 0 over gloo: the rank-time all_gather, comm_us_per_step,
    the fault slot on the collective and the collective layout string all run.  A PLUMBING check: the line says so and is never a
    scaling number (two processes time-slice one GPU)."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CPFN_BENCH_BACKEND="gloo", CPFN_BENCH_SHARE_GPU="1",
               CPFN_DP_COLLECTIVE=collective)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "4",
                        "--probe-replays", "2"], capture_output=True, text=True, env=env, timeout=840)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]                      # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 32 and d["config"]["parallelism"] == "dp2"
    assert "plumbing_check" in d and "NOT a scaling number" in d["plumbing_check"]
    assert len(d["ms_per_step_ranks"]["all"]) == 2 and all(v > 0 for v in d["ms_per_step_ranks"]["all"])
    assert collective in d["config"]["collective"] and "eager launches after the graph" in d["config"]["collective"]
    c = d["comm_us_per_step"]
    assert len(c["all"]) == 2 and all(v is not None and 0 < v < 1e6 for v in c["all"]) and c["median_over_ranks"] > 0
    assert d["value"] > 0 and d["config"]["launch"].startswith("hipGraph replay")
