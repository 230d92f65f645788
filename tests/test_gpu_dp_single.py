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
