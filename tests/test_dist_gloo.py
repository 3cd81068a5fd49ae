"""world_size-2 tests of the pixel-row-sharded path.

CPU (gloo, runs everywhere): partition + callback plumbing + the two all-reduces, against the
single-rank oracle. GPU (-m gpu): two ranks sharing cuda:0 run the real HIP path and must
reproduce the single-rank output."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import glf
import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(mode, out_pattern, world=2, timeout=300):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, WORKER, mode, out_pattern], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o.decode())
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    return outs


def test_shard_rows_partition():
    for h, g in ((4096, 8), (37, 2), (5, 8), (1, 1), (300, 7)):
        spans = [glf.shard_rows(h, r, g) for r in range(g)]
        assert spans[0][0] == 0 and spans[-1][1] == h
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))     # contiguous, no overlap
        assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
    with pytest.raises(glf.GlfError):
        glf.shard_rows(10, 2, 2)


def test_two_rank_gloo_cpu(tmp_path):
    pattern = str(tmp_path / "cpu_rank%d.npz")
    _launch("cpu", pattern)
    r = [np.load(pattern % k) for k in range(2)]
    img = glf.synth_image(61, 47, seed=9)
    h, w = img.shape
    idx = glf.Sampling(w, h, 30)
    # both ranks hold the identical all-reduced vectors
    np.testing.assert_array_equal(r[0]["D"], r[1]["D"])
    np.testing.assert_array_equal(r[0]["c64"], r[1]["c64"])
    np.testing.assert_array_equal(r[0]["c32"], r[1]["c32"])
    # ... equal to the single-rank stages
    np.testing.assert_allclose(r[0]["D"], orc.degree(img, idx), rtol=1e-13)
    zf_ref, _, info = orc.image_processing(img, 30, 6, epsilon=0.1)
    assert int(r[0]["outer"]) == info["outer_its"]
    rows = [tuple(x["rows"]) for x in r]
    assert rows == [glf.shard_rows(h, 0, 2), glf.shard_rows(h, 1, 2)]
    z = np.concatenate([r[0]["z"], r[1]["z"]]).reshape(h, w)
    np.testing.assert_allclose(z, zf_ref, rtol=0, atol=1e-8)
    np.testing.assert_allclose(r[0]["c32"], r[0]["c64"], rtol=1e-5)
    for k in range(2):   # all-gathered mat-vec complete and identical on both ranks
        np.testing.assert_array_equal(r[k]["Y"], r[k]["Yref"])


@pytest.mark.gpu
@pytest.mark.parametrize("mode,paths", [("gpu", "direct"), ("gpu_replicated", "direct"), ("gpu", "grid")])
def test_two_ranks_share_one_gpu(tmp_path, mode, paths, monkeypatch):
    """mode gpu: L_A column-sharded, mat-vec rows all-gathered; gpu_replicated: no allgather
    callback, every rank solves the whole eigenproblem. Both must reproduce the single-rank run.
    paths: the entry-by-entry kernels or the grid-factored forms (which shard by image rows too)."""
    import torch
    monkeypatch.setenv("GLF_NYS_PATH", paths)   # inherited by the rank processes
    monkeypatch.setenv("GLF_DEG_PATH", paths)
    monkeypatch.setenv("GLF_MV_PATH", "grid" if paths == "grid" else "dense")
    assert torch.cuda.is_available()
    pattern = str(tmp_path / "gpu_rank%d.npz")
    _launch(mode, pattern, timeout=600)
    r = [np.load(pattern % k) for k in range(2)]
    img = glf.synth_image(96, 80, seed=4)
    h, w = img.shape
    with glf.Context(0) as ctx:
        opt = glf.default_options(num_samples=60, num_eigvals=8, epsilon=0.05)
        out1, zf1, info1 = ctx.image_processing(ctx.to_device(img), opt, want_float=True)
    out1, zf1 = out1.cpu().numpy(), zf1.cpu().numpy()
    assert [tuple(x["rows"]) for x in r] == [glf.shard_rows(h, 0, 2), glf.shard_rows(h, 1, 2)]
    assert int(r[0]["outer"]) == int(r[1]["outer"]) == info1["outer_its"]
    np.testing.assert_allclose(r[0]["eigvals"], info1["eigvals"], rtol=1e-5)
    np.testing.assert_array_equal(r[0]["eigvals"], r[1]["eigvals"])            # replicated eigen-solve
    (a0, a1), (b0, b1) = r[0]["rows"], r[1]["rows"]
    zf = np.vstack([r[0]["zf"][a0:a1], r[1]["zf"][b0:b1]])
    out = np.vstack([r[0]["out"][a0:a1], r[1]["out"][b0:b1]])
    np.testing.assert_allclose(zf, zf1, rtol=0, atol=5e-4)
    assert np.mean(out != out1) < 1e-3
    # rows a rank does not own stay untouched (zero-initialised by the binding)
    assert not r[0]["out"][a1:].any() and not r[1]["out"][:b0].any()


@pytest.mark.gpu
def test_one_rank_rccl_device_collectives(tmp_path):
    """The callbacks bench.py plugs in at N > 1 (torch.distributed backend "nccl" = RCCL, in place on the library's device
    buffers, on the context's stream) on a one-rank group with the communicator forced on: sharded eigen-solve with the
    all-gather callback, both all-reduces. Must reproduce the run without a communicator."""
    pattern = str(tmp_path / "nccl_rank%d.npz")
    _launch("gpu_nccl1", pattern, world=1, timeout=600)
    r = np.load(pattern % 0)
    img = glf.synth_image(96, 80, seed=4)
    with glf.Context(0) as ctx:
        opt = glf.default_options(num_samples=60, num_eigvals=8, epsilon=0.05)
        out1, zf1, info1 = ctx.image_processing(ctx.to_device(img), opt, want_float=True)
    assert tuple(r["rows"]) == (0, img.shape[0])
    assert int(r["outer"]) == info1["outer_its"]
    np.testing.assert_allclose(r["eigvals"], info1["eigvals"], rtol=1e-5)
    np.testing.assert_allclose(r["zf"], zf1.cpu().numpy(), rtol=0, atol=5e-4)
    assert np.mean(r["out"] != out1.cpu().numpy()) < 1e-3
