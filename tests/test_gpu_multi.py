"""N > 1 through the C path (`mpirun -n N` of hpc/image_processing.c:30-76): glf_multi_* drives N GPU ranks from one
process -- one context and one host thread per rank, pixel rows sharded, the eigen-solve row-sharded with all-reduces of
the inner products / Gram blocks and an all-gather per operator application, collectives issued by the library itself.
On this one-GPU box the ranks share cuda:0 through the LOOPBACK backend (RCCL refuses two ranks on one device); RCCL
itself is exercised with one rank (ncclCommInitAll / ncclCommInitRank + every collective of the path on a one-rank world).
Every run must reproduce the single-context run: same iteration count, eigenvalues to 1e-5, z to 5e-4 grey levels (the
sums are taken in another order, nothing else changes)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import glf  # noqa: E402
from conftest import psnr  # noqa: E402


def _single(img, opt):
    with glf.Context(0) as ctx:
        out, zf, info = ctx.image_processing(ctx.to_device(img), opt, want_float=True)
        return out.cpu().numpy(), zf.cpu().numpy(), info


def _check(img, opt, n, backend, devices):
    out1, zf1, info1 = _single(img, opt)
    with glf.Multi(n, devices=devices, backend=backend) as world:
        out, zf, infos = world.image_processing(img, opt, want_float=True)
        out_b, zf_b, _ = world.image_processing(img, opt, want_float=True)          # second call on the same world: cached buffers
    h = img.shape[0]
    assert [(i["row0"], i["row1"]) for i in infos] == [glf.shard_rows(h, r, n) for r in range(n)]
    for i in infos:
        assert (i["p"], i["m"], i["outer_its"]) == (info1["p"], info1["m"], info1["outer_its"])
        np.testing.assert_allclose(i["eigvals"], info1["eigvals"], rtol=1e-5)
        np.testing.assert_array_equal(i["eigvals"], infos[0]["eigvals"])            # all-reduced sums: identical on every rank
        assert i["alpha"] == infos[0]["alpha"]
    assert infos[0]["alpha"] == pytest.approx(info1["alpha"], rel=1e-7)     # (the grid degree sums f32 chunks of 512 image rows: a shard moves the chunk boundaries)
    np.testing.assert_allclose(zf, zf1, rtol=0, atol=5e-4)
    assert np.mean(out != out1) < 1e-3 and psnr(out, out1) >= 60.0
    np.testing.assert_array_equal(zf_b.view(np.int32), zf.view(np.int32))          # run-to-run reproducible
    return infos


def _env_paths(monkeypatch, paths):
    """direct: entry-by-entry kernels, stored L_A; grid / rank: the grid-factored forms; band: entry by entry within the radius (the default at benchmark sizes)"""
    monkeypatch.setenv("GLF_NYS_PATH", paths)
    monkeypatch.setenv("GLF_DEG_PATH", "direct" if paths == "direct" else "grid")
    monkeypatch.setenv("GLF_MV_PATH", {"direct": "dense", "grid": "grid", "rank": "rank", "band": "band"}[paths])


@pytest.mark.parametrize("n", [1, 2, 3])
@pytest.mark.parametrize("paths", ["direct", "grid", "rank", "band"])
def test_loopback_ranks_on_one_device_match_single_context(n, paths, monkeypatch):
    _env_paths(monkeypatch, paths)
    img = glf.synth_image(96, 80, seed=4)
    opt = glf.default_options(num_samples=60, num_eigvals=8, epsilon=0.05)
    _check(img, opt, n, glf.MULTI_LOOPBACK, [0] * n)


def test_band_form_replicates_the_eigen_solve_unless_told_otherwise(monkeypatch):
    """With the band form every rank runs the eigen-solve on all rows by default (a sweep costs less than the all-gather of its
    operand; glf_stats.eigen_sharded = 0) and only the pixel rows are sharded; GLF_EIG_SHARD=1 keeps the row-sharded solve with
    its collectives, =0 replicates it for the factored forms too. All against the single-context result."""
    img = glf.synth_image(96, 80, seed=4)
    opt = glf.default_options(num_samples=60, num_eigvals=8, epsilon=0.05)
    for paths, env, want in (("band", None, 0), ("band", "1", 1), ("rank", None, 1), ("rank", "0", 0)):
        _env_paths(monkeypatch, paths)
        if env is None:
            monkeypatch.delenv("GLF_EIG_SHARD", raising=False)
        else:
            monkeypatch.setenv("GLF_EIG_SHARD", env)
        infos = _check(img, opt, 3, glf.MULTI_LOOPBACK, [0] * 3)
        assert [i["eigen_sharded"] for i in infos] == [want] * 3, (paths, env)
    monkeypatch.delenv("GLF_EIG_SHARD", raising=False)


def test_loopback_more_ranks_than_grid_rows_and_odd_shards(monkeypatch):
    """5 ranks on a 53 x 37 image (p = 24: a 4 x 6 sample grid -- fewer grid rows than ranks, so some ranks own no row of the
    eigen-solve; image row shards of 7 or 8 rows) in both kernel families."""
    img = glf.synth_image(53, 37, seed=3)
    opt = glf.default_options(num_samples=20, num_eigvals=5, epsilon=0.1)
    for paths in ("direct", "grid", "rank", "band"):
        _env_paths(monkeypatch, paths)
        _check(img, opt, 5, glf.MULTI_LOOPBACK, [0] * 5)


def test_loopback_two_ranks_1024_default_paths():
    """1024 x 1024, 0.5 %, m = 64 on 2 ranks with the kernels chosen by default at that size (grid-factored degree pass, band-form
    Nystroem pass; stored L_A column blocks for the p = 5329 eigen-solve)."""
    img = glf.synth_image(1024, 1024, seed=5)
    opt = glf.default_options(num_samples=int(1024 * 1024 * 0.005), num_eigvals=64, epsilon=0.1)
    infos = _check(img, opt, 2, glf.MULTI_LOOPBACK, [0, 0])
    assert infos[0]["nystroem_path"] == 4   # the band form (the samples within the radius of each pixel only)


def test_rccl_one_rank_world():
    """RCCL inside the library: ncclCommInitAll on one device, then every collective of the path (all-reduce f64 / f32,
    in-place all-gather) on the context's stream -- on a one-rank world, which is all one GPU allows."""
    img = glf.synth_image(96, 80, seed=4)
    opt = glf.default_options(num_samples=60, num_eigvals=8, epsilon=0.05)
    _check(img, opt, 1, glf.MULTI_RCCL, [0])


def test_rccl_init_rank_on_a_context():
    """One process per GPU (bench.py): unique id -> ncclCommInitRank -> the native callbacks, forced on for the one rank."""
    img = glf.synth_image(96, 80, seed=4)
    opt = glf.default_options(num_samples=60, num_eigvals=8, epsilon=0.05)
    out1, zf1, info1 = _single(img, opt)
    uid = glf.rccl_unique_id()
    assert len(uid) == glf.RCCL_ID_BYTES
    with glf.Context(0) as ctx:
        ctx.set_comm_rccl(0, 1, uid, force=True)
        out, zf, info = ctx.image_processing(ctx.to_device(img), opt, want_float=True)
        assert info["outer_its"] == info1["outer_its"]
        np.testing.assert_allclose(info["eigvals"], info1["eigvals"], rtol=1e-5)
        np.testing.assert_allclose(zf.cpu().numpy(), zf1, rtol=0, atol=5e-4)
    with pytest.raises(glf.GlfError):
        glf.Multi(2, devices=[0, 0], backend=glf.MULTI_RCCL)       # RCCL refuses one device twice: loud, not a hang


def test_loopback_two_ranks_nlm_kernel():
    """The non-local-means affinity through the sharded path: degree partial sums per rank, L_A column blocks generated by
    k_nlm_matrix(col0, ncols), row-sharded eigen-solve, Nystroem per pixel-row shard."""
    img = glf.synth_image(96, 80, seed=4)
    opt = glf.default_options(num_samples=60, num_eigvals=8, epsilon=0.05)
    opt.kernel, opt.h_val = glf.KERNEL_NLM, 3.0
    _check(img, opt, 2, glf.MULTI_LOOPBACK, [0, 0])


def test_rccl_two_gpus_match_single_context():
    """N > 1 over real RCCL (ncclCommInitAll, one rank thread per GPU): skipped on a one-GPU box -- the driver's multi-GPU node
    runs it. Same comparison as the loopback tests: the sharded result against one context on the whole image."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    img = glf.synth_image(1024, 1024, seed=5)
    opt = glf.default_options(num_samples=int(1024 * 1024 * 0.005), num_eigvals=64, epsilon=0.1)
    infos = _check(img, opt, 2, glf.MULTI_RCCL, [0, 1])
    assert infos[0]["row1"] == 512 and infos[1]["row0"] == 512
