"""CPU-side checks of the drop-in boundary: libglf.so loads, exports every
symbol include/glf.h declares, and the host-only entry points (sampling, X0
stream, synthetic image, PNG codec) agree with the oracle / the golden files.
No device compute here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import glf
import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# CRC32 of the synthetic benchmark images glf_synth_image(size, size, seed 0) ("commit CRC32 of each generated
# image"): the workload of BASELINE configs 3-5 is pinned byte for byte. libm's sin / log / cos enter the generator, so a
# different libm could in principle move a pixel; the GPU tests assert the same values on the GPU box.
SYNTH_CRC32 = {64: 0x4ae65dbf, 1024: 0x3d97f4f9, 2048: 0x40fa122b, 4096: 0xdc7203de}


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "glf.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(glf_[A-Za-z0-9_]+)\s*\(", header))
    declared -= {"glf_comm", "glf_mat", "glf_ctx"}
    assert len(declared) >= 30
    lib = C.CDLL(glf.LIB_PATH)
    missing = [name for name in sorted(declared) if not hasattr(lib, name)]
    assert not missing, "declared in glf.h but not exported: %s" % missing
    assert set(glf.EXPORTS) == declared


def test_struct_sizes_match_header_layout():
    # glf_options.struct_size is checked by the library itself; the default must round-trip
    opt = glf.default_options()
    assert opt.struct_size == C.sizeof(glf.Options)
    assert (opt.sample_frac, opt.opti_gs, opt.epsilon, opt.inner_rtol) == (0.01, 1, 0.1, 1e-5)
    assert (opt.gain, opt.h_loc, opt.h_val, opt.kernel, opt.filter_pow) == (3.0, 40.0, 30.0, 0, 1)
    assert glf._lib.glf_strerror(0) == b"ok"
    assert b"converge" in glf._lib.glf_strerror(glf.ERR_NOCONV)


@pytest.mark.parametrize("w,h,p_req", [(100, 100, 100), (450, 300, 50), (450, 300, 1350), (512, 512, 2621),
                                        (53, 37, 20), (32, 32, 10), (1024, 1024, 5242), (7, 5, 3)])
def test_sampling_matches_oracle(w, h, p_req):
    np.testing.assert_array_equal(glf.Sampling(w, h, p_req), orc.sampling(w, h, p_req))


def test_sampling_golden_and_errors(golden):
    g = golden("sampling.npz")
    np.testing.assert_array_equal(glf.Sampling(450, 300, 50), g["cat_50_idx"])
    assert glf.Sampling(4096, 4096, int(4096 * 4096 * 0.005)).size == 85264
    with pytest.raises(glf.GlfError):
        glf.Sampling(4, 4, 1000)
    with pytest.raises(glf.GlfError):
        glf.Sampling(0, 4, 2)


def test_random_vectors_match_oracle_stream():
    for p, m, seed in ((17, 3, 1), (2601, 16, 1), (100, 5, 12345)):
        a = glf.random_vectors(p, m, seed)
        b = orc.random_vectors(p, m, seed)
        np.testing.assert_array_equal(a, b)
        assert a.min() >= 0.0 and a.max() < 1.0


def test_synth_image_is_deterministic():
    a = glf.synth_image(256, 192, seed=0)
    b = glf.synth_image(256, 192, seed=0)
    c = glf.synth_image(256, 192, seed=1)
    np.testing.assert_array_equal(a, b)
    assert (a != c).mean() > 0.5
    assert a.dtype == np.uint8 and a.shape == (192, 256)
    assert 90 < a.mean() < 165 and a.std() > 25  # noisy mid-grey image
    import zlib
    for size in (64, 1024, 2048):   # (4096: asserted by tests/test_gpu_large.py and bench.py on the GPU box)
        assert zlib.crc32(glf.synth_image(size, size, seed=0).tobytes()) == SYNTH_CRC32[size], size
    with pytest.raises(glf.GlfError):   # hpc/sampling.c:16-18 admits no sample on a one-row / one-column image
        glf.Sampling(1, 50, 5)
    with pytest.raises(glf.GlfError):
        glf.Sampling(50, 1, 5)


def test_png_codec_against_pillow(tmp_path, png):
    from PIL import Image
    for name in ("test.png", "cat_small.png", "barbara.png"):
        ours = glf.read_png(os.path.join(ROOT, "tests", "golden", name))
        np.testing.assert_array_equal(ours, png(name))
    # write -> read back with Pillow and with ourselves
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (37, 53)).astype(np.uint8)
    path = str(tmp_path / "rt.png")
    glf.write_png(path, img)
    back = Image.open(path)
    assert back.mode == "L"
    np.testing.assert_array_equal(np.array(back), img)
    np.testing.assert_array_equal(glf.read_png(path), img)
    # all five scanline filters (Pillow picks adaptively at optimize=True)
    grad = (np.add.outer(np.arange(64), np.arange(80)) * 3 % 256).astype(np.uint8)
    p2 = str(tmp_path / "grad.png")
    Image.fromarray(grad).save(p2, optimize=True)
    np.testing.assert_array_equal(glf.read_png(p2), grad)


def test_png_rgb_to_gray_and_rejects(tmp_path):
    from PIL import Image
    # RGB(A) -> gray exactly as libpng 1.6.37 does it (hpc/read_img.c:47-50); the
    # *_gray_libpng.png fixtures were produced by real libpng (tools/png_libpng_golden.c):
    # pixel_mountains.png carries an sRGB chunk (gamma-table path), the other two no gamma.
    gdir = os.path.join(ROOT, "tests", "golden")
    for name in ("pixel_mountains", "rgb_nogamma", "rgba_nogamma"):
        ours = glf.read_png(os.path.join(gdir, name + ".png"))
        expect = np.array(Image.open(os.path.join(gdir, name + "_gray_libpng.png")))
        np.testing.assert_array_equal(ours, expect)
    # 16-bit, palette and gray+alpha are rejected instead of misread (survey quirk Q13)
    p16 = str(tmp_path / "g16.png")
    Image.fromarray((np.arange(64 * 64).reshape(64, 64) * 16).astype(np.uint16)).save(p16)
    with pytest.raises(glf.GlfError):
        glf.read_png(p16)
    pla = str(tmp_path / "la.png")
    Image.fromarray(np.zeros((8, 8, 2), dtype=np.uint8), mode="LA").save(pla)
    with pytest.raises(glf.GlfError):
        glf.read_png(pla)
    with pytest.raises(glf.GlfError):
        glf.read_png(str(tmp_path / "missing.png"))
    bad = str(tmp_path / "bad.png")
    open(bad, "wb").write(b"\x89PNG\r\n\x1a\n" + b"\0" * 64)
    with pytest.raises(glf.GlfError):
        glf.read_png(bad)


def test_context_needs_a_gpu_and_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = C.c_void_p()
    rc = glf._lib.glf_ctx_create(C.byref(ctx), 0, None)
    assert rc == glf.ERR_NODEVICE and not ctx.value


def test_png_rgb_codec(tmp_path):
    """The colour variant of the codec (rows of 3 * width bytes): the reference's RGB fixture against Pillow, a gray file
    replicated into the three channels, and a write / read round trip."""
    from PIL import Image
    src = os.path.join(ROOT, "tests", "golden", "pixel_mountains.png")
    rgb = glf.read_png_rgb(src)
    np.testing.assert_array_equal(rgb, np.array(Image.open(src).convert("RGB")))
    gray = glf.read_png_rgb(os.path.join(ROOT, "tests", "golden", "test.png"))
    ref = np.array(Image.open(os.path.join(ROOT, "tests", "golden", "test.png")))
    for ch in range(3):
        np.testing.assert_array_equal(gray[:, :, ch], ref)
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    rows = (C.POINTER(C.c_uint8) * 37)()
    flat = np.ascontiguousarray(img)
    for r in range(37):
        rows[r] = C.cast(flat.ctypes.data + r * 53 * 3, C.POINTER(C.c_uint8))
    path = str(tmp_path / "rgb.png")
    assert glf._lib.glf_write_png_rgb(path.encode(), rows, C.c_uint(53), C.c_uint(37)) == 0
    np.testing.assert_array_equal(glf.read_png_rgb(path), img)
    np.testing.assert_array_equal(np.array(Image.open(path)), img)


def test_random_sampler_contract():
    """The PoC's random sampler (python/sampling/random.py:8-16) through the C-ABI: exactly the requested number of distinct
    pixel indices, ascending, reproducible per seed, different between seeds; degenerate requests are refused."""
    a = glf.RandomSampling(64, 48, 200, seed=7)
    assert a.size == 200 and np.all(np.diff(a.astype(np.int64)) > 0) and a.max() < 64 * 48
    np.testing.assert_array_equal(a, glf.RandomSampling(64, 48, 200, seed=7))
    assert not np.array_equal(a, glf.RandomSampling(64, 48, 200, seed=8))
    assert glf.RandomSampling(8, 8, 64, seed=1).size == 64                      # every pixel
    with pytest.raises(glf.GlfError):
        glf.RandomSampling(8, 8, 65)
