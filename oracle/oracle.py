"""ctypes binding of the fp64 CPU oracle (oracle/glf_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.

All matrices cross this boundary as numpy float64 arrays. Vector sets (X, phi)
are "m vectors of length n": numpy shape (m, n), C-contiguous, which is the
column-major n x m layout the C side uses (one PETSc Vec per row here).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BILATERAL, PHOTOMETRIC, SPATIAL, NLM = 0, 1, 2, 3


class Params(C.Structure):
    _fields_ = [("h_loc", C.c_double), ("h_val", C.c_double), ("kernel", C.c_int)]


class EigStats(C.Structure):
    _fields_ = [("outer_its", C.c_int), ("inner_its_total", C.c_int), ("residual", C.c_double)]


class Run(C.Structure):
    _fields_ = [
        ("p_requested", C.c_uint), ("m", C.c_uint), ("opti_gs", C.c_int),
        ("epsilon", C.c_double), ("inner_rtol", C.c_double), ("max_outer", C.c_int),
        ("seed", C.c_uint64), ("gain", C.c_double),
        ("p", C.c_uint), ("alpha", C.c_double), ("eig", EigStats),
        ("t_affinity", C.c_double), ("t_laplacian", C.c_double), ("t_eigen", C.c_double),
        ("t_nystroem", C.c_double), ("t_filter", C.c_double),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libglf_oracle.so")
    src = os.path.join(_HERE, "glf_oracle.c")
    if force or not os.path.exists(so) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "libglf_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_kernel_entry.restype = C.c_double
        _LIB.orc_kernel_entry.argtypes = [C.POINTER(Params)] + [C.c_double] * 6
        _LIB.orc_residual_norm.restype = C.c_double
    return _LIB


def _p(a, t=C.c_double):
    return a.ctypes.data_as(C.POINTER(t))


def default_params(kernel=BILATERAL):
    prm = Params()
    lib().orc_default_params(C.byref(prm))
    prm.kernel = kernel
    return prm


def _img(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    assert img.ndim == 2
    return img


def sampling(width, height, p_requested):
    n = C.c_uint(p_requested)
    ptr = C.POINTER(C.c_uint)()
    rc = lib().orc_sampling(C.c_int(width), C.c_int(height), C.byref(n), C.byref(ptr))
    if rc != 0:
        raise ValueError("orc_sampling failed")
    idx = np.ctypeslib.as_array(ptr, shape=(n.value,)).astype(np.uint32).copy()
    lib().orc_free(ptr)
    return idx


def kernel_entry(prm, a, b):
    return lib().orc_kernel_entry(C.byref(prm), *[C.c_double(x) for x in (*a, *b)])


def affinity(img, idx, prm=None, want_KB=True):
    img = _img(img)
    prm = prm or default_params()
    h, w = img.shape
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    p = idx.size
    KA = np.empty((p, p))
    KB = np.empty((p, h * w - p)) if want_KB else None
    rc = lib().orc_affinity(C.byref(prm), _p(img, C.c_uint8), w, h, C.c_uint(p), _p(idx, C.c_uint),
                            _p(KA), _p(KB) if want_KB else None)
    assert rc == 0
    return KA, KB


def degree(img, idx, prm=None, row0=0, row1=None):
    img = _img(img)
    prm = prm or default_params()
    h, w = img.shape
    row1 = h if row1 is None else row1
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    D = np.empty(idx.size)
    rc = lib().orc_degree(C.byref(prm), _p(img, C.c_uint8), w, h, row0, row1, C.c_uint(idx.size),
                          _p(idx, C.c_uint), _p(D))
    assert rc == 0
    return D


def laplacian(KA, D):
    KA = np.ascontiguousarray(KA, dtype=np.float64)
    D = np.ascontiguousarray(D, dtype=np.float64)
    LA = np.empty_like(KA)
    alpha = C.c_double()
    rc = lib().orc_laplacian(_p(KA), _p(D), C.c_uint(D.size), _p(LA), C.byref(alpha))
    assert rc == 0
    return LA, alpha.value


def random_vectors(p, m, seed):
    X = np.empty((m, p))
    lib().orc_random_vectors(_p(X), C.c_uint(p), C.c_uint(m), C.c_uint64(seed))
    return X


def orthonormalise(X):
    X = np.array(X, dtype=np.float64, order="C")
    m, n = X.shape
    norms = np.empty(m)
    lib().orc_orthonormalise(_p(X), C.c_uint(n), C.c_uint(m), _p(norms))
    return X, norms


def residual_norm(A, X):
    A = np.ascontiguousarray(A, dtype=np.float64)
    X = np.ascontiguousarray(X, dtype=np.float64)
    m, p = X.shape
    return lib().orc_residual_norm(_p(A), _p(X), C.c_uint(p), C.c_uint(m))


def block_pcg(A, B, rtol=1e-5, max_it=10000):
    A = np.ascontiguousarray(A, dtype=np.float64)
    B = np.ascontiguousarray(B, dtype=np.float64)
    m, p = B.shape
    X = np.empty_like(B)
    its = lib().orc_block_pcg(_p(A), _p(B), _p(X), C.c_uint(p), C.c_uint(m), C.c_double(rtol), max_it)
    return X, its


def inverse_power_iteration(A, m, X0, opti_gs=1, epsilon=0.1, inner_rtol=1e-5, max_outer=100000):
    A = np.ascontiguousarray(A, dtype=np.float64)
    X0 = np.ascontiguousarray(X0, dtype=np.float64)
    p = A.shape[0]
    assert X0.shape == (m, p)
    vecs = np.empty((m, p))
    vals = np.empty(m)
    st = EigStats()
    rc = lib().orc_inverse_power_iteration(_p(A), C.c_uint(p), C.c_uint(m), _p(X0), opti_gs,
                                           C.c_double(epsilon), C.c_double(inner_rtol), max_outer,
                                           _p(vecs), _p(vals), C.byref(st))
    assert rc == 0
    return vecs, vals, dict(outer_its=st.outer_its, inner_its_total=st.inner_its_total,
                            residual=st.residual)


def nystroem(img, idx, alpha, phi_A, eigvals, prm=None):
    img = _img(img)
    prm = prm or default_params()
    h, w = img.shape
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    phi_A = np.ascontiguousarray(phi_A, dtype=np.float64)
    eigvals = np.ascontiguousarray(eigvals, dtype=np.float64)
    m, p = phi_A.shape
    phi = np.empty((m, h * w))
    rc = lib().orc_nystroem(C.byref(prm), _p(img, C.c_uint8), w, h, C.c_uint(p), _p(idx, C.c_uint),
                            C.c_double(alpha), _p(phi_A), _p(eigvals), C.c_uint(m), _p(phi))
    assert rc == 0
    return phi


def permutation(phi_sf, idx, literal=False):
    phi_sf = np.ascontiguousarray(phi_sf, dtype=np.float64)
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    m, N = phi_sf.shape
    out = np.empty_like(phi_sf)
    rc = lib().orc_permutation(_p(phi_sf), _p(out), C.c_uint(N), C.c_uint(m), _p(idx, C.c_uint),
                               C.c_uint(idx.size), int(bool(literal)))
    assert rc == 0
    return out


def result_from_laplacian(img, phi, f_eigvals, gain=3.0):
    img = _img(img)
    h, w = img.shape
    phi = np.ascontiguousarray(phi, dtype=np.float64)
    f = np.ascontiguousarray(f_eigvals, dtype=np.float64)
    m = phi.shape[0]
    zf = np.empty(h * w)
    out = np.empty(h * w, dtype=np.uint8)
    rc = lib().orc_result_from_laplacian(_p(img, C.c_uint8), w, h, _p(phi), _p(f), C.c_uint(m),
                                         C.c_double(gain), _p(zf), _p(out, C.c_uint8))
    assert rc == 0
    return zf.reshape(h, w), out.reshape(h, w)


def image_processing(img, p_requested, m, opti_gs=1, epsilon=0.1, inner_rtol=1e-5,
                     max_outer=100000, seed=1, gain=3.0, prm=None):
    img = _img(img)
    prm = prm or default_params()
    h, w = img.shape
    run = Run()
    run.p_requested, run.m, run.opti_gs = p_requested, m, opti_gs
    run.epsilon, run.inner_rtol, run.max_outer = epsilon, inner_rtol, max_outer
    run.seed, run.gain = seed, gain
    lam = np.empty(max(m, 1))
    zf = np.empty(h * w)
    out = np.empty(h * w, dtype=np.uint8)
    rc = lib().orc_image_processing(C.byref(prm), _p(img, C.c_uint8), w, h, C.byref(run), _p(lam),
                                    _p(zf), _p(out, C.c_uint8))
    if rc != 0:
        raise RuntimeError("orc_image_processing failed")
    info = dict(p=run.p, m=run.m, alpha=run.alpha, outer_its=run.eig.outer_its,
                inner_its_total=run.eig.inner_its_total, residual=run.eig.residual,
                t_affinity=run.t_affinity, t_laplacian=run.t_laplacian, t_eigen=run.t_eigen,
                t_nystroem=run.t_nystroem, t_filter=run.t_filter, eigvals=lam[:run.m].copy())
    return zf.reshape(h, w), out.reshape(h, w), info


def entire_computation(img, prm=None):
    img = _img(img)
    prm = prm or default_params()
    h, w = img.shape
    zf = np.empty(h * w)
    out = np.empty(h * w, dtype=np.uint8)
    rc = lib().orc_entire_computation(C.byref(prm), _p(img, C.c_uint8), w, h, _p(zf), _p(out, C.c_uint8))
    assert rc == 0
    return zf.reshape(h, w), out.reshape(h, w)


def laplacian_rows(img, idx, D, alpha, i0, i1, prm=None):
    img = _img(img)
    prm = prm or default_params()
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    D = np.ascontiguousarray(D, dtype=np.float64)
    out = np.empty((i1 - i0, idx.size))
    rc = lib().orc_laplacian_rows(C.byref(prm), _p(img, C.c_uint8), img.shape[1], img.shape[0], C.c_uint(idx.size),
                                  _p(idx, C.c_uint), _p(D), C.c_double(alpha), C.c_uint(i0), C.c_uint(i1), _p(out))
    assert rc == 0
    return out


def matvec_rows(Arows, X):
    Arows = np.ascontiguousarray(Arows, dtype=np.float64)
    X = np.ascontiguousarray(X, dtype=np.float64)
    nrows, p = Arows.shape
    m = X.shape[0]
    Y = np.empty((m, nrows))
    rc = lib().orc_matvec_rows(_p(Arows), C.c_uint(nrows), C.c_uint(p), _p(X), C.c_uint(m), _p(Y))
    assert rc == 0
    return Y


def nystroem_rows(img, idx, alpha, phi_A, eigvals, row0, row1, prm=None):
    img = _img(img)
    prm = prm or default_params()
    h, w = img.shape
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    phi_A = np.ascontiguousarray(phi_A, dtype=np.float64)
    eigvals = np.ascontiguousarray(eigvals, dtype=np.float64)
    m, p = phi_A.shape
    out = np.empty((m, (row1 - row0) * w))
    rc = lib().orc_nystroem_rows(C.byref(prm), _p(img, C.c_uint8), w, h, row0, row1, C.c_uint(p),
                                 _p(idx, C.c_uint), C.c_double(alpha), _p(phi_A), _p(eigvals), C.c_uint(m), _p(out))
    assert rc == 0
    return out


def num_threads():
    return lib().orc_num_threads()


# ---- the PoC's alternative filters and balancing steps, numpy restatements -----------------------------------
# python/image_processing.py: nystroem :69-86, sinkhorn :90-107, orthogonalisation :110-127, smoothing_matrix :151-194,
# smoothing :197-219, sharpening :222-241. Dense fp64 on small images only (the PoC itself forms N x N matrices in
# smoothing_matrix); pinned against the PoC's own outputs in tests/golden/f4.npz (tools/gen_golden_f4.py).

def poc_nystroem(K_A, K_B):
    """:69-86 -- SVD of the symmetric K_A (descending Pi), phi = [phi_A ; K_B^T phi_A / Pi] in sample-first row order."""
    phi_A, Pi, _ = np.linalg.svd(K_A)
    return np.concatenate((phi_A, K_B.T @ (phi_A * (1.0 / Pi)))), Pi


def poc_sinkhorn_scalings(phi, Pi, iterations=100):
    """:93-98 -- the alternating scalings r, c (length = rows of phi) of K = phi diag(Pi) phi^T, never forming K."""
    r = np.ones(phi.shape[0])
    c = r
    for _ in range(iterations):
        c = np.nan_to_num(1.0 / (phi @ (Pi * (phi.T @ r))))
        r = np.nan_to_num(1.0 / (phi @ (Pi * (phi.T @ c))))
    return r, c


def poc_sinkhorn(phi, Pi, iterations=100):
    """:90-107 -- W_AB[i, :] = r_i phi_i Pi (phi c)^T for the first n rows i (n = columns of phi: the sample rows when all
    p pairs are kept), split into W_A (n x n) and W_B (n x rest)."""
    M, n = phi.shape
    r, c = poc_sinkhorn_scalings(phi, Pi, iterations)
    W_AB = ((r[:n, None] * phi[:n]) * Pi) @ (phi * c[:, None]).T
    return W_AB[:, :n], W_AB[:, n:M]


def poc_orthogonalisation(A, B):
    """:110-127 -- V = [A ; B^T] A^-1/2 phi_Q Pi_Q^-1/2 with Q = A + A^-1/2 B B^T A^-1/2; eigenvalues clipped at 1."""
    phi, Pi, _ = np.linalg.svd(A)
    A_sqrt_inv = (phi * (1.0 / np.sqrt(Pi))) @ phi.T
    Q = A + A_sqrt_inv @ B @ B.T @ A_sqrt_inv
    phi_Q, Pi_Q, _ = np.linalg.svd(Q)
    V = np.concatenate((A, B.T)) @ A_sqrt_inv @ phi_Q @ np.diag(1.0 / np.sqrt(Pi_Q))
    return V, np.minimum(Pi_Q, 1.0)


def poc_smoothing_matrix(idx, phi, Pi):
    """:151-194 -- W = I + alpha (K - D) of the dense K = phi diag(Pi) phi^T, eigenpairs of its leading p x p block W_A
    (descending) extended through W_B and brought into raster order."""
    K = (phi * Pi) @ phi.T
    D = K.sum(axis=1)
    alpha = 1.0 / D.mean()
    W = np.identity(K.shape[0]) + alpha * (K - np.diag(D))
    p = len(idx)
    L, phi_A = np.linalg.eigh(W[:p, :p])
    L, phi_A = L[::-1], phi_A[:, ::-1]
    V = np.concatenate((phi_A, W[:p, p:].T @ (phi_A * (1.0 / L))))
    return permutation(V.T, idx).T, L


def poc_smoothing_filter(y, V, L):
    """:213 -- z = V diag(L) V^T y."""
    return (V @ (L * (V.T @ np.asarray(y, dtype=np.float64).reshape(-1)))).reshape(np.shape(y))


def poc_sharpening_filter(y, V, L, beta=1.5):
    """:231-235 -- z = (1 + beta) W^2 y - beta W^3 y with W = V diag(L) V^T applied factor by factor (V is NOT orthonormal:
    the Gram matrix V^T V sits between the factors)."""
    yv = np.asarray(y, dtype=np.float64).reshape(-1)
    w2 = V @ (L * (V.T @ (V @ (L * (V.T @ yv)))))
    w3 = V @ (L * (V.T @ w2))
    return ((1.0 + beta) * w2 - beta * w3).reshape(np.shape(y))


def sharpening_weights(G, L, c, beta=1.5):
    """The same filter from the m x m Gram matrix G = V^T V and c = V^T y: z = V w with
    w = (1 + beta) L G L c - beta L G L G L c (what the HIP path evaluates: filter.hip, GLF_FILTER_SHARPEN)."""
    u = L * (G @ (L * c))
    return (1.0 + beta) * u - beta * (L * (G @ u))
