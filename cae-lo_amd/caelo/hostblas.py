"""The BLAS / LAPACK entry points of THIS process's NumPy, handed to libcaelo's host half (csrc/certify.hip).

The reference fits every RANSAC hypothesis with NumPy (Match.py:141-157): ``np.dot`` is cblas_sgemm / cblas_sgemv of
whatever BLAS NumPy was built with, ``np.linalg.svd`` is that library's dgesdd.  The bits of R, T and of the residuals depend
on that library (fused or unfused multiply-adds, accumulation order), so the host half calls the same functions through the
same shared object instead of re-deriving their arithmetic.  ``bind()`` finds the object among the libraries NumPy has
loaded (pip wheels: ``numpy.libs/libscipy_openblas64_*.so`` with ``scipy_`` prefixes and ILP64 ``64_`` suffixes; conda:
libopenblas / libcblas + liblapack or MKL with plain LP64 names), verifies on random samples that the bound calls reproduce
``np.dot`` / ``np.linalg.svd`` bit for bit, and refuses the binding otherwise -- there is no silent substitute.
"""
import ctypes as C
import glob
import os

import numpy as np

# (cblas_sgemm, cblas_sgemv, dgesdd Fortran symbol, ilp64)
_CANDIDATES = [
    ("scipy_cblas_sgemm64_", "scipy_cblas_sgemv64_", "scipy_dgesdd_64_", 1),
    ("cblas_sgemm64_", "cblas_sgemv64_", "dgesdd_64_", 1),
    ("cblas_sgemm_64", "cblas_sgemv_64", "dgesdd_64", 1),
    ("cblas_sgemm", "cblas_sgemv", "dgesdd_", 0),
]
_bound = None


def _loaded_libraries():
    """Shared objects this process has mapped that can hold a BLAS (NumPy's own first)."""
    import numpy.linalg  # noqa: F401  (maps the LAPACK NumPy uses)
    paths = []
    nlibs = os.path.join(os.path.dirname(os.path.dirname(np.__file__)), "numpy.libs")
    paths += sorted(glob.glob(os.path.join(nlibs, "*openblas*.so*")))
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                p = line.rsplit(None, 1)[-1]
                b = os.path.basename(p)
                if ".so" in b and any(k in b.lower() for k in ("openblas", "mkl_rt", "libblas", "libcblas", "liblapack", "blis", "flexiblas")):
                    if p not in paths:
                        paths.append(p)
    except OSError:
        pass
    # NumPy's own libraries first; a foreign BLAS that happens to be mapped (scipy.libs, torch) only as a last resort -- and only
    # if it passes the bit-for-bit check below
    own = os.path.dirname(os.path.dirname(np.__file__))
    paths.sort(key=lambda p: (0 if p.startswith(own) else 1))
    return paths


def _addresses(path):
    try:
        lib = C.CDLL(path)
    except OSError:
        return None
    for g, v, d, ilp in _CANDIDATES:
        try:
            fg, fv = getattr(lib, g), getattr(lib, v)
        except AttributeError:
            continue
        fd = None
        for holder in (lib,):
            try:
                fd = getattr(holder, d)
            except AttributeError:
                fd = None
        if fd is None:
            continue
        return (C.cast(fg, C.c_void_p).value, C.cast(fv, C.c_void_p).value, C.cast(fd, C.c_void_p).value, ilp, lib)
    return None


def _verify(lib, n_samples=400):
    """The bound entry points must give NumPy's bits, call by call: the five BLAS / LAPACK calls the host half makes
    (caelo_host_blas_probe issues each exactly as csrc/certify.hip does) against ``np.dot`` / ``np.linalg.svd`` on the same
    operands -- covariance products with k = 4 and k = n, 3 x 3 SVDs of well-conditioned, nearly planar and rank-deficient
    matrices, the 3 x 3 products, the matrix-vector product and the residual product.  (That the host half COMPOSES these calls
    the way Match.py:138-158 does is a test's business: tests/test_host_exact.py compares it with the oracle's SolveRT.)"""
    rng = np.random.RandomState(20240229)
    vp = lambda x: C.c_void_p(x.ctypes.data)
    for t in range(n_samples):
        n = 4 if t % 4 else int(rng.randint(5, 400))
        p1 = (rng.standard_normal((n, 3)) * [30.0, 30.0, 0.5 if t % 3 else 1e-3]).astype(np.float32)
        ang = rng.uniform(-0.1, 0.1)
        rot = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32)
        p0 = (p1 @ rot.T + rng.standard_normal((n, 3)).astype(np.float32) * np.float32(0.05 if t % 5 else 5.0)).astype(np.float32)
        if t % 7 == 0:
            p0[1] = p0[0]          # a repeated point: rank-2 covariance
        if t % 11 == 0:
            p0[:, 2] = p0[0, 2]    # exactly coplanar
        a0 = np.ascontiguousarray(p0 - np.mean(p0, axis=0).reshape(1, 3))
        a1 = np.ascontiguousarray(p1 - np.mean(p1, axis=0).reshape(1, 3))
        H = np.empty((3, 3), np.float32)
        if lib.caelo_host_blas_probe(0, vp(a1), vp(a0), n, vp(H)) != 0 or not np.array_equal(H, np.dot(a1.T, a0)):
            return False
        uv = np.empty((2, 3, 3), np.float32)
        U, _, V = np.linalg.svd(H)
        if lib.caelo_host_blas_probe(1, vp(H), None, 1, vp(uv)) != 0 or U.dtype != np.float32 or not (np.array_equal(uv[0], U) and np.array_equal(uv[1], V)):
            return False
        R = np.empty((3, 3), np.float32)
        if lib.caelo_host_blas_probe(2, vp(uv[1]), vp(uv[0]), 1, vp(R)) != 0 or not np.array_equal(R, np.dot(V.T, U.T)):
            return False
        m1 = np.ascontiguousarray(np.mean(p1, axis=0).reshape(1, 3))
        y = np.empty(3, np.float32)
        if lib.caelo_host_blas_probe(3, vp(R), vp(m1), 1, vp(y)) != 0 or not np.array_equal(y, np.dot(R, m1.T).ravel()):
            return False
        x = np.empty((3, n), np.float32)
        if lib.caelo_host_blas_probe(4, vp(R), vp(p1), n, vp(x)) != 0 or not np.array_equal(x, np.dot(R, p1.T)):
            return False
    return True


def bind(lib):
    """Bind libcaelo's host half to NumPy's BLAS.  Returns a description dict; raises CaeloError when no library of this process
    reproduces NumPy bit for bit."""
    global _bound
    if _bound is not None:
        return _bound
    from ._ffi import CaeloError
    tried = []
    for path in _loaded_libraries():
        adr = _addresses(path)
        if adr is None:
            tried.append((path, "no cblas_sgemm / cblas_sgemv / dgesdd"))
            continue
        g, v, d, ilp, keep = adr
        if lib.caelo_host_bind_blas(C.c_void_p(g), C.c_void_p(v), C.c_void_p(d), ilp) != 0:
            tried.append((path, "workspace query failed"))
            continue
        if _verify(lib):
            _bound = {"library": path, "ilp64": bool(ilp), "handle": keep}
            return _bound
        lib.caelo_host_unbind_blas()   # a library that does not reproduce NumPy must not stay bound behind the exception
        tried.append((path, "results differ from np.dot / np.linalg.svd"))
    raise CaeloError("no BLAS of this process reproduces NumPy bit for bit: %s" % (tried,))


def describe():
    return None if _bound is None else {k: v for k, v in _bound.items() if k != "handle"}
