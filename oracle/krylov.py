"""PCG and the Lanczos condition-number estimator on NumPy arrays.
Test infrastructure (see oracle/__init__.py)."""
from math import sqrt

import numpy as np


def _dot(a, b):
    return float(np.dot(a.reshape(-1), b.reshape(-1)))


def pcg(T, P, b, w0=None, kmax=100000, eps=1e-6, callback=None):
    """Preconditioned CG with the algebraic stopping rule r.Pr < eps^2
    (reference linalg.py:6-42).  T, P: callables array -> array.
    Returns (w, iters, history) where history[k] = r.Pr after k iterations."""
    w = np.zeros_like(b) if w0 is None else w0
    iters, hist = 0, []
    if _dot(b, b) == 0:  # linalg.py:17-18
        return w, iters, hist
    r = b - T(w)
    p = P(r)
    abs_r = _dot(r, p)
    hist.append(abs_r)
    if abs_r < eps * eps:  # linalg.py:24
        return w, iters, hist
    for k in range(1, kmax):
        iters += 1
        t = T(p)
        alpha = abs_r / _dot(p, t)
        w += alpha * p
        r -= alpha * t
        if callback is not None:
            callback(w, r, k)
        z = P(r)
        abs_r_old, abs_r = abs_r, _dot(r, z)
        hist.append(abs_r)
        if abs_r < eps * eps:
            break
        p *= abs_r / abs_r_old
        p += z
    return w, iters, hist


class Lanczos:
    """lambda_max / lambda_min of P A by preconditioned Lanczos with Sturm
    bisection (reference lanczos.py:9-171)."""
    def __init__(self, A, P, w, maxIterations=2000, tol=1e-4, tolBisec=1e-6):
        self.alpha = np.zeros(maxIterations)
        self.beta = np.zeros(maxIterations - 1)
        self.converged = True
        w = w.copy()
        v = A(w)  # lanczos.py:109-112
        nrm = sqrt(_dot(v, w))
        v = v / nrm
        w = w / nrm
        v = P(v)
        u = A(v)
        self.alpha[0] = _dot(u, w)
        lmax = lmin = self.alpha[0]
        k = 0
        while True:  # lanczos.py:121-150
            if k == maxIterations - 1:
                self.converged = False
                break
            v = v - self.alpha[k] * w
            u = A(v)
            self.beta[k] = sqrt(_dot(u, v))
            w, v = v / self.beta[k], -self.beta[k] * w
            u = P(A(w))
            v = v + u
            k += 1
            u = A(v)
            self.alpha[k] = _dot(u, w)
            lmax_old, lmin_old = lmax, lmin
            lmax, lmin = self.bisec(k, lmax, lmin, tolBisec)
            if (lmax - lmax_old) < tol * lmax_old and (lmin_old -
                                                       lmin) < tol * lmin:
                break
        self.iterations = k + 1
        self.lmax, self.lmin = lmax, lmin
        self.alpha = np.resize(self.alpha, k)  # as lanczos.py:158-159
        self.beta = np.resize(self.beta, k - 1)

    def pol(self, k, x):
        """Sturm polynomial of the leading (k+1) x (k+1) tridiagonal
        (lanczos.py:77-85)."""
        r, p = 1, self.alpha[0] - x
        for l in range(1, k + 1):
            p, r = (self.alpha[l] - x) * p - self.beta[l - 1]**2 * r, p
        return p

    def bisec(self, k, ymax, zmin, tolBisec):
        """lanczos.py:20-75."""
        a, b = self.alpha, np.abs(self.beta)
        zmax, ymin = a[0] + b[0], a[0] - b[0]
        for l in range(1, k):
            zmax = max(zmax, a[l] + b[l - 1] + b[l])
            ymin = min(ymin, a[l] - b[l - 1] - b[l])
        zmax = max(zmax, a[k] + b[k - 1])
        ymin = max(min(ymin, a[k] - b[k - 1]), 0.0)

        pz = self.pol(k, zmax)
        while abs(zmax - ymax) > tolBisec * min(abs(zmax), abs(ymax)):
            x = (ymax + zmax) / 2.0
            px = self.pol(k, x)
            if np.signbit(px) != np.signbit(pz):
                ymax = x
            else:
                zmax, pz = x, px
        py = self.pol(k, ymax)
        if np.signbit(pz) != np.signbit(py) and py != 0:
            ymax = zmax

        py = self.pol(k, ymin)
        while abs(zmin - ymin) > tolBisec * min(abs(zmin), abs(ymin)):
            x = (ymin + zmin) / 2.0
            px = self.pol(k, x)
            if np.signbit(px) != np.signbit(py):
                zmin = x
            else:
                ymin, py = x, px
        pz = self.pol(k, zmin)
        if np.signbit(pz) != np.signbit(py) and pz != 0:
            zmin = ymin
        return ymax, zmin

    def cond(self):
        return self.lmax / self.lmin
