"""Condition-number estimate of a preconditioned operator (counterpart of
reference source/lanczos.py, itself a port of C code by R.P. Stevenson).

The three-term Lanczos recurrence runs on whatever vector / operator types it
is given (KronVectorMPI + LinearOperatorMPI: on the GPU); the Sturm-sequence
bisection on the small tridiagonal matrix is scalar host work."""
import time
from math import sqrt

import numpy as np
import scipy.sparse as sp


class Lanczos:
    """Smallest and largest eigenvalue of P A for symmetric positive definite
    A and P (reference lanczos.py:9-171)."""

    MAXLANCZOS = 2000
    TOLBISEC = 0.000001
    TOL = 0.0001

    def __init__(self, A, P=None, w=None, maxIterations=MAXLANCZOS, tol=TOL,
                 tolBisec=TOLBISEC):
        self.alpha = np.zeros(maxIterations)
        self.beta = np.zeros(maxIterations - 1)
        self.converged = True
        if P is None:
            P = sp.identity(A.shape[0])
        start = time.process_time()
        if w is None:
            w = 2.0 * np.random.rand(A.shape[0]) - 1.0

        # normalise the start vector in the A inner product
        v = A @ w
        nrm = sqrt(v.dot(w))
        v /= nrm
        w /= nrm
        v = P @ v
        self.alpha[0] = (A @ v).dot(w)
        lmax = lmin = self.alpha[0]

        k = 0
        while True:
            if k == maxIterations - 1:
                self.converged = False
                break
            v -= self.alpha[k] * w
            self.beta[k] = sqrt(max((A @ v).dot(v), 0.0))
            if self.beta[k] == 0.0:
                # breakdown: the Krylov space is invariant (P A = alpha I on it,
                # e.g. an exact preconditioner): the extreme Ritz values are final
                k += 1
                self.alpha[k] = self.alpha[k - 1]
                break
            w_prev, w = w, v / self.beta[k]
            v = -self.beta[k] * w_prev
            v += P @ (A @ w)
            k += 1
            self.alpha[k] = (A @ v).dot(w)

            lmax_prev, lmin_prev = lmax, lmin
            lmax, lmin = self.bisec(k, lmax, lmin, tolBisec)
            if (lmax - lmax_prev) < tol * lmax_prev and (lmin_prev -
                                                         lmin) < tol * lmin:
                break

        self.iterations = k + 1
        self.time = time.process_time() - start
        self.lmax, self.lmin = lmax, lmin
        self.alpha = np.resize(self.alpha, k)
        self.beta = np.resize(self.beta, k - 1)

    def pol(self, k, x):
        """Characteristic polynomial of the leading (k+1)x(k+1) Lanczos matrix
        by its three-term recurrence (reference lanczos.py:77-85)."""
        prev, cur = 1, self.alpha[0] - x
        for l in range(1, k + 1):
            prev, cur = cur, (self.alpha[l] -
                              x) * cur - self.beta[l - 1] * self.beta[l - 1] * prev
        return cur

    def _gershgorin(self, k):
        a, b = self.alpha, np.abs(self.beta)
        hi, lo = a[0] + b[0], a[0] - b[0]
        for l in range(1, k):
            hi = max(hi, a[l] + b[l - 1] + b[l])
            lo = min(lo, a[l] - b[l - 1] - b[l])
        hi = max(hi, a[k] + b[k - 1])
        lo = max(min(lo, a[k] - b[k - 1]), 0.0)  # spd: eigenvalues >= 0
        return hi, lo

    def bisec(self, k, ymax, zmin, tolBisec):
        """Updates the bracket [ymax, zmax] of lambda_max and [ymin, zmin] of
        lambda_min after Lanczos step k by bisection on the sign of pol
        (reference lanczos.py:20-75)."""
        zmax, ymin = self._gershgorin(k)
        sign = np.signbit

        pz = self.pol(k, zmax)
        while abs(zmax - ymax) > tolBisec * min(abs(zmax), abs(ymax)):
            mid = (ymax + zmax) / 2.0
            pm = self.pol(k, mid)
            if sign(pm) != sign(pz):
                ymax = mid
            else:
                zmax, pz = mid, pm
        py = self.pol(k, ymax)
        if sign(pz) != sign(py) and py != 0:
            ymax = zmax

        py = self.pol(k, ymin)
        while abs(zmin - ymin) > tolBisec * min(abs(zmin), abs(ymin)):
            mid = (ymin + zmin) / 2.0
            pm = self.pol(k, mid)
            if sign(pm) != sign(py):
                zmin = mid
            else:
                ymin, py = mid, pm
        pz = self.pol(k, zmin)
        if sign(pz) != sign(py) and pz != 0:
            zmin = ymin
        return (ymax, zmin)

    def cond(self):
        return self.lmax / self.lmin

    def __str__(self):
        return '{}converged\tits={}\tlmax={}\tlmin={}\tkappa={}\ttime={} s'.format(
            '' if self.converged else 'NOT ', self.iterations, self.lmax,
            self.lmin, self.cond(), self.time)
