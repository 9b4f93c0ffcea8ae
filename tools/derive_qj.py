#!/usr/bin/env python3
"""Re-derive the Q_J change-of-basis constants without lie_learn and compare them with the
closed forms shipped in oracle/rf_oracle.py::q_j_constants (and the HIP constant tables).

Method: the Wigner-D matrices are those implied by the reference's own
real spherical harmonics, D_J(R) Y_J(x) = Y_J(R x) (least squares over random x); Q_J spans the
1-dimensional null space of  kron(D_do (x) D_di, I) - kron(I, D_J^T)  stacked over random
rotations -- the same Sylvester system as ea/from_se3cnn/utils_steerable.py:46-68.
Sign rule: first non-zero entry (row-major) positive."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import rf_oracle as O  # noqa: E402

torch.set_default_dtype(torch.float64)


def rand_rot(g):
    q, r = torch.linalg.qr(torch.randn(3, 3, generator=g))
    q = q * torch.sign(torch.diagonal(r))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q


def wigner(J, R, g):
    x = torch.randn(200, 3, generator=g)
    Yx = O.real_sh(x)[J]
    YRx = O.real_sh(x @ R.t())[J]
    return torch.linalg.lstsq(Yx, YRx).solution.t()  # D with D Y(x) = Y(Rx)


def kron(a, b):
    return torch.kron(a.contiguous(), b.contiguous())


def derive(di, do, J, g):
    rows = []
    for _ in range(5):
        R = rand_rot(g)
        Rt = kron(wigner(do, R, g), wigner(di, R, g))
        DJ = wigner(J, R, g)
        rows.append(kron(Rt, torch.eye(2 * J + 1)) - kron(torch.eye(Rt.shape[0]), DJ.t()))
    A = torch.cat(rows)
    _, s, vh = torch.linalg.svd(A)
    assert (s < 1e-8).sum() == 1, s[-3:]
    q = vh[-1].view((2 * do + 1) * (2 * di + 1), 2 * J + 1)
    nz = q.flatten()[q.flatten().abs() > 1e-9][0]
    return q * torch.sign(nz)


def main():
    g = torch.Generator().manual_seed(0)
    Q = O.q_j_constants()
    worst = 0.0
    for (di, do), lst in Q.items():
        for n, J in enumerate(range(abs(di - do), di + do + 1)):
            q = derive(di, do, J, g)
            err = (q - lst[n]).abs().max().item()
            worst = max(worst, err)
            print(f"(d_in={di}, d_out={do}, J={J})  shape {tuple(q.shape)}  |derived - closed form|_max = {err:.2e}")
    print("worst", worst)
    return worst


if __name__ == "__main__":
    sys.exit(0 if main() < 1e-9 else 1)
