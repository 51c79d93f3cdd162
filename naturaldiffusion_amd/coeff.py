"""Coefficient-matrix files of the reference (the drop-in data format) and their device form.

* ``.npz`` written by ``src/Utils.py:49``: ``past_xstart_coeff`` C [N,N], ``past_epsilon_coeff`` B
  [N,N+1] (or [N,N] in ``weights/step_*``), ``node_coeff`` [N+1,3] = (t, alpha, sigma); read
  POSITIONALLY like ``src/CIFAR10NaturalInference.py:273`` / ``src/ValidateNaturalInference.py:319``.
* SD3 ``.csv`` read like ``src/SD3NaturalInference.py:196`` (``index_col=0``).

``SparseRows`` is what the kernels consume: per row, the (index, value) pairs in ascending index
order, diagonal split off, uploaded once.  Host-side only (numpy); no arithmetic of the path happens here.
"""
from __future__ import annotations

import csv
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np


def load_coeff_npz(path) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    with np.load(path) as z:
        vals = [np.asarray(z[k], dtype=np.float64) for k in z.files]
    if len(vals) != 3:
        raise ValueError(f"{path}: expected three arrays (C, B, node_coeff), found {len(vals)}")
    C, B, node = vals
    if C.ndim != 2 or C.shape[0] != C.shape[1] or B.shape[0] != C.shape[0] or node.shape != (C.shape[0] + 1, 3):
        raise ValueError(f"{path}: inconsistent shapes C{C.shape} B{B.shape} node{node.shape}")
    return C, B, node


def load_sd3_csv(path) -> np.ndarray:
    with open(path, newline="") as fh:
        rows = list(csv.reader(fh))
    body = [[float(v) for v in r[1:]] for r in rows[1:] if len(r) > 1]
    W = np.asarray(body, dtype=np.float64)
    if W.ndim != 2 or W.shape[0] != W.shape[1]:
        raise ValueError(f"{path}: expected a square weight table, got {W.shape}")
    return W


@dataclass
class Row:
    start: int          # offset into the flat idx / val arrays
    n: int              # number of off-diagonal terms
    diag: float         # coefficient of column == diag index (0.0 if absent)
    total: float        # sum of the row's coefficients (numpy float64 running sum, ascending)


class SparseRows:
    """Rows of a coefficient matrix as device-resident (idx, val) lists.

    ``diag_index(k)`` tells which column of row k multiplies the value produced in the same launch
    (k for the signal matrix: the x0 just computed).  ``dense=True`` keeps zero coefficients.
    """

    def __init__(self, mat: np.ndarray, width_of_row, val_dtype, device=None, dense: bool = False,
                 diag: bool = True):
        import torch
        mat = np.asarray(mat, dtype=np.float64)
        idx: List[int] = []
        val: List[float] = []
        self.rows: List[Row] = []
        for k in range(mat.shape[0]):
            w = int(width_of_row(k))
            start, dg = len(idx), 0.0
            tot = 0
            for j in range(w):
                c = float(mat[k, j])
                tot = tot + mat[k, j]
                if diag and j == k:
                    dg = c
                    continue
                if c != 0.0 or dense:
                    idx.append(j)
                    val.append(c)
            self.rows.append(Row(start, len(idx) - start, dg, float(tot)))
        self.val_dtype = val_dtype
        np_val = {torch.float64: np.float64, torch.float32: np.float32}[val_dtype]
        self.idx_host = np.asarray(idx + [0], dtype=np.int32)
        self.val_host = np.asarray(val + [0.0], dtype=np.float64).astype(np_val)
        self.idx = torch.from_numpy(self.idx_host).to(device) if device is not None else None
        self.val = torch.from_numpy(self.val_host).to(device) if device is not None else None

    def ptrs(self, k: int):
        r = self.rows[k]
        esz = 8 if self.val_host.dtype == np.float64 else 4
        return self.idx.data_ptr() + 4 * r.start, self.val.data_ptr() + esz * r.start, r.n

    def nnz(self, k: int) -> int:
        return self.rows[k].n + (1 if self.rows[k].diag != 0.0 else 0)
