"""Input side of the hot path: parity-check matrices and synthetic workloads.

    parity_check_matrix(n, wr, wc)   src/parity_generator.jl:21-45  (Gallager regular LDPC)
    save_pcm / load_pcm              src/parity_generator.jl:47-54

The reference draws its column shuffles from Julia's unseeded global RNG
(:41), so its matrices are not reproducible; this generator keeps the same
*structure* (block 0 = consecutive runs of ``wr`` ones, blocks 1..wc-1 = column
permutations of block 0) with a documented, seeded PRNG: splitmix64 driving a
Fisher-Yates shuffle, seed = ``seed + block index``.

Pure host code (numpy); nothing here touches the GPU.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import scipy.sparse as sp

DEFAULT_SEED = 0x4C445043  # "LDPC"
_M64 = (1 << 64) - 1


class SplitMix64:
    """splitmix64 (Steele, Lea, Flood 2014); the documented PRNG of this package."""

    def __init__(self, seed: int):
        self.x = seed & _M64

    def next(self) -> int:
        self.x = (self.x + 0x9E3779B97F4A7C15) & _M64
        z = self.x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        return z ^ (z >> 31)


def shuffled_indices(n: int, seed: int) -> np.ndarray:
    """Fisher-Yates permutation of 0..n-1 (stands in for `shuffle(1:end)`, parity_generator.jl:41)."""
    rng = SplitMix64(seed)
    p = list(range(n))
    for i in range(n - 1, 0, -1):
        j = rng.next() % (i + 1)
        p[i], p[j] = p[j], p[i]
    return np.asarray(p, dtype=np.int64)


def parity_check_csc(n: int, wr: int, wc: int, seed: int = DEFAULT_SEED) -> sp.csc_matrix:
    """Gallager (wc, wr)-regular parity-check matrix as a sparse ``(n*wc/wr) x n`` pattern.

    Block b, column c has its single one in row ``b*block_size + perm_b[c] // wr``
    where perm_0 is the identity (parity_generator.jl:32-42: ``block[:, shuffle(1:end)]``
    puts old column ``perm[c]`` at position ``c``)."""
    if n % wr != 0:
        raise AssertionError("n % wr == 0")  # parity_generator.jl:25
    n_equations = (n * wc) // wr
    block_size = n_equations // wc
    rows = np.empty((wc, n), dtype=np.int64)
    rows[0] = np.arange(n) // wr
    for b in range(1, wc):
        perm = shuffled_indices(n, seed + b)
        rows[b] = b * block_size + perm // wr
    indices = rows.T.reshape(-1)               # column-major: per bit, blocks (= rows) ascending
    indptr = np.arange(0, n * wc + 1, wc, dtype=np.int64)
    data = np.ones(n * wc, dtype=np.bool_)
    return sp.csc_matrix((data, indices, indptr), shape=(n_equations, n))


def parity_check_matrix(n: int, wr: int, wc: int, seed: int = DEFAULT_SEED) -> np.ndarray:
    """`parity_check_matrix(n, wr, wc)` (parity_generator.jl:21-45) as a dense bool matrix
    (the reference returns a BitMatrix)."""
    return np.asarray(parity_check_csc(n, wr, wc, seed).todense()).astype(np.bool_)


def save_pcm(H, file_path) -> None:
    """`save_pcm(H, file_path)` (parity_generator.jl:47-49): `writedlm(file_path, Int.(H))`,
    i.e. tab-delimited 0/1 rows."""
    A = np.asarray(H.todense() if sp.issparse(H) else H).astype(np.int64)
    with open(file_path, "w") as f:
        for row in A:
            f.write("\t".join(str(int(v)) for v in row) + "\n")


def load_pcm(file_path) -> np.ndarray:
    """`load_pcm(file_path)` (parity_generator.jl:51-54): `Int.(readdlm(file_path))`."""
    rows = []
    with open(file_path) as f:
        for line in f:
            line = line.strip()
            if line:
                rows.append([int(float(t)) for t in line.replace(",", " ").split()])
    return np.asarray(rows, dtype=np.int64)


def bivariate_bicycle_72_12_6() -> Tuple[np.ndarray, np.ndarray]:
    """[[72,12,6]] bivariate-bicycle code (BASELINE config 5; not in the reference).

    l = m = 6, x = S_6 (x) I_6, y = I_6 (x) S_6, A = x^3 + y + y^2, B = y^3 + x + x^2,
    H_X = [A | B], H_Z = [B' | A'].  Returns (H_X, H_Z), each 36 x 72, row weight 6."""
    ell = m = 6
    S_l = np.roll(np.eye(ell, dtype=np.int64), 1, axis=1)
    S_m = np.roll(np.eye(m, dtype=np.int64), 1, axis=1)
    x = np.kron(S_l, np.eye(m, dtype=np.int64))
    y = np.kron(np.eye(ell, dtype=np.int64), S_m)
    mp = np.linalg.matrix_power
    A = (mp(x, 3) + y + mp(y, 2)) % 2
    B = (mp(y, 3) + x + mp(x, 2)) % 2
    HX = np.concatenate([A, B], axis=1) % 2
    HZ = np.concatenate([B.T, A.T], axis=1) % 2
    return HX.astype(np.bool_), HZ.astype(np.bool_)


def random_errors(n: int, batch: int, per: float, seed: int) -> np.ndarray:
    """i.i.d. Bernoulli(per) error patterns, [batch][n] uint8 (numpy PCG64, seeded)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return (rng.random((batch, n)) < per).astype(np.uint8)


def syndromes_of(H, errors_bn: np.ndarray) -> np.ndarray:
    """syndrome = (H * e) .% 2 for every row of errors [B][n]; returns [B][s] uint8."""
    M = sp.csr_matrix(H, dtype=np.int32) if not sp.issparse(H) else sp.csr_matrix(H.astype(np.int32))
    return (np.asarray((M @ errors_bn.T.astype(np.int32))) % 2).T.astype(np.uint8).copy()
