"""Second, independently written CPU restatement of the reference BP decoder.

TEST INFRASTRUCTURE ONLY (see oracle/bp_oracle.c for the rules).  PARITY
UNPINNED: no Julia runtime and no reference golden vectors exist here; this
file exists so that two restatements written separately from the same source
lines can be diffed bit for bit.

It walks src/decoders/belief_propagation.jl:121-188 literally -- dense
``s x n`` message matrices, Python ``float`` (IEEE-754 double) scalars, the same
loop nesting and the same statement order -- and deliberately shares no code
with bp_oracle.c.  Pure-Python loops: small cases only.
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple


class DensePyBP:
    """`BeliefPropagationDecoder` (belief_propagation.jl:38-67) on a dense 0/1 H."""

    def __init__(self, H: Sequence[Sequence[int]], per: float, max_iters: int):
        self.s = len(H)
        self.n = len(H[0]) if self.s else 0
        self.per = float(per)
        self.max_iters = int(max_iters)
        # sparse(H): for every column j the row indices of its ones, ascending (:63)
        self.col_rows: List[List[int]] = [
            [i for i in range(self.s) if H[i][j]] for j in range(self.n)
        ]
        # sparse(H'): for every row i the column indices of its ones, ascending (:64)
        self.row_cols: List[List[int]] = [
            [j for j in range(self.n) if H[i][j]] for i in range(self.s)
        ]
        self.reset()

    def reset(self) -> None:  # :83-91
        s, n = self.s, self.n
        self.log_probabs = [0.0] * n
        self.channel_probs = [self.per] * n
        self.bit_2_check = [[0.0] * n for _ in range(s)]
        self.check_2_bit = [[0.0] * n for _ in range(s)]
        self.err = [0.0] * n
        self.iters = 0

    @staticmethod
    def _div(a: float, b: float) -> float:
        """IEEE-754 division (Python raises on /0 where Julia returns Inf/NaN)."""
        try:
            return a / b
        except ZeroDivisionError:
            if a != a or a == 0.0:
                return math.nan
            neg = (math.copysign(1.0, a) < 0) != (math.copysign(1.0, b) < 0)
            return -math.inf if neg else math.inf

    @staticmethod
    def _log(x: float) -> float:
        if x == 0.0:
            return -math.inf
        if x != x:
            return math.nan
        return math.log(x)

    def decode(self, syndrome: Sequence[int]) -> Tuple[List[float], bool]:  # :121-188
        div = self._div
        self.reset()  # :122
        b2c, c2b = self.bit_2_check, self.check_2_bit
        for j in range(self.n):  # :127-131
            for i in self.col_rows[j]:
                b2c[i][j] = div(self.channel_probs[j], 1 - self.channel_probs[j])
        converged = False
        for it in range(1, self.max_iters + 1):  # :134
            self.iters = it
            for i in range(self.s):  # :135-150
                temp = float((-1) ** int(syndrome[i]))  # :136
                for j in self.row_cols[i]:  # :137-141
                    c2b[i][j] = temp
                    temp *= div(2, 1 + b2c[i][j]) - 1
                temp = 1.0  # :143
                for j in reversed(self.row_cols[i]):  # :144-149
                    c2b[i][j] *= temp
                    c2b[i][j] = div(1 - c2b[i][j], 1 + c2b[i][j])
                    temp *= div(2, 1 + b2c[i][j]) - 1
            for j in range(self.n):  # :152-178
                temp = div(self.channel_probs[j], 1 - self.channel_probs[j])  # :153
                for i in self.col_rows[j]:  # :155-161
                    b2c[i][j] = temp
                    temp *= c2b[i][j]
                    if temp != temp:
                        temp = 1.0
                self.log_probabs[j] = self._log(div(1, temp))  # :163
                self.err[j] = 1.0 if temp >= 1 else 0.0  # :164-168
                temp = 1.0  # :170
                for i in reversed(self.col_rows[j]):  # :171-177
                    b2c[i][j] *= temp
                    temp *= c2b[i][j]
                    if temp != temp:
                        temp = 1.0
            # :180-184
            decoded = [math.fmod(sum(self.err[j] for j in self.row_cols[i]), 2.0) for i in range(self.s)]
            if all(decoded[i] == syndrome[i] for i in range(self.s)):
                converged = True
                break
        return self.err, converged  # :187

    def messages_csc(self) -> Tuple[List[float], List[float]]:
        """Messages at the structural non-zeros, CSC edge order (column by column)."""
        b, c = [], []
        for j in range(self.n):
            for i in self.col_rows[j]:
                b.append(self.bit_2_check[i][j])
                c.append(self.check_2_bit[i][j])
        return b, c
