"""Second, independently written restatement of the reference BP-OTS decoder
(src/decoders/bpots_decoder.jl) -- TEST INFRASTRUCTURE, parity unpinned.  Dense s x n message
matrices, Python floats, the reference's helper functions one for one.  tanh / atanh come from the
portable implementations (ldpcdecoders.jl_amd/csrc/portable_math.h, exported by the C oracle's
shared object) so that this file and bpots_oracle.c can be compared bit for bit."""
from __future__ import annotations

import ctypes
import math

from .bpots import _lib


def _fn(name):
    f = getattr(_lib(), name)
    f.restype = ctypes.c_double
    f.argtypes = [ctypes.c_double]
    return f


class DensePyBPOTS:
    def __init__(self, H, per, max_iters, T=9, C=2.0):
        self.s, self.n = len(H), len(H[0]) if H else 0
        self.per, self.max_iters, self.T, self.C = per, max_iters, T, C
        self.var_neighbors = [[i for i in range(self.s) if H[i][j]] for j in range(self.n)]      # :97-109
        self.check_neighbors = [[j for j in range(self.n) if H[i][j]] for i in range(self.s)]
        self.tanh, self.atanh = _fn("pm_tanh_export"), _fn("pm_atanh_export")

    def update_variable_to_check(self, j, i, Om):          # :158-172
        msg_sum = 0.0
        for check in self.var_neighbors[j]:
            if check != i:
                msg_sum += self.cv[check][j]
        self.vc[i][j] = Om[j] + msg_sum

    def update_check_to_variable(self, i, j, syndrome):    # :178-210
        prod_tanh = 1.0
        MAX_TANH = 0.99999
        for var in self.check_neighbors[i]:
            if var != j:
                t = self.tanh(0.5 * self.vc[i][var])
                t = min(MAX_TANH, max(-MAX_TANH, t))
                prod_tanh *= t
        if syndrome[i] != 0:
            prod_tanh = -prod_tanh
        if abs(prod_tanh) >= MAX_TANH:
            prod_tanh = MAX_TANH if prod_tanh > 0 else -MAX_TANH
        msg = 2.0 * self.atanh(prod_tanh)
        msg = min(100.0, max(-100.0, msg))
        self.cv[i][j] = msg

    def decode(self, syndrome):                            # :225-340
        s, n = self.s, self.n
        self.vc = [[0.0] * n for _ in range(s)]
        self.cv = [[0.0] * n for _ in range(s)]
        osc = [0] * n
        prior_dec = [0] * n
        Pi = [math.log((1 - (2 * self.per / 3)) / (2 * self.per / 3))] * n
        Om = list(Pi)
        best = [0] * n
        best_mismatch, best_weight = len(syndrome), n
        self.iters = 0
        for it in range(1, self.max_iters + 1):
            self.iters = it
            for j in range(n):
                for i in self.var_neighbors[j]:
                    self.update_variable_to_check(j, i, Om)
            for i in range(s):
                for j in self.check_neighbors[i]:
                    self.update_check_to_variable(i, j, syndrome)
            llrs, dec = [0.0] * n, [0] * n
            for j in range(n):                             # compute_beliefs! :120-136
                llr = Om[j]
                for i in self.var_neighbors[j]:
                    llr += self.cv[i][j]
                llrs[j] = llr
                dec[j] = 1 if llr < 0.0 else 0
            if it > 1:
                for j in range(n):
                    osc[j] += dec[j] ^ prior_dec[j]
            prior_dec = list(dec)
            mismatch = sum(1 for i in range(s)
                           if (sum(dec[j] for j in self.check_neighbors[i]) % 2) != syndrome[i])
            weight = sum(dec)
            if mismatch < best_mismatch or (mismatch == best_mismatch and weight < best_weight):
                best_mismatch, best_weight, best = mismatch, weight, list(dec)
                if mismatch == 0:
                    return best, True
            if mismatch > 0 and it % self.T == 0:
                Om = list(Pi)
                if max(osc) > 0:
                    max_osc, j1, min_llr = 0, -1, math.inf
                    for j in range(n):
                        if osc[j] > max_osc:
                            max_osc, j1, min_llr = osc[j], j, abs(llrs[j])
                        elif osc[j] == max_osc and abs(llrs[j]) < min_llr:
                            j1, min_llr = j, abs(llrs[j])
                    if j1 >= 0:
                        osc[j1] = 0
                        Om[j1] = -self.C
                    j2, min_llr = 0, abs(llrs[0])
                    for j in range(1, n):
                        if abs(llrs[j]) < min_llr:
                            j2, min_llr = j, abs(llrs[j])
                    Om[j2] = -self.C
        return best, False
