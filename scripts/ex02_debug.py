import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from bensolve_amd.lp import LpEngine
import oracle_api
inf = np.inf
A = np.array([[3, 1], [1, 2], [1, 1.0]]); m, n, q = 3, 2, 2
M, N = m + q + q + 1, n + q + 1
L = np.zeros((M, N))
L[:m, :n] = A
L[m:m + q, :n] = -np.eye(2); L[m:m + q, n:n + q] = np.eye(2)
L[m + q:m + q + q, n:n + q] = np.eye(2); L[m + q:m + q + q, n + q] = -1
lo = np.array([0, 0, 0, 0, 0, -inf, -inf, -inf] + [-inf] * N, float)
up = np.array([0, 0, 0, 0, 0, 0, 0, 1] + [inf] * N, float)
cost = np.zeros(N + 1); cost[N] = 1
o = oracle_api.OracleLP(L, lo, up, cost)
print("oracle", o.solve(1), o.obj())
e = LpEngine(M, N, L, lo, up, cost, 0, 0, 2)
e.reset_slot(0)
st, it = e.solve_batch([0], [0], np.zeros((1, 0)), np.zeros((1, 0)))
print("gpu", st, it, e.obj([0]))
