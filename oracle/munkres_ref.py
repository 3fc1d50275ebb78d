"""Restatement of the Kuhn-Munkres solver the reference calls.  TEST INFRASTRUCTURE.

Third-party dependency, absent from /root/reference: ``munkres`` (requirements.txt:12,
unpinned; pinned here to 1.1.4 = the copy on this image at
/opt/conda/lib/python3.9/site-packages/munkres.py).  Call site:
/root/reference/lib/core/group.py:19-23 (``Munkres().compute(scores)``), :80.

The published algorithm (Munkres 1957, six-step "star/prime/cover" form) is
restated with the tie-breaking of that implementation, because the result among
equal-cost optima depends on it:
  * non-square input is padded to n = max(rows, cols) with 0          (pad_matrix)
  * step 2 stars the first zero of each row whose row/col are uncovered
  * step 4's zero search starts at the previous (row, col), walks rows
    cyclically, and inside the first row that has an uncovered zero keeps the
    LAST hit of the cyclic column walk                              (__find_a_zero)
  * step 6 applies ``+= m`` (covered row) then ``-= m`` (uncovered col) as two
    separate float64 roundings
Pinned against the real module by tests/golden/gen_golden.py (random float,
tied-integer and 1e10-padded matrices).
"""
import numpy as np


def compute(cost):
    """cost: 2-D array-like (rows <= cols after the caller's padding is NOT
    required).  Returns list of (row, col) for the original rows/cols, row-major."""
    cost = np.asarray(cost, dtype=np.float64)
    rows, cols = cost.shape
    n = max(rows, cols)
    C = np.zeros((n, n), dtype=np.float64)
    C[:rows, :cols] = cost
    row_cov = [False] * n
    col_cov = [False] * n
    marked = [[0] * n for _ in range(n)]

    # step 1
    for i in range(n):
        C[i, :] = C[i, :] - C[i, :].min()
    # step 2
    for i in range(n):
        for j in range(n):
            if C[i, j] == 0 and not col_cov[j] and not row_cov[i]:
                marked[i][j] = 1
                col_cov[j] = True
                row_cov[i] = True
                break
    row_cov = [False] * n
    col_cov = [False] * n

    def find_a_zero(i0, j0):
        row = col = -1
        i = i0
        done = False
        while not done:
            j = j0
            while True:
                if C[i, j] == 0 and not row_cov[i] and not col_cov[j]:
                    row, col = i, j
                    done = True
                j = (j + 1) % n
                if j == j0:
                    break
            i = (i + 1) % n
            if i == i0:
                done = True
        return row, col

    step = 3
    z0r = z0c = 0
    while True:
        if step == 3:
            count = 0
            for i in range(n):
                for j in range(n):
                    if marked[i][j] == 1 and not col_cov[j]:
                        col_cov[j] = True
                        count += 1
            if count >= n:
                break
            step = 4
        elif step == 4:
            row = col = 0
            while True:
                row, col = find_a_zero(row, col)
                if row < 0:
                    step = 6
                    break
                marked[row][col] = 2
                star_col = -1
                for j in range(n):
                    if marked[row][j] == 1:
                        star_col = j
                        break
                if star_col >= 0:
                    col = star_col
                    row_cov[row] = True
                    col_cov[col] = False
                else:
                    z0r, z0c = row, col
                    step = 5
                    break
        elif step == 5:
            path = [(z0r, z0c)]
            while True:
                c = path[-1][1]
                r = -1
                for i in range(n):
                    if marked[i][c] == 1:
                        r = i
                        break
                if r < 0:
                    break
                path.append((r, c))
                cc = -1
                for j in range(n):
                    if marked[r][j] == 2:
                        cc = j
                        break
                path.append((r, cc))
            for (r, c) in path:
                marked[r][c] = 0 if marked[r][c] == 1 else 1
            row_cov = [False] * n
            col_cov = [False] * n
            for i in range(n):
                for j in range(n):
                    if marked[i][j] == 2:
                        marked[i][j] = 0
            step = 3
        else:  # step 6
            m = None
            for i in range(n):
                if row_cov[i]:
                    continue
                for j in range(n):
                    if not col_cov[j] and (m is None or m > C[i, j]):
                        m = C[i, j]
            for i in range(n):
                for j in range(n):
                    if row_cov[i]:
                        C[i, j] = C[i, j] + m
                    if not col_cov[j]:
                        C[i, j] = C[i, j] - m
            step = 4
    out = []
    for i in range(rows):
        for j in range(cols):
            if marked[i][j] == 1:
                out.append((i, j))
    return out
