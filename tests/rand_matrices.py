"""random symmetric matrices of many shapes for the schedule / parity tests (test data only)"""
import numpy as np
import scipy.sparse as sp


def random_matrix(rng, n, kind):
    if kind == "band":
        bw = int(rng.integers(1, 40))
        dens = rng.uniform(0.2, 1.0)
        rows, cols = [], []
        for d in range(1, bw + 1):
            keep = rng.random(n - d) < dens
            i = np.arange(d, n)[keep]
            rows.append(i)
            cols.append(i - d)
        r = np.concatenate(rows) if rows else np.zeros(0, int)
        c = np.concatenate(cols) if cols else np.zeros(0, int)
    elif kind == "random":
        m = int(n * rng.uniform(0.5, 8))
        r = rng.integers(1, n, m)
        c = (r * rng.random(m)).astype(int)
    elif kind == "nodes":  # dof-blocks: rows of a node share their columns
        dof = int(rng.integers(2, 8))
        nodes = max(2, n // dof)
        n = nodes * dof
        m = int(nodes * rng.uniform(1, 6))
        a = rng.integers(1, nodes, m)
        b = np.maximum(0, a - 1 - (rng.exponential(8, m)).astype(int))
        keep = b < a
        a, b = a[keep], b[keep]
        r = (a[:, None, None] * dof + np.arange(dof)[None, :, None]).repeat(dof, 2)
        c = (b[:, None, None] * dof + np.arange(dof)[None, None, :]).repeat(dof, 1)
        r, c = r.ravel(), c.ravel()
        # in-node lower entries
        ii, jj = np.tril_indices(dof, -1)
        r = np.concatenate([r, (np.arange(nodes)[:, None] * dof + ii[None, :]).ravel()])
        c = np.concatenate([c, (np.arange(nodes)[:, None] * dof + jj[None, :]).ravel()])
    else:  # hub: a few rows / columns touch very many
        m = int(n * 3)
        r = rng.integers(1, n, m)
        c = (r * rng.random(m)).astype(int)
        hubs = rng.integers(0, n, 3)
        for h in hubs:
            k = int(rng.integers(n // 8, n // 2))
            o = rng.choice(n, k, replace=False)
            o = o[o != h]
            r = np.concatenate([r, np.maximum(o, h)])
            c = np.concatenate([c, np.minimum(o, h)])
    keep = c < r
    r, c = r[keep], c[keep]
    L = sp.coo_matrix((rng.uniform(-1, 1, r.size), (r, c)), shape=(n, n)).tocsr()
    L.sum_duplicates()
    L.data[L.data == 0] = 0.5
    d = rng.uniform(1, 2, n)
    d[rng.random(n) < 0.05] = 0.0  # missing diagonal entries
    A = (L + L.T + sp.diags(d)).tocsr()
    A.eliminate_zeros()
    if rng.random() < 0.3:  # some empty rows
        kill = rng.choice(n, max(1, n // 50), replace=False)
        mask = np.ones(n)
        mask[kill] = 0
        D = sp.diags(mask)
        A = (D @ A @ D).tocsr()
        A.eliminate_zeros()
    A.sort_indices()
    return n, A
