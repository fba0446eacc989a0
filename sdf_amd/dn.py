"""Dimension-agnostic operators (reference sdf/dn.py:7-114).

Same call signatures as the reference; each returns an IR :class:`~sdf_amd.ir.Node`
where the reference returns a closure.  d2.py / d3.py register them as methods through
``op2`` / ``op3`` exactly like reference sdf/d3.py:524-532 and sdf/d2.py:290-298.
"""
import itertools

import numpy as np

from .ir import Node


def _boolean(op, a, bs, k):
    # the smoothing constant is resolved per right operand when the tree is lowered
    # (reference sdf/dn.py:12: ``K = k or getattr(b, '_k', None)`` runs inside f(p))
    return Node(op, (), (a,) + tuple(bs), dim=0, meta={'k': k})


def union(a, *bs, k=None):
    """min fold, polynomial smooth-min when a constant applies (reference sdf/dn.py:7-20)"""
    return _boolean('union', a, bs, k)


def difference(a, *bs, k=None):
    """max(d1, -d2) fold (reference sdf/dn.py:22-34)"""
    return _boolean('difference', a, bs, k)


def intersection(a, *bs, k=None):
    """max fold (reference sdf/dn.py:36-48)"""
    return _boolean('intersection', a, bs, k)


def blend(a, *bs, k=0.5):
    """K*d2 + (1-K)*d1 fold (reference sdf/dn.py:50-58)"""
    return _boolean('blend', a, bs, k)


def negate(other):
    """reference sdf/dn.py:60-63"""
    return Node('negate', (), (other,), dim=0)


def dilate(other, r):
    """reference sdf/dn.py:65-68"""
    return Node('dilate', (r,), (other,), dim=0)


def erode(other, r):
    """reference sdf/dn.py:70-73"""
    return Node('erode', (r,), (other,), dim=0)


def shell(other, thickness):
    """|d| - thickness/2 (reference sdf/dn.py:75-78)"""
    return Node('shell', (thickness / 2,), (other,), dim=0)


def repeat(other, spacing, count=None, padding=0):
    """domain repetition: min over the (2*padding+1)^dim neighbour cells of the rounded
    cell index (reference sdf/dn.py:80-112).  The dimension is only known where the node
    is used, so the neighbour list is expanded by `repeat_params` at lowering time."""
    count = np.array(count) if count is not None else None
    spacing = np.array(spacing)
    return Node('repeat', (), (other,), dim=0,
                meta={'spacing': spacing, 'count': count, 'padding': padding})


def repeat_params(node, dim):
    """flat constants of a repeat node used on `dim`-dimensional points:
    [dim, s0,s1,s2, has_count, c0,c1,c2, n_neighbours, (n0,n1,n2)*]"""
    spacing, count, padding = node.meta['spacing'], node.meta['count'], node.meta['padding']
    try:
        pad = [padding[i] for i in range(dim)]
    except Exception:
        pad = [padding] * dim
    try:
        sp = [spacing[i] for i in range(dim)]
    except Exception:
        sp = [spacing] * dim
    for i, s in enumerate(sp):
        if s == 0:
            pad[i] = 0
    neigh = list(itertools.product(*[list(range(-p, p + 1)) for p in pad]))
    s3 = np.zeros(3)
    s3[:dim] = np.broadcast_to(np.asarray(spacing, dtype=np.float64), (dim,))
    c3 = np.zeros(3)
    if count is not None:
        c3[:dim] = np.broadcast_to(np.asarray(count, dtype=np.float64), (dim,))
    out = [float(dim)] + list(s3) + [0.0 if count is None else 1.0] + list(c3) + [float(len(neigh))]
    for n in neigh:
        n3 = list(n) + [0] * (3 - dim)
        out += [float(v) for v in n3]
    return out
