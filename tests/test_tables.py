"""Base-graph tables: structural invariants of TS 38.212 Tables 5.3.2-1/-2/-3 and, where the
reference checkout exists (build container only), entry-by-entry equality with
get_3gpp_base_graph.m / get_3gpp_valid_lifting_sizes.m."""
import os
import re

import numpy as np
import pytest

from conftest import ALL_Z

BG1_DEG = [19, 19, 19, 19, 3, 8, 9, 7, 10, 9, 7, 8, 7, 6, 7, 7, 6, 6, 6, 6, 6, 6, 5, 5, 6, 5, 5, 4, 5, 5, 5, 5, 5, 5, 5,
           5, 5, 4, 5, 5, 4, 5, 4, 5, 5, 4]
BG2_DEG = [8, 10, 8, 10, 4, 6, 6, 6, 4, 5, 5, 5, 4, 5, 5, 4, 5, 5, 4, 4, 4, 4, 3, 4, 4, 3, 5, 3, 4, 3, 5, 3, 4, 4, 4,
           4, 4, 3, 4, 4, 4, 4]
BG1_MAX = [255, 383, 319, 223, 283, 351, 207, 237]
BG2_MAX = [254, 190, 158, 222, 143, 175, 205, 239]
REF = "/root/reference/get_3gpp_base_graph.m"


def _raw(orc, bg, ils):
    """(rows, cols, raw shifts for set ils): use the largest Z of the set so that 'mod Z' is the identity
    on everything but the maximum-shift check below."""
    zmax = max(z for z in ALL_Z if orc.set_index(z) == ils)
    return orc.graph_edges(bg, zmax), zmax


@pytest.mark.parametrize("bg,deg,nnz,dims", [(1, BG1_DEG, 316, (46, 68)), (2, BG2_DEG, 197, (42, 52))])
def test_structure(orc, bg, deg, nnz, dims):
    (r, c, s), _ = _raw(orc, bg, 0)
    assert len(r) == nnz and r.max() + 1 == dims[0] and c.max() + 1 == dims[1]
    assert np.bincount(r).tolist() == deg
    kb = 22 if bg == 1 else 10
    # extension-parity columns: degree 1, on the diagonal, shift 0
    for col in range(kb + 4, dims[1]):
        e = np.nonzero(c == col)[0]
        assert len(e) == 1 and r[e[0]] == col - kb and s[e[0]] == 0
    # every row >= 4 ends with its own extension column
    for row in range(4, dims[0]):
        assert c[r == row].max() == kb + row
    assert (np.diff(r) >= 0).all()  # CSR order
    for row in range(dims[0]):
        assert (np.diff(c[r == row]) > 0).all()


def test_lifting_sets(orc):
    assert len(ALL_Z) == 51
    for ils, a in enumerate((2, 3, 5, 7, 9, 11, 13, 15)):
        for z in ALL_Z:
            if z % a == 0 and (z // a) & (z // a - 1) == 0:
                assert orc.set_index(z) == ils
    for z in (1, 17, 19, 21, 23, 25, 27, 29, 31, 33, 385, 0, 512):
        assert orc.set_index(z) == -1
    assert orc.lifting_size(22, 8448) == 384 and orc.lifting_size(10, 3840) == 384
    assert orc.lifting_size(6, 116) == 20 and orc.lifting_size(10, 1957) == 208
    assert orc.lifting_size(22, 8449) == -1


@pytest.mark.parametrize("bg,mx", [(1, BG1_MAX), (2, BG2_MAX)])
def test_max_shift_per_set(orc, bg, mx):
    for ils in range(8):
        (r, c, s), zmax = _raw(orc, bg, ils)
        assert s.max() == mx[ils] % zmax or s.max() == mx[ils]


@pytest.mark.skipif(not os.path.exists(REF), reason="reference checkout not present (GPU box)")
def test_equals_reference_file(orc):
    txt = open(REF).read()
    tabs = re.findall(r"table\{(\d)\}\s*=\s*\[(.*?)\];", txt, re.S)
    assert len(tabs) == 2
    for bg_s, body in tabs:
        bg = int(bg_s)
        rows = np.array([[int(t) for t in ln.split()] for ln in body.strip().splitlines() if ln.strip()])
        assert rows.shape[1] == 10
        for ils in range(8):
            for z in (zz for zz in ALL_Z if orc.set_index(zz) == ils):
                r, c, s = orc.graph_edges(bg, z)
                assert (r == rows[:, 0]).all() and (c == rows[:, 1]).all()
                assert (s == rows[:, 2 + ils] % z).all()  # get_pcm.m:8
