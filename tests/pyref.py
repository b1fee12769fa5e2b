"""Independent numpy / pure-Python re-derivations used to cross-check oracle/orb_oracle.c.

These are second statements of the same published algorithms, written in a different shape
(vectorised formulas, the array form of the quadtree) so a coding slip in one of the two
shows up as a diff.  Test infrastructure only.
"""
from __future__ import annotations

import math

import numpy as np

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3),
        (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
GAUSS = np.array([18, 34, 48, 56, 48, 34, 18], np.int64)


def fast_response(img: np.ndarray) -> np.ndarray:
    """R(x,y) = max over the 16 arcs of 9 contiguous ring pixels of min(v-p), and of
    min(p-v).  A pixel is a FAST-9 corner at threshold t iff R > t; its score is R - 1."""
    img = img.astype(np.int64)
    h, w = img.shape
    r = np.full((h, w), -256, np.int64)
    c = img[3:h - 3, 3:w - 3]
    d = np.stack([c - img[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in RING])  # v - p
    best = np.full(c.shape, -256, np.int64)
    for s in range(16):
        idx = [(s + k) % 16 for k in range(9)]
        best = np.maximum(best, d[idx].min(axis=0))
        best = np.maximum(best, (-d[idx]).min(axis=0))
    r[3:h - 3, 3:w - 3] = best
    return r


def fast_score_map(img: np.ndarray, t: int) -> np.ndarray:
    r = fast_response(img)
    return np.where(r > t, r - 1, 0).astype(np.uint8)


def resize_linear(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    sh, sw = src.shape
    s = src.astype(np.int64)

    def taps(dn, sn):
        scale = 1.0 / (float(dn) / sn)
        d = np.arange(dn, dtype=np.float64)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        i = np.floor(f).astype(np.int64)
        f = (f - i.astype(np.float32)).astype(np.float32)
        return i, f

    ix, fx = taps(dw, sw)
    fx = np.where(ix < 0, np.float32(0), fx)
    ix = np.maximum(ix, 0)
    fx = np.where(ix >= sw - 1, np.float32(0), fx)
    ix = np.minimum(ix, sw - 1)
    a0 = np.rint((np.float32(1) - fx) * np.float32(2048)).astype(np.int64)
    a1 = np.rint(fx * np.float32(2048)).astype(np.int64)
    ix1 = np.minimum(ix + 1, sw - 1)
    hrow = s[:, ix] * a0[None, :] + s[:, ix1] * a1[None, :]

    iy, fy = taps(dh, sh)
    b0 = np.rint((np.float32(1) - fy) * np.float32(2048)).astype(np.int64)
    b1 = np.rint(fy * np.float32(2048)).astype(np.int64)
    y0 = np.clip(iy, 0, sh - 1)
    y1 = np.clip(iy + 1, 0, sh - 1)
    r0, r1 = hrow[y0], hrow[y1]
    out = (((b0[:, None] * (r0 >> 4)) >> 16) + ((b1[:, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def blur(img: np.ndarray) -> np.ndarray:
    p = np.pad(img.astype(np.int64), 3, mode="reflect")  # numpy 'reflect' == BORDER_REFLECT_101
    h, w = img.shape
    hp = sum(GAUSS[i] * p[:, i:i + w] for i in range(7))
    vp = sum(GAUSS[j] * hp[j:j + h, :] for j in range(7))
    return ((vp + 32768) >> 16).astype(np.uint8)


def fast_atan2_deg(y: float, x: float) -> float:
    a = math.degrees(math.atan2(y, x))
    return a + 360.0 if a < 0 else a


def popcount_dist(q: np.ndarray, t: np.ndarray) -> np.ndarray:
    """(nq, nt) Hamming distances of 32-byte descriptors."""
    x = q[:, None, :] ^ t[None, :, :]
    return np.unpackbits(x, axis=2).sum(axis=2).astype(np.int64)


def match(q, t, th=50, num=9, den=10, exclude_self=False):
    d = popcount_dist(q, t)
    if exclude_self:
        d[np.arange(min(len(q), len(t))), np.arange(min(len(q), len(t)))] = 1 << 20
    nq = len(q)
    idx = np.full(nq, -1, np.int32)
    d1 = np.full(nq, 0xFFFF, np.int64)
    d2 = np.full(nq, 0xFFFF, np.int64)
    for i in range(nq):
        row = d[i]
        ok = row < (1 << 20)
        if not ok.any():
            continue
        j1 = int(np.argmin(row))  # first minimum = lowest index
        d1[i] = row[j1]
        rest = np.delete(row, j1)
        rest = rest[rest < (1 << 20)]
        if len(rest):
            d2[i] = rest.min()
        if d1[i] <= th and d1[i] * den < d2[i] * num:
            idx[i] = j1
    return idx, d1.astype(np.uint16), d2.astype(np.uint16)


# ---------------------------------------------------------------------------------------
# Quadtree in ARRAY form: the formulation the device kernel uses.  std::list with
# push_front / erase is replaced by an append-only node array whose list order is
# "descending creation index"; every node owns a contiguous, order-preserving segment of
# a permutation array.  tests check it against the oracle's literal std::list version.
# ---------------------------------------------------------------------------------------
def distribute_array_form(cand, w, h, n_wanted, sort_fn):
    """cand: sequence of (x, y, response) relative to the (16,16) origin, upstream order.
    sort_fn(sizes, ulxs) -> permutation of range(n) as libstdc++ std::sort leaves it."""
    min_x, max_x, min_y, max_y = 16, w - 16, 16, h - 16
    n_ini = int(math.floor(np.float32(max_x - min_x) / np.float32(max_y - min_y) + np.float32(0.5)))
    hx = np.float32(max_x - min_x) / np.float32(n_ini)

    nodes = []  # dict(x0,x1,y0,y1,keys,alive,no_more)

    def new_node(x0, x1, y0, y1, keys):
        nodes.append(dict(x0=x0, x1=x1, y0=y0, y1=y1, keys=keys, alive=True,
                          no_more=(len(keys) == 1)))
        return len(nodes) - 1

    # roots: list order r0, r1, ... == descending creation index -> create in reverse
    roots = [None] * n_ini
    buckets = [[] for _ in range(n_ini)]
    for i, (x, y, r) in enumerate(cand):
        buckets[int(np.float32(x) / hx)].append(i)
    for i in range(n_ini - 1, -1, -1):
        roots[i] = new_node(int(hx * np.float32(i)), int(hx * np.float32(i + 1)), 0, max_y - min_y,
                            buckets[i])
    size = 0
    for nd in nodes:
        if len(nd["keys"]) == 0:
            nd["alive"] = False
        else:
            size += 1

    def split(idx):
        nonlocal size
        nd = nodes[idx]
        half_x = int(math.ceil(np.float32(nd["x1"] - nd["x0"]) / np.float32(2)))
        half_y = int(math.ceil(np.float32(nd["y1"] - nd["y0"]) / np.float32(2)))
        xm, ym = nd["x0"] + half_x, nd["y0"] + half_y
        k = [[], [], [], []]
        for i in nd["keys"]:
            x, y, _ = cand[i]
            k[(0 if x < xm else 1) + (0 if y < ym else 2)].append(i)
        rects = [(nd["x0"], xm, nd["y0"], ym), (xm, nd["x1"], nd["y0"], ym),
                 (nd["x0"], xm, ym, nd["y1"]), (xm, nd["x1"], ym, nd["y1"])]
        made = []
        for q in range(4):
            if k[q]:
                made.append(new_node(*rects[q], k[q]))
                size += 1
        nd["alive"] = False
        size -= 1
        return made

    finish = False
    while not finish:
        prev_size = size
        n_at_start = len(nodes)
        n_to_expand = 0
        vsz = []
        for idx in range(n_at_start - 1, -1, -1):
            nd = nodes[idx]
            if not nd["alive"] or nd["no_more"]:
                continue
            for c in split(idx):
                if len(nodes[c]["keys"]) > 1:
                    n_to_expand += 1
                    vsz.append(c)
        if size >= n_wanted or size == prev_size:
            finish = True
        elif size + n_to_expand * 3 > n_wanted:
            while not finish:
                prev_size = size
                vprev = vsz
                vsz = []
                perm = sort_fn([len(nodes[c]["keys"]) for c in vprev], [nodes[c]["x0"] for c in vprev])
                for j in range(len(vprev) - 1, -1, -1):
                    for c in split(vprev[perm[j]]):
                        if len(nodes[c]["keys"]) > 1:
                            vsz.append(c)
                    if size >= n_wanted:
                        break
                if size >= n_wanted or size == prev_size:
                    finish = True

    out = []
    for idx in range(len(nodes) - 1, -1, -1):
        nd = nodes[idx]
        if not nd["alive"]:
            continue
        best = nd["keys"][0]
        for i in nd["keys"][1:]:
            if cand[i][2] > cand[best][2]:
                best = i
        out.append(cand[best])
    return out


def parse_c_int_table(path: str, macro: str):
    """Integers of a `#define MACRO { ... }` brace list in a C header."""
    import re
    text = open(path).read()
    start = text.index("#define " + macro)
    body = text[text.index("{", start):]
    depth, end = 0, 0
    for i, ch in enumerate(body):
        depth += ch == "{"
        depth -= ch == "}"
        if depth == 0:
            end = i
            break
    return [int(v) for v in re.findall(r"-?\d+", body[:end].replace("\\\n", " "))]


# ---------------------------------------------------------------------------------------
# libstdc++ std::sort (introsort) in Python, comparator-driven, used to (a) cross-check the
# oracle's C restatement once more and (b) run McIlroy's "killer adversary" against it to obtain
# inputs that reach the depth limit and the heapsort fallback.
# ---------------------------------------------------------------------------------------
def std_sort_py(a, less):
    n = len(a)
    if n == 0:
        return a

    def adjust_heap(first, hole, length, value):
        top, second = hole, hole
        while second < (length - 1) // 2:
            second = 2 * (second + 1)
            if less(a[first + second], a[first + second - 1]):
                second -= 1
            a[first + hole] = a[first + second]
            hole = second
        if (length & 1) == 0 and second == (length - 2) // 2:
            second = 2 * (second + 1)
            a[first + hole] = a[first + second - 1]
            hole = second - 1
        parent = (hole - 1) // 2
        while hole > top and less(a[first + parent], value):
            a[first + hole] = a[first + parent]
            hole = parent
            parent = (hole - 1) // 2
        a[first + hole] = value

    def heap_sort(first, last):
        length = last - first
        if length >= 2:
            parent = (length - 2) // 2
            while True:
                adjust_heap(first, parent, length, a[first + parent])
                if parent == 0:
                    break
                parent -= 1
        while last - first > 1:
            last -= 1
            v = a[last]
            a[last] = a[first]
            adjust_heap(first, 0, last - first, v)

    def linear_insert(last):
        val = a[last]
        nxt = last - 1
        while less(val, a[nxt]):
            a[last] = a[nxt]
            last = nxt
            nxt -= 1
        a[last] = val

    def insertion(first, last):
        for i in range(first + 1, last):
            if less(a[i], a[first]):
                val = a[i]
                a[first + 1:i + 1] = a[first:i]
                a[first] = val
            else:
                linear_insert(i)

    stack = [(0, n, 2 * (n.bit_length() - 1))]
    while stack:
        first, last, depth = stack.pop()
        while last - first > 16:
            if depth == 0:
                heap_sort(first, last)
                break
            depth -= 1
            mid = first + (last - first) // 2
            ia, ib, ic = first + 1, mid, last - 1
            if less(a[ia], a[ib]):
                pick = ib if less(a[ib], a[ic]) else (ic if less(a[ia], a[ic]) else ia)
            elif less(a[ia], a[ic]):
                pick = ia
            elif less(a[ib], a[ic]):
                pick = ic
            else:
                pick = ib
            a[first], a[pick] = a[pick], a[first]
            lo, hi = first + 1, last
            while True:
                while less(a[lo], a[first]):
                    lo += 1
                hi -= 1
                while less(a[first], a[hi]):
                    hi -= 1
                if not lo < hi:
                    break
                a[lo], a[hi] = a[hi], a[lo]
                lo += 1
            stack.append((lo, last, depth))
            last = lo
    if n > 16:
        insertion(0, 16)
        for i in range(16, n):
            linear_insert(i)
    else:
        insertion(0, n)
    return a


def quicksort_killer(n):
    """McIlroy, 'A Killer Adversary for Quicksort': values that drive std_sort_py to its depth limit."""
    gas = n - 1
    val = [gas] * n
    state = {"nsolid": 0, "candidate": 0}

    def less(x, y):
        if val[x] == gas and val[y] == gas:
            if x == state["candidate"]:
                val[x] = state["nsolid"]
            else:
                val[y] = state["nsolid"]
            state["nsolid"] += 1
        if val[x] == gas:
            state["candidate"] = x
        elif val[y] == gas:
            state["candidate"] = y
        return val[x] < val[y]

    std_sort_py(list(range(n)), less)
    return val
