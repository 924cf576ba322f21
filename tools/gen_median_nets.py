#!/usr/bin/env python3
"""Generate mlvfs_amd/csrc/median_nets.h: straight-line min/max selection code
for the chroma-smoothing medians (reference: opt_med5/9/25, mlvfs/opt_med.h,
used by mlvfs/chroma_smooth.c:61-62 -- any exact selector gives the same value).

Design (sliding 5x5 window on a plane, one thread produces a horizontal strip):
  1. every column of 5 vertically adjacent cells is sorted once        (sort5)
  2. adjacent column pairs are merged into sorted 10-lists             (merge 5+5)
  3. two pair lists give the ranks 8..13 of their 20 elements           (quad)
     -- the only ranks of a 4-column group that can still be the median of 25
  4. median of 25 = 6th smallest of {6 quad candidates} U {5th column} (final)
  Steps 1-3 are shared by neighbouring outputs of the strip.

The networks are built functionally (Batcher odd-even merge for arbitrary
lengths), dead nodes are removed for the requested outputs, every network is
verified here by the 0-1 principle / exhaustive sorted-input enumeration and
against Python's sorted() on random integers, and then emitted as C++ that
compiles for both host and device.
"""
from __future__ import annotations

import itertools
import os
import random
import sys


class Net:
    """SSA builder for min/max expressions."""

    def __init__(self, n_in):
        self.n_in = n_in
        self.nodes = []                      # (op, a, b) with operands = node ids; inputs are ids 0..n_in-1

    def inp(self, i):
        return i

    def _add(self, op, a, b):
        self.nodes.append((op, a, b))
        return self.n_in + len(self.nodes) - 1

    def mn(self, a, b):
        return self._add("mn", a, b)

    def mx(self, a, b):
        return self._add("mx", a, b)

    def ce(self, a, b):
        return self.mn(a, b), self.mx(a, b)

    # Batcher odd-even merge of two sorted lists of node ids (any lengths)
    def merge(self, A, B):
        if not A:
            return list(B)
        if not B:
            return list(A)
        if len(A) == 1 and len(B) == 1:
            lo, hi = self.ce(A[0], B[0])
            return [lo, hi]
        V = self.merge(A[0::2], B[0::2])
        W = self.merge(A[1::2], B[1::2])
        out = [V[0]]
        i = 0
        while i < len(W) and i + 1 < len(V):
            lo, hi = self.ce(W[i], V[i + 1])
            out += [lo, hi]
            i += 1
        out += V[i + 1:]
        out += W[i:]
        return out

    def sort(self, X):
        if len(X) <= 1:
            return list(X)
        h = len(X) // 2
        return self.merge(self.sort(X[:h]), self.sort(X[h:]))

    def min_of(self, xs):
        xs = list(xs)
        while len(xs) > 1:
            xs = [self.mn(xs[0], xs[1])] + xs[2:]
        return xs[0]

    # evaluation / pruning / emission
    def evaluate(self, vals, outs):
        v = list(vals)
        for op, a, b in self.nodes:
            v.append(min(v[a], v[b]) if op == "mn" else max(v[a], v[b]))
        return [v[o] for o in outs]

    def live(self, outs):
        need = set(outs)
        for idx in range(len(self.nodes) - 1, -1, -1):
            nid = self.n_in + idx
            if nid in need:
                _, a, b = self.nodes[idx]
                need.add(a)
                need.add(b)
        return need

    def emit(self, name, in_groups, outs, out_name="o"):
        """in_groups: list of (c_name, count) describing the flat input order."""
        need = self.live(outs)
        nops = sum(1 for idx in range(len(self.nodes)) if self.n_in + idx in need)
        names = {}
        k = 0
        params = []
        for gname, cnt in in_groups:
            params.append(f"const T (&{gname})[{cnt}]")
            for j in range(cnt):
                names[k] = f"{gname}[{j}]"
                k += 1
        assert k == self.n_in
        lines = [f"// {nops} min/max ops", "template <class T>",
                 f"MLV_NET_FN void {name}({', '.join(params)}, T (&{out_name})[{len(outs)}])", "{"]
        for idx, (op, a, b) in enumerate(self.nodes):
            nid = self.n_in + idx
            if nid not in need:
                continue
            names[nid] = f"t{idx}"
            lines.append(f"    const T t{idx} = mlv_{op}({names[a]}, {names[b]});")
        for j, o in enumerate(outs):
            lines.append(f"    {out_name}[{j}] = {names[o]};")
        lines.append("}")
        return "\n".join(lines), nops


def check_merge(net, la, lb, outs, ranks):
    """all sorted 0-1 inputs + random ints"""
    for za in range(la + 1):
        for zb in range(lb + 1):
            vals = [0] * za + [1] * (la - za) + [0] * zb + [1] * (lb - zb)
            ref = sorted(vals)
            got = net.evaluate(vals, outs)
            assert got == [ref[r] for r in ranks], (la, lb, za, zb)
    rnd = random.Random(1)
    for _ in range(300):
        a = sorted(rnd.randrange(-50, 50) for _ in range(la))
        b = sorted(rnd.randrange(-50, 50) for _ in range(lb))
        ref = sorted(a + b)
        assert net.evaluate(a + b, outs) == [ref[r] for r in ranks]


def check_select(net, n, outs, ranks, exhaustive_limit=20):
    if n <= exhaustive_limit:
        for bits in itertools.product((0, 1), repeat=n):
            ref = sorted(bits)
            assert net.evaluate(list(bits), outs) == [ref[r] for r in ranks]
    rnd = random.Random(2)
    for _ in range(2000):
        v = [rnd.randrange(-9, 9) for _ in range(n)]
        ref = sorted(v)
        assert net.evaluate(v, outs) == [ref[r] for r in ranks]


def main():
    out_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "mlvfs_amd", "csrc", "median_nets.h")
    parts = []
    summary = []

    # merge of two sorted 5-lists -> sorted 10
    n = Net(10)
    res = n.merge([0, 1, 2, 3, 4], [5, 6, 7, 8, 9])
    check_merge(n, 5, 5, res, list(range(10)))
    code, ops = n.emit("mlv_merge55", [("a", 5), ("b", 5)], res)
    parts.append(code)
    summary.append(("mlv_merge55", ops))

    # ranks 8..13 (1-based) of two sorted 10-lists
    n = Net(20)
    res = n.merge(list(range(10)), list(range(10, 20)))
    outs = res[7:13]
    check_merge(n, 10, 10, outs, list(range(7, 13)))
    code, ops = n.emit("mlv_quad_mid6", [("a", 10), ("b", 10)], outs)
    parts.append(code)
    summary.append(("mlv_quad_mid6", ops))

    # 6th smallest of a sorted 6-list and a sorted 5-list:
    # k-th smallest of a union = min over splits i+j=k of max(A_i, B_j)
    n = Net(11)
    A, B = list(range(6)), list(range(6, 11))
    terms = [A[5]]
    for i in range(1, 6):                    # i elements from A (1-based A_i), 6-i from B
        terms.append(n.mx(A[i - 1], B[6 - i - 1]))
    o = n.min_of(terms)
    check_merge(n, 6, 5, [o], [5])
    code, ops = n.emit("mlv_final6of11", [("c", 6), ("s", 5)], [o])
    parts.append(code)
    summary.append(("mlv_final6of11", ops))

    # full sort of a column of 5 with two-input ops only (the packed 16-bit path has no three-input min/med/max)
    n = Net(5)
    res = n.sort(list(range(5)))
    check_select(n, 5, res, list(range(5)))
    code, ops = n.emit("mlv_sort5", [("v", 5)], res)
    parts.append(code)
    summary.append(("mlv_sort5", ops))

    # plain selection networks (used for the 5- and 9-element windows, and as the
    # 25-element cross-check in the self test)
    for cnt in (5, 9, 25):
        n = Net(cnt)
        res = n.sort(list(range(cnt)))
        check_select(n, cnt, [res[cnt // 2]], [cnt // 2])
        code, ops = n.emit(f"mlv_median{cnt}", [("v", cnt)], [res[cnt // 2]])
        parts.append(code)
        summary.append((f"mlv_median{cnt}", ops))

    header = [
        "// GENERATED by tools/gen_median_nets.py -- do not edit.",
        "// Exact min/max selection networks for the chroma-smoothing medians",
        "// (replaces opt_med5/9/25 of mlvfs/opt_med.h; values are ints, so any exact",
        "// selector is bit-identical).  Templates over the element type: int, or a pair",
        "// of 16-bit lanes with element-wise mlv_mn/mlv_mx (k_frame's packed path).",
        "// Verified at generation time (0-1 principle)",
        "// and again at run time by mlvfs_amd_selftest_host().",
        "#pragma once",
        "#ifndef MLV_NET_FN",
        "#define MLV_NET_FN static inline",
        "#endif",
        "#ifndef mlv_mn",
        "#define mlv_mn(a, b) ((a) < (b) ? (a) : (b))",
        "#define mlv_mx(a, b) ((a) > (b) ? (a) : (b))",
        "#endif",
        "",
    ]
    with open(out_path, "w") as f:
        f.write("\n".join(header) + "\n\n".join(parts) + "\n")
    for name, ops in summary:
        print(f"{name}: {ops} ops")
    print("wrote", os.path.normpath(out_path))


if __name__ == "__main__":
    sys.setrecursionlimit(10000)
    main()
