"""Executable specification of the data-parallel quadtree retention the HIP kernel implements.

DistributeOctTree (reference src/ORBextractor.cc:572-852) keeps a std::list of nodes and a
vector of points per node.  The same result follows from three facts:
  * a node is a box; its points are the candidates inside it, in candidate order, so the point
    kept for a node is argmax(response) with ties to the lowest candidate index;
  * children are always pushed at the front, so list order == reverse creation order, and an
    explicit `order` array (front -> back) replaces the list;
  * every pass (whole-list sweep or the one-at-a-time phase) splits the nodes with > 1 point;
    passes differ only in processing order (list order, or (size desc, newest first)) and in
    the early stop once the node count reaches N.
Slots: the first non-empty child re-uses its parent's slot, so at most N+3 (+roots) slots exist.
"""
import numpy as np


def _f32(v):
    return np.float32(v)


def octree_model(xs, ys, resp, W, H, N):
    xs = np.asarray(xs, np.int64)
    ys = np.asarray(ys, np.int64)
    resp = np.asarray(resp, np.int64)
    n = len(xs)
    nIni = int(np.floor(float(_f32(W) / _f32(H)) + 0.5))  # roundf, positive argument
    hX = _f32(W) / _f32(nIni)
    box = []  # per slot [x0, x1, y0, y1]
    count = []
    node_of = np.zeros(n, np.int64)
    for i in range(nIni):
        box.append([int(hX * _f32(i)), int(hX * _f32(i + 1)), 0, H])
        count.append(0)
    for p in range(n):
        r = int(_f32(xs[p]) / hX)
        node_of[p] = r
        count[r] += 1
    order = [s for s in range(nIni) if count[s] > 0]

    def child_counts(s):
        x0, x1, y0, y1 = box[s]
        mx = x0 + (x1 - x0 + 1) // 2
        my = y0 + (y1 - y0 + 1) // 2
        sel = node_of == s
        q = (xs[sel] >= mx).astype(int) + 2 * (ys[sel] >= my).astype(int)
        return [int((q == k).sum()) for k in range(4)], mx, my

    def split(s, created):
        """Split slot s; append the non-empty children (n1..n4 order) to `created`."""
        cc, mx, my = child_counts(s)
        x0, x1, y0, y1 = box[s]
        cbox = [[x0, mx, y0, my], [mx, x1, y0, my], [x0, mx, my, y1], [mx, x1, my, y1]]
        sel = np.nonzero(node_of == s)[0]
        q = (xs[sel] >= mx).astype(int) + 2 * (ys[sel] >= my).astype(int)
        first = True
        for k in range(4):
            if cc[k] == 0:
                continue
            if first:
                slot = s
                first = False
            else:
                slot = len(box)
                box.append(None)
                count.append(0)
            box[slot] = cbox[k]
            count[slot] = cc[k]
            node_of[sel[q == k]] = slot
            created.append(slot)
        return sum(1 for c in cc if c > 0)

    careful = False
    while True:
        prev = len(order)
        pos = {s: i for i, s in enumerate(order)}
        expand = [s for s in order if count[s] > 1]
        if careful:
            expand.sort(key=lambda s: (-count[s], pos[s]))
        created = []
        was_split = set()
        nalive = len(order)
        for s in expand:
            k = split(s, created)
            was_split.add(pos[s])
            nalive += k - 1
            if careful and nalive >= N:
                break
        order = created[::-1] + [s for i, s in enumerate(order) if i not in was_split]
        assert len(order) == nalive
        n_expand = sum(1 for s in created if count[s] > 1)
        if len(order) >= N or len(order) == prev:
            break
        if not careful and len(order) + 3 * n_expand > N:
            careful = True
    out = []
    for s in order:
        sel = np.nonzero(node_of == s)[0]
        best = sel[0]
        for p in sel[1:]:
            if resp[p] > resp[best]:
                best = p
        out.append(int(best))
    return out
