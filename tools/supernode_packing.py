#!/usr/bin/env python3
"""ANALYSIS TOOLING (round 6): what the context-tree layout of k_maniac_decode costs in records and walk rounds, from the REAL trees of a
stream and the number of walks through every node (oracle dump: FO_DUMP_TREES=/tmp/trees.bin python ... Port().decode(blob)).

    python tools/supernode_packing.py /tmp/trees.bin [min_nodes]

For every group with at least `min_nodes` tree nodes: the supernodes of the shipped cut (complete 6-level subtrees, one record each) and of
the BUDDY packing (a subtree that fits k < 6 levels takes a 2^-(6-k) share of a record: the sub-heap rooted at a slot of level 6 - k), the
walk rounds behind the root per symbol (identical for both: the cut is the same, only the record assignment differs) and the context bytes."""
import struct
import sys


def load(path):
    data = open(path, "rb").read()
    pos, groups = 0, []
    while pos < len(data):
        c, size, nref, nprops = struct.unpack_from("<4i", data, pos); pos += 16
        nodes = [struct.unpack_from("<4i", data, pos + 16 * k) for k in range(size)]; pos += 16 * size
        groups.append((c, nref, nprops, nodes))
    return groups


def analyse(c, nodes):
    n = len(nodes)
    inner = sum(1 for p, _, _, _ in nodes if p >= 0)
    leaves = n - inner
    walks = nodes[0][3]
    # the 6-level cut: supernode roots = the tree root and every inner node at depth 6 below a supernode root
    roots, stack = [], [(0, 0)]
    depth_in = {}
    sub_nodes, sub_depth = {}, {}
    order = []
    stack = [0]
    sn_roots = [0]
    k = 0
    rounds = 0
    while k < len(sn_roots):
        r = sn_roots[k]; k += 1
        cnt, maxd = 0, 0
        frontier = [(r, 0)]
        while frontier:
            t, d = frontier.pop()
            p, child, _, vis = nodes[t]
            if p < 0:
                continue
            if d == 6:
                sn_roots.append(t)
                rounds += vis
                continue
            cnt += 1; maxd = max(maxd, d + 1)
            frontier.append((child, d + 1)); frontier.append((child + 1, d + 1))
        sub_nodes[r] = cnt; sub_depth[r] = maxd
    n_super = len(sn_roots)
    # buddy packing of the supernodes behind the root: a subtree of depth k (levels) takes 2^(k-6) of a record
    shares = sorted((2.0 ** (sub_depth[r] - 6) for r in sn_roots[1:]), reverse=True)
    records, free = 0, []   # first-fit decreasing over power-of-two shares = exact buddy allocation
    for s in shares:
        for i, f in enumerate(free):
            if f >= s - 1e-12:
                free[i] = f - s
                break
        else:
            records += 1; free.append(1.0 - s)
    hist = {}
    for r in sn_roots[1:]:
        hist[sub_depth[r]] = hist.get(sub_depth[r], 0) + 1
    print("c%-3d nodes %5d leaves %5d walks %8d  supernodes %4d (root %2d nodes; behind it %s by depth, %.1f nodes each)  rounds behind the root %.3f / symbol" % (
        c, n, leaves, walks, n_super, sub_nodes[0], " ".join("%d:%d" % (d, hist[d]) for d in sorted(hist)), (inner - sub_nodes[0]) / max(1, n_super - 1), rounds / max(1, walks)))
    for name, rec_bytes, nrec in (("wide (8 B per lane)", 512, n_super), ("narrow (4 B per lane)", 256, n_super), ("narrow + buddy packing", 256, 1 + records)):
        print("       %-24s %4d records  %6.1f KB of supernodes + %5.1f KB of 64-byte leaves = %6.1f KB" % (name, nrec, nrec * rec_bytes / 1024.0, leaves * 64 / 1024.0, (nrec * rec_bytes + leaves * 64) / 1024.0))


if __name__ == "__main__":
    min_nodes = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    for c, nref, nprops, nodes in load(sys.argv[1]):
        if len(nodes) >= min_nodes:
            analyse(c, nodes)


def simulate_tails(c, nodes):
    """narrow supernodes + TAILS: a child subtree of at most two levels (<= 3 nodes) is stored in free lanes (absent heap slots, lane 63) of the record
    that exits to it and walked by scalar code after the round -- no record, no memory round trip of its own.  Exact placement: free lanes of every
    record in heap order, a tail's nodes in consecutive free lanes."""
    walks = nodes[0][3]

    def height(t, cap=3):
        p, ch, _, _ = nodes[t]
        if p < 0 or cap == 0:
            return 0 if p < 0 else 99
        return 1 + max(height(ch, cap - 1), height(ch + 1, cap - 1))

    records, rounds, tail_walks, tails = 0, 0, 0, 0
    todo = [0]
    while todo:
        r = todo.pop(0)
        records += 1
        slot_used = [False] * 64
        exits = []   # (exit node t) for inner nodes at depth 6
        fr = [(r, 0, 0)]
        while fr:
            t, d, slot = fr.pop()
            p, ch, _, vis = nodes[t]
            if p < 0:
                continue
            if d == 6:
                exits.append(t)
                continue
            slot_used[slot] = True
            fr.append((ch, d + 1, 2 * slot + 1)); fr.append((ch + 1, d + 1, 2 * slot + 2))
        free = [i for i in range(64) if not slot_used[i]]   # (lane 63 is never a slot)
        for t in sorted(exits, key=lambda t: -nodes[t][3]):
            h = height(t)
            n_nodes = 0
            if h <= 2:
                p, ch, _, _ = nodes[t]
                n_nodes = 1 + (1 if nodes[ch][0] >= 0 else 0) + (1 if nodes[ch + 1][0] >= 0 else 0)
                # consecutive free lanes
                pos = next((k for k in range(len(free) - n_nodes + 1) if free[k + n_nodes - 1] - free[k] == n_nodes - 1), None)
                if pos is not None and r != 0:
                    del free[pos:pos + n_nodes]
                    tails += 1; tail_walks += nodes[t][3]
                    continue
            todo.append(t); rounds += nodes[t][3]
    leaves = sum(1 for p, _, _, _ in nodes if p < 0)
    print("       narrow + tails           %4d records  %6.1f KB of supernodes + %5.1f KB of 64-byte leaves = %6.1f KB;  rounds behind the root %.3f / symbol, "
          "%d tails walked by %.3f of the symbols" % (records, records * 0.25, leaves * 64 / 1024.0, records * 0.25 + leaves * 64 / 1024.0, rounds / walks, tails, tail_walks / walks))


if __name__ == "__main__" and len(sys.argv) > 1:
    print("---- tails")
    for c, nref, nprops, nodes in load(sys.argv[1]):
        if len(nodes) >= (int(sys.argv[2]) if len(sys.argv) > 2 else 600):
            print("c%d" % c)
            simulate_tails(c, nodes)
