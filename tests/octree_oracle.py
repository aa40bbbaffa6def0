"""Brute-force restatement of apply_oct (several_leg_octree.cu:19-488, octree_util.cu.h) -- TEST
INFRASTRUCTURE.  Pure Python over the CPU oracle's distance_global.  The reference path is dead
code (its call site sits after several_leg.cpp:224's `return 0`), launches kernels from kernels and
ORs its flags through racy shared booleans, so no reference run pins it: "parity unpinned" except
through the (reference-pinned) distance_global it composes.  Semantics as in csrc/lrm_octree.hip:
global ORs of the three per-child flags, onEdge = edge_any and not leaf_any."""
import numpy as np

F = np.float32


def quat_from_angle_index(oracle, idx, st):
    rpy = []
    red = idx
    for i in range(3):
        max_ind = st.angle_sample[i]
        ind = red % max_ind
        ind = (ind + ind // 2) % max_ind
        red //= max_ind
        x = F(ind) / F(max(max_ind - 1, 1))
        rpy.append(F(F(F(1) - x) * F(st.angle_minmax[2 * i]) + F(x * F(st.angle_minmax[2 * i + 1]))))
    q = oracle.qt_multiply(oracle.quat_from_vect_angle((0, 1, 0), rpy[1]), oracle.quat_from_vect_angle((1, 0, 0), rpy[0]))
    return oracle.qt_multiply(oracle.quat_from_vect_angle((0, 0, 1), rpy[2]), q)


def create_child_box(pc, ph, index, min_box):
    quadr = ((index & 1) << 2) | (index & 2) | ((index >> 2) & 1)
    c, h = pc.copy(), ph.copy()
    div = [F(2), F(2), F(2)]
    missing = 0
    for q in range(3):
        if h[q] < F(min_box):
            missing += 1
            if (quadr >> 2) & 1:
                return None
            mask = 0xF ^ ((1 << q) - 1)
            quadr = (quadr & ~mask) | (((quadr & mask) << 1) & mask)
            div[q] = F(1)
    for q in range(3):
        old = h[q]
        h[q] = F(old / div[q])
        move = F(old - h[q])
        c[q] = F(c[q] + (-move if (quadr >> q) & 1 else move))
    return c, h, missing


def in_box(v, h):
    h = np.abs(h)
    return ((h[0] >= v[:, 0]) & (h[1] >= v[:, 1]) & (h[2] >= v[:, 2]) &
            (-h[0] < v[:, 0]) & (-h[1] < v[:, 1]) & (-h[2] < v[:, 2]))


def octree_legs(dim, st):
    legs = []
    for l in range(st.leg_count):
        leg = np.array(dim, F).copy()
        leg[0] = F(st.leg_mount[l])
        legs.append(leg)
    return legs


def child_flags(oracle, footholds, c, h, parent_h, parent_valid, rot, st, legs, quats, reach_len):
    """validity_child (several_leg_octree.cu:19-151) of ONE child box over the given footholds, with global ORs:
    -> (reach_any, leaf_any, edge_any), the three flag bits csrc/lrm_octree.hip's kernels return per child."""
    n_angles_max = st.angle_sample[0] * st.angle_sample[1] * st.angle_sample[2]
    vect = (footholds - c).astype(F)
    keep = in_box(vect, (parent_h + reach_len).astype(F))
    vect = vect[keep]
    reach_any = leaf_any = edge_any = False
    if len(vect):
        margin = F(0) if rot else F(F(st.enable_rot_below) / F(3))
        h2 = F(F(h[0] * h[0] + h[1] * h[1]) + h[2] * h[2])
        for a in range(n_angles_max if rot else 1):
            reach_count = np.zeros(len(vect), int)
            cross_count = np.zeros(len(vect), int)
            for leg in legs:
                d, sub = oracle.dist(vect, leg, quats[a])
                if h2 > F(st.convex_radius) * F(st.convex_radius):
                    cross = in_box(d, h)
                else:
                    dd = ((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(F) + d[:, 2] * d[:, 2]).astype(F)
                    cross = dd < F(h2 + margin)
                cross_count += cross
                reach_count += sub.astype(int)
            edge = cross_count > st.leg_count - st.leg_number_for_stab
            reach = (reach_count >= st.leg_number_for_stab) | bool(parent_valid)
            reach_any |= bool(reach.any())
            leaf_any |= bool((reach & ~edge).any())
            edge_any |= bool(edge.any())
    return reach_any, leaf_any, edge_any


def apply_oct(oracle, footholds, dim, st):
    footholds = np.ascontiguousarray(footholds, F).reshape(-1, 3)
    n_angles_max = st.angle_sample[0] * st.angle_sample[1] * st.angle_sample[2]
    quats = [quat_from_angle_index(oracle, a, st) for a in range(n_angles_max)]
    legs = octree_legs(dim, st)
    reach_len = F(F(F(dim[1] + dim[3]) + dim[5]) + dim[4])
    nodes = [dict(c=np.array(list(st.box_center), F), h=np.array(list(st.box_size), F), validity=False, leaf=False,
                  raw=True, on_edge=False, dead=False, children=None)]
    expand = [0]
    for depth in range(st.max_depth):
        if not expand:
            break
        level = []
        for pi in expand:
            parent = nodes[pi]
            parent["children"] = []
            rot = parent["h"][0] < F(st.enable_rot_below)
            for ci in range(8):
                r = create_child_box(parent["c"], parent["h"], ci, st.min_box)
                if r is None:
                    n = dict(c=np.zeros(3, F), h=np.zeros(3, F), validity=True, leaf=True, raw=False, on_edge=True,
                             dead=True, children=None)
                else:
                    c, h, missing = r
                    n = dict(c=c, h=h, validity=False, leaf=(3 - missing <= 0), raw=not (3 - missing <= 0),
                             on_edge=False, dead=False, children=None)
                nodes.append(n)
                idx = len(nodes) - 1
                parent["children"].append(idx)
                level.append((idx, parent, rot))
            parent["raw"] = False
        for idx, parent, rot in level:
            n = nodes[idx]
            if n["validity"]:
                continue
            reach_any, leaf_any, edge_any = child_flags(oracle, footholds, n["c"], n["h"], parent["h"], bool(parent["validity"]), rot,
                                                        st, legs, quats, reach_len)
            if reach_any:
                n["validity"] = True
            if leaf_any:
                n["leaf"] = True
            if edge_any and not leaf_any:
                n["on_edge"] = True
        nxt = []
        if depth + 1 < st.max_depth:
            for idx, _, _ in level:
                n = nodes[idx]
                if not n["on_edge"]:
                    n["leaf"] = True
                if not n["leaf"]:
                    nxt.append(idx)
        expand = nxt
    out = []

    def walk(i):
        for ci in nodes[i]["children"] or []:
            c = nodes[ci]
            endpoint = not (c["leaf"] or c["raw"] or c["dead"])
            valid = (not c["dead"]) and (c["leaf"] or c["raw"]) and c["validity"]
            if endpoint:
                walk(ci)
            elif valid:
                out.append(c["c"])
    walk(0)
    return np.array(out, F).reshape(-1, 3), len(nodes)
