"""Robot self-collision tables (host side): which geom pairs MuJoCo's ``mj_collision`` would test between two bodies of
the robot, and the capsule proxy of every collision geom used by the narrow phase of those pairs.

MuJoCo's pair filter (engine_collision_driver.c, restated from its documentation - the reference models rely on the
defaults, e.g. ``aliengo.xml:8-10,42,61,71``, ``mini_cheetah.xml:33-35,66,75,92``; ``spot.xml:177-186`` adds excludes):
two geoms are tested iff they sit on different bodies, ``(contype1 & conaffinity2) | (contype2 & conaffinity1)`` is
non-zero, the bodies are not parent and child (``filterparent``; every body of these models has a joint, so weld groups
are the bodies themselves) and the body pair is not named in a ``<contact><exclude>``.  Pairs are ordered like MuJoCo's
mid phase: by body pair (body1 < body2), then geom1, then geom2.

Narrow phase (csrc/gq_selfcol.h, oracle/gq_oracle.c): every geom is represented by a capsule in its BODY frame - exact for
sphere and capsule geoms; the inscribed capsule along the principal axis of the hull for box / cylinder / mesh geoms
(MuJoCo runs box-box SAT or its general convex routine there: documented deviation) - and a pair yields
at most one contact at the closest points of the two segments.
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation

GEOM_SPHERE, GEOM_CAPSULE = 2, 3


def geom_capsules(md) -> np.ndarray:
    """``[ngeom, 7]`` (p0, p1, r) in the body frame of each collision geom's proxy capsule; zeros for the others."""
    out = np.zeros((md.ngeom, 7))
    for g in range(md.ngeom):
        c = int(md.geom_cloudid[g])
        if c < 0:
            continue
        R = Rotation.from_quat(np.asarray(md.geom_quat[g]), scalar_first=True).as_matrix()
        pos = np.asarray(md.geom_pos[g])
        v = np.asarray(md.vert_pos[md.cloud_vertadr[c]:md.cloud_vertadr[c] + md.cloud_vertnum[c]])
        r0 = float(md.cloud_radius[c])
        if len(v) == 1:
            p0 = p1 = v[0]; r = r0
        elif len(v) == 2:
            p0, p1, r = v[0], v[1], r0
        else:
            # INSCRIBED capsule of the hull along its principal axis: the largest radius that fits at the centroid (90 % of
            # the distance to the nearest facet plane), then the longest segment through the centroid along the axis that
            # keeps the capsule inside every facet.  A proxy that never sticks out of the geom cannot create contacts MuJoCo
            # would not have (bounding capsules of trunk boxes / meshes overlap the thighs in the rest pose); it finds them
            # late, once the hulls overlap by the difference between the hull and its proxy.
            from scipy.spatial import ConvexHull
            cen = v.mean(0)
            _, _, vt = np.linalg.svd(v - cen, full_matrices=False)
            ax = vt[0] / np.linalg.norm(vt[0])
            try:
                eq = ConvexHull(v).equations          # n.x + d <= 0 inside
                n, d = eq[:, :3], eq[:, 3]
                depth = -(n @ cen + d)                # distance of the centroid to each facet plane (> 0)
                r = 0.9 * float(depth.min())
                slope = n @ ax
                room = depth - r                      # >= 0
                tp = np.min(np.where(slope > 1e-9, room / np.maximum(slope, 1e-9), np.inf))
                tm = np.min(np.where(slope < -1e-9, room / np.maximum(-slope, 1e-9), np.inf))
                tp, tm = (0.0 if not np.isfinite(tp) else float(tp)), (0.0 if not np.isfinite(tm) else float(tm))
            except Exception:   # degenerate (flat) hull
                r, tp, tm = 1e-3, 0.0, 0.0
            p0, p1 = cen - tm * ax, cen + tp * ax
            r += r0
        out[g, 0:3] = pos + R @ p0
        out[g, 3:6] = pos + R @ p1
        out[g, 6] = r
    return out


def self_pairs(md) -> np.ndarray:
    """``[npair, 2]`` int32 geom ids (geom1 < geom2) of the robot-robot pairs MuJoCo would pass to its narrow phase."""
    excl = {(int(a), int(b)) for a, b in zip(md.exclude_body1, md.exclude_body2)} | {(int(b), int(a)) for a, b in zip(md.exclude_body1, md.exclude_body2)}
    col = [g for g in range(md.ngeom) if md.geom_cloudid[g] >= 0 and md.geom_bodyid[g] > 0]
    pairs = []
    for i, g1 in enumerate(col):
        for g2 in col[i + 1:]:
            b1, b2 = int(md.geom_bodyid[g1]), int(md.geom_bodyid[g2])
            if b1 == b2:
                continue
            if not ((md.geom_contype[g1] & md.geom_conaffinity[g2]) or (md.geom_contype[g2] & md.geom_conaffinity[g1])):
                continue
            if md.body_parentid[b1] == b2 or md.body_parentid[b2] == b1:
                continue
            if (b1, b2) in excl:
                continue
            pairs.append((b1, b2, g1, g2))
    pairs.sort()   # MuJoCo walks body pairs (body1 < body2) and, inside one, geom1 x geom2
    return np.asarray([(g1, g2) for _, _, g1, g2 in pairs], dtype=np.int32).reshape(-1, 2)
